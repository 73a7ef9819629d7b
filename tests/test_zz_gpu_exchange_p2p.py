"""The peer-to-peer transport of the multi-GPU exchange (ddgi_exchange_p2p_*; csrc/ddgi_exchange.cpp) with MORE THAN
ONE RANK on the one-GPU test box.

RCCL refuses two ranks on one device, so the RCCL tests (test_gpu_exchange.py) only ever run rank 0 of 1 here.  The
peer-to-peer transport has no such limit: every rank pushes its z-slab into the other ranks' textures through
pointers mapped with hipIpcOpenMemHandle (one process per rank), with flag words in device memory as the rendezvous.
These tests run world = 2 and 4 on device 0 — the product's own slab offsets, pair alternation, events and flags at
rank > 0 — and require every rank's gathered field to equal the unsharded engine's, frame by frame, with data that
changes every frame."""
import hashlib
import multiprocessing as mp
import os
import sys

import numpy as np
import pytest

from tests.common import CONFIGS, c3_oracle_albedo, shading_points

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# a quarter of BASELINE's C4 grid (64x32x64 probes x 512 rays in a 32 x 16 ray tile): 16 z-layers, 16.8 M rays, 67 MB per texture —
# a launch is capped at 64 Mi rays, so a group is 2 updates (4 pairs under the pipelined exchange, not 16), the ray tile is not square
SHAPES = dict(CONFIGS, c4_slab=((64, 32, 16), 1, 16, (1.4, 0.0, 1.0), 0))
TILES = {"c4_slab": (32, 16)}


def _engine(ddgi, name, **kw):
    counts, side, s, origin, scene = SHAPES[name]
    eng = ddgi.ProbeEngine(ddgi.make_field(counts, side, s, origin), ddgi.make_settings(scene, 8), **kw)
    if name in TILES:
        eng.set_ray_tile(*TILES[name])
    return eng


def _digest(*arrays):
    h = hashlib.sha1()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def _frame_settings(ddgi, scene, frame):
    return ddgi.make_settings(scene, 8, time=2.0 * (frame + 1))


def _table_batch(name):
    """A batch of shading points large enough for the REF sampler's per-texel table (csrc/ddgi_engine.cpp: sample_box_pays) on this grid."""
    counts, side, s, origin, _ = SHAPES[name]
    tx, ty = TILES.get(name, (s, s))
    texels = counts[0] * counts[1] * counts[2] * tx * ty
    return shading_points(np.random.default_rng(41), counts, side, origin, max(70000, texels // 8 + 1000))


def _run_frames(ddgi, eng, mode, scene, frames, read_at, name=None, progress=None):
    """Drives `frames` updates (+ exchanges when the handle has one) and returns {frame: digest of the full field}; REF modes: and
    {"sample": digest of a batch sampled through the per-texel table of the GATHERED field} — the table spans the whole grid on every rank."""
    out = {}
    has_exchange = eng.exchange_transport()[0] != "none"
    if mode in ("ref_static", "ref_static_engine"):
        eng.generate_probe_rays(seed=1, reseed=True)   # one ray set, updates back to back: consecutive updates are CONTINUED (frames in flight) across the exchanges
    for frame in range(frames):
        if mode == "ref":
            eng.generate_probe_rays(seed=frame + 1, reseed=True)   # new jitter: every frame's texels differ
        eng.probe_update(_frame_settings(ddgi, scene, frame))      # (DDGI mode: new rotation, animated light, hysteresis)
        if has_exchange:
            eng.exchange()
        if progress:
            progress(f"frame {frame} submitted")
        if frame in read_at:
            out[frame] = _digest(*(eng.read_tiles() if mode == "ddgi" else eng.read_textures()))
            if progress:
                progress(f"frame {frame} read back")
    if mode != "ddgi" and name is not None:
        out["sample"] = _digest(*eng.sample(*_table_batch(name)))   # (a consumer: waits for the last exchange by itself)
        if progress:
            progress("sampled")
    return out


_EXPECTED = {}


def _expected(ddgi, name, mode, frames, read_at, oracle=None):
    """What every rank's gathered field must be (computed once per scenario and session: world 2, 4 and 8 share it)."""
    key = (name, mode, frames, tuple(read_at))
    if key not in _EXPECTED:
        _EXPECTED[key] = _expected_now(ddgi, name, mode, frames, read_at, oracle)
    return _EXPECTED[key]


def _expected_now(ddgi, name, mode, frames, read_at, oracle=None):
    counts, side, s, origin, scene = SHAPES[name]
    if mode == "ref_static":
        # the oracle's raster (every frame writes the same texels: Q18); `distances` is never assigned
        if name == "c3_cave":
            want = c3_oracle_albedo(oracle, "pinned", seed=1)
        else:
            f = oracle.make_field(counts, side, s, origin)
            want = oracle.probe_update(f, oracle.make_settings(scene, 8), oracle.generate_probe_rays(f, oracle.new_rand_state(1)))[0]
        out = {frame: _digest(want, np.zeros_like(want)) for frame in read_at}
        out["sample"] = _digest(*oracle.sample(oracle.make_field(counts, side, s, origin), want, np.zeros_like(want), *_table_batch(name)))
        return out
    with _engine(ddgi, name) as eng:
        if mode == "ddgi":
            eng.set_mode(ddgi.MODE_DDGI)
        return _run_frames(ddgi, eng, mode, scene, frames, read_at, name)


SCENARIOS = [
    # (configuration, mode, pipelined, frames, frames after which a consumer reads the whole field)
    ("cave_small", "ref", False, 3, (0, 1, 2)),
    ("cave_small", "ref", True, 5, (1, 3, 4)),      # updates 1, 3 are issued while the exchange before them is in flight
    ("cave_small", "ddgi", False, 3, (0, 1, 2)),
    ("cave_small", "ddgi", True, 5, (0, 2, 3, 4)),  # the temporal blend reads the previous tiles from the OTHER pair
    # BASELINE's C2 and C3 with the pipelined exchange, updates back to back (frames in flight: a launch goes on with the next update's
    # rays into the next pair of the ring while the previous pairs' slabs are still leaving) — expected = the ORACLE's raster
    ("c2_cornell", "ref_static", True, 6, (2, 5)),
    ("c3_cave", "ref_static", True, 7, (3, 6)),
    # round 5 — DDGI mode, updates back to back (frames in flight with inputs that change: rotation, key and the animated light travel
    # in per-update records, the ray records in a ring of buffers), pipelined exchange of the tiles; expected = the unsharded engine
    ("c3_cave", "ddgi", True, 7, (6,)),
    # ... and a C4-shaped slab: non-square ray tile, a ring of pairs shortened by the cap on rays per launch, pipelined; expected = the unsharded engine
    ("c4_slab", "ref_static_engine", True, 6, (5,)),
]


def _worker(rank, world, conn, scenario, log_path):
    """One rank = one process (what bench.py / a real host does); `conn` carries the 512-byte addresses, the progress notes and
    the results.  Every stage is reported with its time, so that a rank that stops answering can be placed; `log_path` receives a
    Python traceback of all threads every 30 s while the worker lives (faulthandler), i.e. where a blocked call was made from."""
    import faulthandler
    import time

    sys.path.insert(0, ROOT)
    log = open(log_path, "w")
    faulthandler.enable(log)
    faulthandler.dump_traceback_later(30, repeat=True, file=log)
    t0 = time.monotonic()

    def progress(stage):
        conn.send(("progress", (stage, round(time.monotonic() - t0, 2))))

    try:
        import ddgi_amd as ddgi

        ddgi.load_library()
        progress("library loaded")
        name, mode, pipelined, frames, read_at = scenario
        scene = SHAPES[name][4]
        eng = _engine(ddgi, name, device=0, rank=rank, world=world)
        if mode == "ddgi":
            eng.set_mode(ddgi.MODE_DDGI)
        else:
            # ranks that have done DIFFERENT numbers of updates before they attach (asymmetric warm-up): attaching starts
            # every rank's count over, on pair 0 — a push must land in the pair its receiver reads (csrc/ddgi_exchange.cpp)
            eng.generate_probe_rays(seed=77)
            for _ in range(1 + rank % 3):
                eng.probe_update()
        progress("engine ready")
        conn.send(("address", eng.exchange_p2p_export(pipelined)))
        progress("exported")
        addresses = conn.recv()
        eng.exchange_p2p_init(addresses)
        progress("peers mapped")
        conn.recv()                 # ("start": every rank has mapped its peers)
        result = _run_frames(ddgi, eng, mode, scene, frames, read_at, name, progress)
        eng.exchange_finish()
        eng.synchronize()
        progress("finished + synchronized")
        conn.send(("done", (eng.get_tuning("p2p_landing_zones"), eng.get_tuning("p2p_exported_mb"))))   # host barrier before any rank tears its buffers down
        conn.recv()
        eng.close()
        conn.send(("results", result))
    except Exception as exc:  # noqa: BLE001 — reported to the parent, which fails the test
        conn.send(("error", repr(exc)))
    finally:
        faulthandler.cancel_dump_traceback_later()
        log.close()


class _Ranks:
    """The parent's side: `world` worker processes and what each was last heard doing."""

    def __init__(self, world, scenario, tmp_path):
        ctx = mp.get_context("spawn")
        self.world = world
        self.logs = [str(tmp_path / f"rank{r}.log") for r in range(world)]
        pipes = [ctx.Pipe() for _ in range(world)]
        self.procs = [ctx.Process(target=_worker, args=(r, world, pipes[r][1], scenario, self.logs[r]), daemon=True) for r in range(world)]
        for p in self.procs:
            p.start()
        self.conns = [pp[0] for pp in pipes]
        self.last = [("spawned", 0.0)] * world

    def _report(self):
        lines = []
        for r in range(self.world):
            p = self.procs[r]
            state = "alive" if p.is_alive() else f"exited with {p.exitcode}"
            lines.append(f"  rank {r}: {state}; last heard: {self.last[r][0]!r} at {self.last[r][1]} s")
            try:
                with open(self.logs[r]) as fh:
                    tail = fh.read()[-1500:]
                if tail.strip():
                    lines.append("    where it stands (faulthandler, newest last):\n      " + tail.strip().replace("\n", "\n      "))
            except OSError:
                pass
        return "\n".join(lines)

    def gather(self, kind, deadline):
        """One message of `kind` from every rank before `deadline` (time.monotonic()); progress notes are kept on the way."""
        import time

        out = []
        for r, c in enumerate(self.conns):
            while True:
                left = deadline - time.monotonic()
                if left <= 0 or not c.poll(left):
                    pytest.fail(f"rank {r} did not send {kind!r} in time.\n" + self._report(), pytrace=False)
                tag, payload = c.recv()
                if tag == "progress":
                    self.last[r] = payload
                    continue
                if tag != kind:
                    pytest.fail(f"rank {r}: {tag} {payload}\n" + self._report(), pytrace=False)
                out.append(payload)
                break
        return out

    def send(self, what):
        for c in self.conns:
            c.send(what)

    def send_in_turn(self, what, stage, deadline):
        """`what` to one rank at a time, the next when the one before has reported `stage`: the ranks map their peers' buffers ONE AFTER THE OTHER (round 6:
        with every rank inside hipIpcOpenMemHandle at once, each waiting for its exporter's process to hand a dmabuf over, a bring-up can stand forever
        — bench.py's C5 case, profiles/r06_c5_bring_up_backtrace.txt; a real host takes turns over whatever channel carries the addresses)."""
        import time

        for r, c in enumerate(self.conns):
            c.send(what)
            while self.last[r][0] != stage:
                left = deadline - time.monotonic()
                if left <= 0 or not c.poll(left):
                    pytest.fail(f"rank {r} did not reach {stage!r} in time.\n" + self._report(), pytrace=False)
                tag, payload = c.recv()
                if tag != "progress":
                    pytest.fail(f"rank {r}: {tag} {payload}\n" + self._report(), pytrace=False)
                self.last[r] = payload

    def close(self):
        for p in self.procs:
            p.join(timeout=10)
            if p.is_alive():
                p.kill()  # (exactly the process started above)


# seconds one scenario may take from the first spawn to the last result, by configuration (spawn + library + engine: a few seconds per process
# on the test box, where every rank shares ONE GPU; C3's frames: ~10 ms each); the deadline covers the mapping of the peers' rings too
BUDGET_S = {"cave_small": 60, "c2_cornell": 60, "c3_cave": 90, "c4_slab": 90}


def _ids():
    return [f"{name}-{mode}-{'pipelined' if pl else 'in_order'}" for name, mode, pl, _, _ in SCENARIOS]


def _run_scenario(ddgi, oracle, tmp_path, world, scenario, landing=False):
    import time

    name, mode, pipelined, frames, read_at = scenario
    want = _expected(ddgi, name, mode, frames, read_at, oracle)
    ranks = _Ranks(world, scenario, tmp_path)
    deadline = time.monotonic() + BUDGET_S[name] * (2 if world > 4 else 1)
    try:
        ranks.send_in_turn(ranks.gather("address", deadline), "peers mapped", deadline)
        ranks.send("start")
        exported = ranks.gather("done", deadline)
        for r in range(world):
            # (textures whose pushes land in zones instead of the ring, MB of this rank a peer maps)
            assert (exported[r][0] > 0) == landing and exported[r][1] > 0, exported
        ranks.send("go")
        results = ranks.gather("results", deadline)
        for r in range(world):
            assert results[r] == want, f"rank {r}, {name} {mode} pipelined={pipelined}: gathered field differs from the unsharded engine's / the oracle's"
    finally:
        ranks.close()


@pytest.mark.parametrize("scenario", SCENARIOS, ids=_ids())
@pytest.mark.parametrize("world", [2, 4])
def test_one_process_per_rank_through_ipc_handles(ddgi, oracle, tmp_path, world, scenario):
    """2 / 4 processes, one GPU: the peers' textures are mapped with hipIpcOpenMemHandle, the flags cross the process
    boundary.  This is the multi-rank path of bench.py --exchange p2p, executed at ranks > 0 on the test box.  One test per
    scenario, each with its own budget; a rank that stops answering is reported with the last stage it reached."""
    _run_scenario(ddgi, oracle, tmp_path, world, scenario)


# Landing zones (csrc/ddgi_exchange.cpp): a grid whose ring reaches 2 GiB (BASELINE's C5: the depth tiles of ONE pair are 2^31 bytes) cannot hand the ring itself
# to its peers on the stack measured; the peers then push into per-parity zones of world - 1 slabs and a stream of the receiver's own copies them into the pair.
# DDGI_P2P_LANDING=1 switches that path on for grids of any size: the same scenarios, the same expectations.
LANDING = [SCENARIOS[i] for i in (0, 1, 3, 5, 6)]


@pytest.mark.parametrize("world, scenario", [(4, sc) for sc in LANDING] + [(2, LANDING[1]), (2, LANDING[4])],
                         ids=lambda v: v if isinstance(v, int) else f"{v[0]}-{v[1]}-{'pipelined' if v[2] else 'in_order'}")
def test_landing_zones_gather_the_same_field(ddgi, oracle, tmp_path, monkeypatch, world, scenario):
    monkeypatch.setenv("DDGI_P2P_LANDING", "1")
    _run_scenario(ddgi, oracle, tmp_path, world, scenario, landing=True)


def test_eight_ranks_on_one_gpu(ddgi, oracle, tmp_path):
    """World 8 — BASELINE's node size — on the one-GPU box: C2 (8 z-layers: one per rank), pipelined, updates back to back, against the oracle."""
    _run_scenario(ddgi, oracle, tmp_path, 8, ("c2_cornell", "ref_static", True, 6, (2, 5)))


def test_p2p_addresses_are_checked(ddgi):
    counts, side, s, origin, scene = CONFIGS["cave_small"]
    with ddgi.ProbeEngine(ddgi.make_field(counts, side, s, origin), ddgi.make_settings(scene, 8), rank=0, world=2) as a, \
            ddgi.ProbeEngine(ddgi.make_field(counts, side, s, origin), ddgi.make_settings(scene, 8), rank=1, world=2) as b:
        with pytest.raises(ddgi.DDGIError):
            a.exchange_p2p_init([b"\0" * ddgi.P2P_ADDRESS_BYTES] * 2)      # no export on this handle yet
        addr_a, addr_b = a.exchange_p2p_export(False), b.exchange_p2p_export(True)
        with pytest.raises(ddgi.DDGIError, match="does not describe rank"):
            a.exchange_p2p_init([addr_a, addr_b])                          # the ranks disagree about pipelining
        addr_a = a.exchange_p2p_export(False)
        with pytest.raises(ddgi.DDGIError, match="does not describe rank"):
            a.exchange_p2p_init([addr_a, addr_a])                          # rank 1's slot holds rank 0's address
        with pytest.raises(ddgi.DDGIError):
            a.exchange()                                                   # (the failed init released the exchange)
        addr_a, addr_b = a.exchange_p2p_export(True), b.exchange_p2p_export(True)
        with pytest.raises(ddgi.DDGIError, match="lives in this process"):
            a.exchange_p2p_init([addr_a, addr_b])                          # one process per rank (see ddgi_exchange.cpp)
        assert a.exchange_transport() == ("none", False)
        a.generate_probe_rays(seed=1)
        a.probe_update()                                                   # the handle is usable, on its own pair
        a.synchronize()
