"""A rank that dies in the middle of an exchange must not hang the others (round 6).

The reference's fences give up after DEFAULT_FENCE_TIMEOUT = 1 s (src/rvpt/vk_util.cpp:65, 94-97).  Here every host wait of a handle
with a multi-GPU exchange attached has the deadline of tuning "wait_timeout_ms": on expiry the call returns DDGI_ERR_TIMEOUT naming the
peer that is behind, the flag, and the exchange numbers expected and seen; the exchange is broken (ddgi_exchange and the consumers
refuse), and — peer-to-peer transport — the survivors' own waits are released by the library itself, so their handles drain and can be
detached, used alone and destroyed.  Three processes on the test box's one GPU; rank 1 leaves with os._exit after the first frame."""
import multiprocessing as mp
import os
import sys
import time

import numpy as np
import pytest

from tests.common import CONFIGS

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORLD, VICTIM, TIMEOUT_MS = 3, 1, 3000
NAME = "cave_odd"   # 3 x 3 x 3 probes: one z-layer per rank


def _worker(rank, conn, pipelined):
    sys.path.insert(0, ROOT)
    try:
        import ddgi_amd as ddgi

        ddgi.load_library()
        counts, side, s, origin, scene = CONFIGS[NAME]
        eng = ddgi.ProbeEngine(ddgi.make_field(counts, side, s, origin), ddgi.make_settings(scene, 8), device=0, rank=rank, world=WORLD)
        eng.set_tuning("wait_timeout_ms", TIMEOUT_MS)
        eng.generate_probe_rays(seed=1)
        conn.send(("address", eng.exchange_p2p_export(pipelined)))
        eng.exchange_p2p_init(conn.recv())
        conn.send(("mapped", None))                 # (the ranks map their peers one after the other: tests/test_zz_gpu_exchange_p2p.py: send_in_turn)
        conn.recv()
        eng.probe_update()
        eng.exchange()
        whole = eng.read_textures()[0].copy()       # frame 0: every rank is there
        conn.send(("frame0", int(whole.astype(np.uint64).sum())))
        conn.recv()                                 # (host barrier: everybody has read frame 0)
        if rank == VICTIM:
            os._exit(0)                             # no teardown, no goodbye: the process is gone
        report = {}
        eng.generate_probe_rays(seed=2, reseed=True)
        eng.probe_update()
        t0 = time.monotonic()
        try:
            eng.exchange()
            eng.read_textures()
            report["first"] = "no error"
        except ddgi.DDGIError as exc:
            report["first"] = (exc.code, str(exc))
        report["first_s"] = time.monotonic() - t0
        t0 = time.monotonic()
        for what, call in (("exchange", eng.exchange), ("sample", lambda: eng.sample(np.zeros((4, 3), np.float32), np.ones((4, 3), np.float32) / np.sqrt(3).astype(np.float32)))):
            try:
                call()
                report[what] = "no error"
            except ddgi.DDGIError as exc:
                report[what] = (exc.code, str(exc))
        report["refusals_s"] = time.monotonic() - t0
        # detached, the handle is a slab of its own again: an update and a read of it work
        t0 = time.monotonic()
        eng.exchange_init(None)
        eng.probe_update()
        eng.synchronize()
        own = eng.read_textures()[0]
        report["alone_ok"] = bool(own.any())
        eng.close()
        report["alone_s"] = time.monotonic() - t0
        conn.send(("report", report))
    except Exception as exc:  # noqa: BLE001 — reported to the parent, which fails the test
        conn.send(("error", repr(exc)))


@pytest.mark.parametrize("pipelined, landing", [(False, False), (True, False), (True, True)], ids=["in_order", "pipelined", "pipelined-landing_zones"])
def test_survivors_of_a_lost_rank_time_out_name_it_and_stay_usable(ddgi, monkeypatch, pipelined, landing):
    if landing:
        # the peers push into landing zones and a stream of this rank's own copies them out (what grids with rings of 2 GiB and more get): ITS waits for
        # the lost rank's slab must end with the others
        monkeypatch.setenv("DDGI_P2P_LANDING", "1")
    ctx = mp.get_context("spawn")
    pipes = [ctx.Pipe() for _ in range(WORLD)]
    procs = [ctx.Process(target=_worker, args=(r, pipes[r][1], pipelined), daemon=True) for r in range(WORLD)]
    for p in procs:
        p.start()
    conns = [pp[0] for pp in pipes]
    deadline = time.monotonic() + 90

    def gather(kind, ranks):
        out = {}
        for r in ranks:
            assert conns[r].poll(max(0.0, deadline - time.monotonic())), f"rank {r} did not send {kind!r} in time"
            tag, payload = conns[r].recv()
            assert tag == kind, f"rank {r}: {tag} {payload}"
            out[r] = payload
        return out

    try:
        everybody = range(WORLD)
        addresses = gather("address", everybody)
        for r in everybody:
            conns[r].send([addresses[q] for q in everybody])
            gather("mapped", [r])
        for c in conns:
            c.send("start")
        sums = gather("frame0", everybody)
        assert len(set(sums.values())) == 1 and sums[0] > 0      # frame 0 was a real, complete exchange
        for c in conns:
            c.send("go")
        survivors = [r for r in everybody if r != VICTIM]
        reports = gather("report", survivors)
        for r in survivors:
            rep = reports[r]
            code, text = rep["first"]
            assert code == ddgi.ERR_TIMEOUT, rep
            # the rank that is gone is named FIRST (a live peer may be listed behind it: its pushes can stand behind its own wait for the victim)
            assert f"rank {VICTIM} is behind: `ready` at exchange 1, `arrived` at exchange 1; this rank ({r} of {WORLD}) waits for exchange 2 / 2" in text, text
            assert "streams have drained" in text, text
            # the live peer is NOT behind: its slab came, and this rank's push to it went out — the exchange's streams have hardware queues of their own
            # priority (csrc/ddgi_exchange.cpp: create_exchange_stream), two peer streams do not share one, and nothing stands behind the wait for the victim
            assert "also behind" not in text, text
            # within the deadline (+ the bounded clean-up: reading the flags, releasing the waits, draining)
            assert TIMEOUT_MS / 1000 * 0.9 <= rep["first_s"] <= TIMEOUT_MS / 1000 + 8, rep
            for what in ("exchange", "sample"):
                assert rep[what][0] == ddgi.ERR_TIMEOUT and "broken" in rep[what][1], rep
            assert rep["refusals_s"] < 2 and rep["alone_ok"] and rep["alone_s"] < 15, rep
    finally:
        for p in procs:
            p.join(timeout=10)
            if p.is_alive():
                p.kill()  # (exactly the process started above)


def _mapping_worker(rank, conn, limit_ms):
    sys.path.insert(0, ROOT)
    try:
        import ddgi_amd as ddgi

        ddgi.load_library()
        counts, side, s, origin, scene = CONFIGS[NAME]
        eng = ddgi.ProbeEngine(ddgi.make_field(counts, side, s, origin), ddgi.make_settings(scene, 8), device=0, rank=rank, world=WORLD)
        eng.set_tuning("wait_timeout_ms", limit_ms)
        eng.set_tuning("ablate", 64)               # (profiling build: the thread that maps a peer's buffers sleeps for a minute)
        eng.generate_probe_rays(seed=1)
        conn.send(("address", eng.exchange_p2p_export(True)))
        everyone = conn.recv()
        t0 = time.monotonic()
        try:
            eng.exchange_p2p_init(everyone)
            outcome = "no error"
        except ddgi.DDGIError as exc:
            outcome = (exc.code, str(exc))
        seconds = time.monotonic() - t0
        eng.set_tuning("ablate", 0)
        eng.probe_update()                         # the handle is on its own again (the failed init released the exchange) and works
        eng.synchronize()
        alone = bool(eng.read_textures()[0].any())
        conn.send(("report", dict(outcome=outcome, seconds=seconds, transport=eng.exchange_transport()[0], alone=alone)))
        conn.recv()
        eng.close()
    except Exception as exc:  # noqa: BLE001
        conn.send(("error", repr(exc)))


def test_a_peer_mapping_that_does_not_come_back_is_a_timeout_not_a_hang(ddgi):
    """hipIpcOpenMemHandle is a driver call that can stand forever (round 6: a texture ring of 2 GiB or more inside an engine's process,
    profiles/r06_p2p_ring_size_bisection.txt).  ddgi_exchange_p2p_init runs it on a helper thread and waits with tuning "wait_timeout_ms":
    here the profiling build's fault injection (tuning "ablate" 64) keeps that thread asleep — every rank must get DDGI_ERR_TIMEOUT naming
    the peer and the buffer within the deadline, be left without an exchange, and go on working alone."""
    limit_ms = 1500
    os.environ["DDGI_LIB"] = ddgi.build_profiling_library()
    try:
        ctx = mp.get_context("spawn")
        pipes = [ctx.Pipe() for _ in range(WORLD)]
        procs = [ctx.Process(target=_mapping_worker, args=(r, pipes[r][1], limit_ms), daemon=True) for r in range(WORLD)]
        for p in procs:
            p.start()
    finally:
        os.environ.pop("DDGI_LIB", None)
    conns = [pp[0] for pp in pipes]
    try:
        addresses = []
        for r in range(WORLD):
            assert conns[r].poll(60), f"rank {r} did not export in time"
            tag, payload = conns[r].recv()
            assert tag == "address", (r, tag, payload)
            addresses.append(payload)
        for c in conns:
            c.send(addresses)
        for r in range(WORLD):
            assert conns[r].poll(60), f"rank {r} did not report in time"
            tag, rep = conns[r].recv()
            assert tag == "report", (r, tag, rep)
            code, text = rep["outcome"]
            assert code == ddgi.ERR_TIMEOUT and "hipIpcOpenMemHandle of rank" in text and "did not return within 1500 ms" in text, rep
            assert limit_ms / 1000 * 0.9 <= rep["seconds"] <= limit_ms / 1000 + 5, rep
            assert rep["transport"] == "none" and rep["alone"], rep
        for c in conns:
            c.send("bye")
    finally:
        for p in procs:
            p.join(timeout=10)
            if p.is_alive():
                p.kill()  # (exactly the process started above)
