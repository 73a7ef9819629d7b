#!/usr/bin/env python3
"""bench.py — probe-rays/s and ms/frame of the DDGI probe update on MI355X.

A "step" is one probe update (one frame's probe pass) of BASELINE.json's headline configuration
C3: Minecraft cave scene, 32x16x32 probes x 256 rays (4 194 304 probe rays), max_bounces = 8,
shipped light table, REF mode (the reference's live behaviour), ray jitter seed 1.  Inputs (the
48 B/ray ProbeRay buffer, the baked scene) are resident in HBM before the timed region.

  python bench.py --gpus 1 --steps 20 --warmup 5
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

N > 1: the probe grid is sharded by z-slab (cz/N layers per rank); every step each rank traces its
slab and the blended textures are exchanged with one RCCL all-gather per texture over xGMI, issued by the
engine itself (ddgi_exchange: double-buffered pairs + a communication stream inside libddgi_probe.so;
torch.distributed only hands the 128-byte RCCL id around and provides the barrier / max-over-ranks).
Total work is fixed, so `scaling` is "strong".

Prints ONE JSON line on rank 0 (contract in the task statement) carrying `roofline` for the
dominant kernel (k_probe_trace_aq, the queue-driven wavefront tracer) and, at N = 1, `cpu_baseline`
(the CPU oracle timed on the box's host cores over a bounded sample of the same workload; its texels
are compared with the GPU's byte for byte, and — `literal_within_1_255` — within the stated tolerance
against the oracle's LITERAL arithmetic) and `fast_march` (the opt-in tolerance-mode march timed on the
same workload next to the exact headline number, with its texel agreement against both arithmetics).

  --workload c5 --mode ddgi   SURVEY.md section 8d's S-Dyn: 128x64x128 probes x 256 rays, the reference's
                              4-light cave table animated by update_lights, hysteresis 0.9, time = 2 frame;
                              frames 8.. are the timed steady state (--warmup defaults to 8 there)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WORKLOAD = {
    "name": "c3_cave_32x16x32_probes_x256_rays_ref",
    "counts": (32, 16, 32),
    "side": 2,
    "s": 16,
    "tile": (16, 16),
    "origin": (1.4, 0.0, 1.0),
    "scene": 0,
    "max_bounces": 8,
    "seed": 1,
}
# BASELINE.json configs[3] (the 8-GPU shard configuration), runnable on one GPU with --workload c4: 131 072 probes x 512 rays
# (a 32 x 16 ray tile, ddgi_set_ray_tile) = 67 108 864 probe rays, 3.2 GB of ProbeRay records resident in HBM
WORKLOAD_C4 = dict(WORKLOAD, name="c4_cave_64x32x64_probes_x512_rays_ref", counts=(64, 32, 64), side=1, tile=(32, 16))
# BASELINE.json configs[4] = SURVEY.md 8d S-Dyn: 1 048 576 probes x 256 rays = 268 435 456 probe rays per frame, DDGI mode, 4 animated lights
WORKLOAD_C5 = dict(WORKLOAD, name="c5_cave_128x64x128_probes_x256_rays_4_dynamic_lights", counts=(128, 64, 128), side=1,
                   lights=[(20.0, (1.0, 1.0, 1.0), (4, 17.5, 8.5)), (10.0, (1.0, 0.5, 0.1), (0, 2, 0)),      # assets/shaders/structs.glsl:65-68
                           (10.0, (0.1, 1.1, 1.0), (5, 0, 0)), (10.0, (1.1, 0.0, 1.1), (0, 5, 0))])
WORKLOADS = {"c3": WORKLOAD, "c4": WORKLOAD_C4, "c5": WORKLOAD_C5}
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
ALGO_BYTES_PER_RAY = 56        # 48 B ProbeRay read + two 4 B rgba8 texel writes


def _issue_from_profiles():
    """VALU occupancy of the trace kernel from the committed rocprofv3 --pmc passes
    (profiles/*_issue.txt, tools/pmc_icache.sh): the bound the kernel actually sits at."""
    import glob
    import re

    out = {}
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_issue.txt"))):
        try:
            txt = open(path).read()
            busy = re.search(r"VALU busy = .* = ([0-9.]+)", txt)
            lanes = re.search(r"mean active lanes per VALU instruction = .* = ([0-9.]+)", txt)
            insts = re.search(r"SQ_INSTS_VALU\s+([0-9.e+]+)", txt)
            if busy and lanes:
                out = {"valu_busy": float(busy.group(1)), "valu_lane_use": float(lanes.group(1)),
                       "valu_wave_instructions_per_launch": float(insts.group(1)) if insts else None,
                       "replayed_from": "profiles/" + os.path.basename(path)}
        except Exception:
            pass
    return out or None


def _traffic_from_profiles(workload=None):
    """Per-launch HBM bytes of the trace kernel from the committed rocprofv3 --pmc passes (profiles/*_traffic.json,
    written by tools/pmc_traffic.py) and the file they come from; (None, None) if not collected.  REPLAYED, not measured
    by this run: counters need their own rocprofv3 passes."""
    import glob

    best, src = None, None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*traffic*.json"))):
        try:
            with open(path) as fh:
                d = json.load(fh)
            if d.get("workload") == (workload or WORKLOAD["name"]) and d.get("kernel", "").startswith("k_probe_trace") and d.get("hbm_bytes_per_launch"):
                best, src = d["hbm_bytes_per_launch"], "profiles/" + os.path.basename(path)
        except Exception:
            pass
    return best, src


def _sample_traffic_from_profiles(workload, mode, kernel):
    """Counter-derived bytes per batch of a cage-sample kernel (profiles/*sample_traffic*.json: FETCH_SIZE x 2 — the gfx950 correction of
    MI355X_MICROARCH.md — + WRITE_SIZE of a rocprofv3 --pmc pass over tools/sample_bench.py) -> (bytes, source) or (None, None).  REPLAYED."""
    import glob

    best, src = None, None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*sample_traffic*.json"))):
        try:
            with open(path) as fh:
                for d in json.load(fh):
                    if d.get("workload") == workload and d.get("mode") == mode and d.get("kernel") == kernel:
                        best, src = d["hbm_bytes_per_batch"], "profiles/" + os.path.basename(path)
        except Exception:
            pass
    return best, src


def _tile_view(raster, w, mask):
    """The tiles of the masked probes out of a reference-layout raster: [n_probes, ty, tx, 4]."""
    tx, ty = w["tile"]
    cxz = w["counts"][0] * w["counts"][2]
    return raster.reshape(w["counts"][1], ty, cxz, tx, 4).transpose(0, 2, 1, 3, 4)[mask]


def _tolerance(a, b):
    d = np.abs(a[..., :3].astype(np.int32) - b[..., :3].astype(np.int32))
    return {"within_1_255": float((d <= 1).mean()), "mean_abs_diff_255": float(d.mean()), "texels_differing": int((d.max(axis=-1) > 0).sum())}


def cpu_baseline(n_probes=96, gpu_albedo=None, w=WORKLOAD, rays=None, fast_albedo=None, sample_seconds=12.0):
    """The oracle (a CPU restatement of the reference's algorithm: procedural getBlockAt per march
    step, exactly what the reference's shader does) over a bounded, evenly spread sample of the
    workload's probes, all host threads.  The texels it computes are also compared, byte for byte, with
    the ones the GPU produced in the timed run (`parity_checked`); the same probes are then evaluated in the
    oracle's LITERAL arithmetic (one IEEE operation per GLSL operator, libm) for the stated tolerance
    (`literal_within_1_255`), and the fast march's texels (fast_albedo) are held against both."""
    from oracle import oracle_py as O

    O.set_arith(True)
    tx, ty = w["tile"]
    O.set_ray_tile(*((0, 0) if tx == ty == w["s"] else (tx, ty)))
    f = O.make_field(w["counts"], w["side"], w["s"], w["origin"])
    st = O.make_settings(w["scene"], w["max_bounces"])
    if rays is None:  # (the engine's own ray buffer when the caller passes it: the host generator's output, checked against the oracle's in tests/)
        rays = O.generate_probe_rays(f, O.new_rand_state(w["seed"]))
    else:
        rays = rays.view(O.RAY_DTYPE)
    total = w["counts"][0] * w["counts"][1] * w["counts"][2]
    # calibrate on a small spread sample, then size the timed sample for ~12 s of wall time
    # (bounded by the whole grid)
    calib = np.linspace(0, total - 1, max(n_probes, 2 * O.num_threads())).astype(np.int32)
    t0 = time.perf_counter()
    O.probe_update_probes(f, st, rays, calib)
    rate = len(calib) / max(time.perf_counter() - t0, 1e-6)
    n = int(min(total, max(len(calib), rate * sample_seconds)))
    probes = np.linspace(0, total - 1, n).astype(np.int32)
    t0 = time.perf_counter()
    want = O.probe_update_probes(f, st, rays, probes)
    dt = time.perf_counter() - t0
    nrays = len(probes) * tx * ty
    out = {
        "value": nrays / dt,
        "unit": "rays/s",
        "cores": O.num_threads(),
        "kind": "port",
        "sample": f"{len(probes)} of {total} probes evenly spread over the grid ({nrays} rays), {dt:.1f} s",
    }
    fast = None
    if gpu_albedo is not None:
        # the probes of the sample as tile masks of the reference raster (tile of probe p at ((p mod cx*cz)*s, (p div cx*cz)*s))
        cxz = w["counts"][0] * w["counts"][2]
        mask = np.zeros((w["counts"][1], cxz), dtype=bool)
        mask[np.unique(probes) // cxz, np.unique(probes) % cxz] = True
        tiles_gpu, tiles_cpu = _tile_view(gpu_albedo, w, mask), _tile_view(want, w, mask)
        differ = int((tiles_gpu != tiles_cpu).any(axis=-1).sum())
        n_tex = int(mask.sum()) * tx * ty
        cfg = w["name"][:2]
        scope = f"{cfg} full grid" if int(mask.sum()) == total else f"{int(mask.sum())} of {total} probes of {cfg}"
        out["parity_checked"] = f"{scope}, {n_tex - differ} of {n_tex} texels equal (HIP vs oracle, rgba8 bytes)"
        out["parity_texels_differing"] = differ
        # the same probes in LITERAL arithmetic: the stated tolerance (|d| <= 1/255 on >= 99.9 % of the rgb channels, mean < 0.05/255)
        O.set_arith(False)
        t0 = time.perf_counter()
        lit = O.probe_update_probes(f, st, rays, probes)
        O.set_arith(True)
        tiles_lit = _tile_view(lit, w, mask)
        tol = _tolerance(tiles_gpu, tiles_lit)
        out["literal_within_1_255"] = tol["within_1_255"]
        out["literal"] = dict(tol, scope=scope, seconds=round(time.perf_counter() - t0, 1),
                              note="HIP (exact march, PINNED arithmetic) vs the oracle's LITERAL arithmetic: IEEE operations in GLSL source order, libm sin/cos/acos")
        if fast_albedo is not None:
            tiles_fast = _tile_view(fast_albedo, w, mask)
            fast = {"vs_pinned_oracle": _tolerance(tiles_fast, tiles_cpu), "vs_literal_oracle": _tolerance(tiles_fast, tiles_lit), "scope": scope}
    O.set_ray_tile(0, 0)
    return out, fast


class _c_stdout_to_stderr:
    """RCCL prints its version banner to the C-level stdout when a communicator is created; bench.py's stdout
    must carry exactly one JSON line, so file descriptor 1 points at stderr while communicators are made (and the C stdio
    buffer, which a pipe makes block-buffered, is flushed before fd 1 is restored)."""

    def __enter__(self):
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        import ctypes

        ctypes.CDLL(None).fflush(None)  # RCCL printf()s into the C stdio buffer: empty it while fd 1 still points at stderr
        os.dup2(self._saved, 1)
        os.close(self._saved)


def _call_bounded(fn, seconds):
    """fn() on a helper thread, waited for at most `seconds`: (True, result) | (False, exception or "timeout").  A call into RCCL
    that never returns (a peer that died during the rendezvous) must not hang the benchmark: the caller falls back or reports."""
    import threading

    box = {}

    def run():
        try:
            box["r"] = fn()
        except BaseException as exc:  # noqa: BLE001 - reported to the caller
            box["e"] = exc

    t = threading.Thread(target=run, daemon=True)
    t.start()
    t.join(seconds)
    if t.is_alive():
        return False, "timeout after %d s" % seconds
    if "e" in box:
        return False, box["e"]
    return True, box.get("r")


# CUs x SIMDs x lanes per SIMD-cycle x clock: CDNA4's SIMDs are 32 lanes wide — a wave64 VALU instruction issues over 2 cycles
# (MI355X_MICROARCH.md "Wave scheduling"; tools/microbench/valu_latency.hip measures 2.5 cycles per v_fma_f32 at 8 waves per SIMD,
# 3.1 at the trace kernel's 4 waves of dependent code) — 7.86e13 lane-ops/s.  (Rounds 1-3 priced the kernel against a 16-lane SIMD.)
VALU_LANE_PEAK = 256 * 4 * 32 * 2.4e9


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=None, help="default 5 (c5: 8 — SURVEY.md 8d times frames 8.. of S-Dyn)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-probes", type=int, default=96)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="c3",
                    help="c3 (default): the configuration the metric is quoted on; c4: BASELINE's 8-GPU shard configuration "
                         "(64x32x64 probes x 512 rays) on however many GPUs are given; c5 (with --mode ddgi): S-Dyn, 128x64x128 probes "
                         "x 256 rays, 4 animated lights + hysteresis")
    ap.add_argument("--no-fast-march", action="store_true", help="skip the extra timed run of the opt-in tolerance-mode march (N = 1, REF)")
    ap.add_argument("--no-extras", action="store_true", help="skip the sampler block, the frames_in_flight 1 / 2 / 4 runs and the set-up timings (profiling runs)")
    ap.add_argument("--mode", choices=["ref", "ddgi"], default="ref",
                    help="ref (default): the reference's live behaviour, the headline metric; ddgi: in-kernel Fibonacci rays + "
                         "octahedral irradiance/depth blend with hysteresis (trace + blend per step)")
    ap.add_argument("--exchange", choices=["auto", "rccl", "p2p"], default="auto",
                    help="N > 1: how the ranks' slabs are exchanged — p2p: every rank pushes its slab into its peers' textures "
                         "(ddgi_exchange_p2p_*, IPC-mapped buffers: copies between GPUs, which overlap the trace kernel's persistent workgroups); "
                         "rccl: one in-place ncclAllGather per texture; auto (default): BOTH are brought up and timed on a short run, the timed "
                         "region runs under the faster (multi_gpu.by_transport reports both).  A transport that cannot be brought up (an error, "
                         "or for RCCL no answer within --rccl-timeout seconds) is reported by name with the reason (config.exchange_fallback)")
    ap.add_argument("--rccl-timeout", type=float, default=90.0)
    ap.add_argument("--wait-timeout", type=float, default=30.0,
                    help="N > 1: seconds any wait for another rank may take inside the library (tuning \"wait_timeout_ms\"; the library's own default is 10 s — "
                         "a first collective over a fresh communicator may take longer than a frame ever does) before it returns DDGI_ERR_TIMEOUT naming the rank that is behind")
    ap.add_argument("--p2p-timeout", type=float, default=120.0, help="seconds the peer-to-peer transport gets to map its peers' buffers")
    ap.add_argument("--frames-in-flight", type=int, default=None, help="tuning \"frames_in_flight\" (default: the library's, 8; the reference's host runs MAX_FRAMES_IN_FLIGHT = 2 ahead)")
    args = ap.parse_args()
    if args.workload == "c5" and args.mode != "ddgi":
        raise SystemExit("--workload c5 is S-Dyn (4 dynamic lights + temporal hysteresis): run it with --mode ddgi")
    if args.warmup is None:
        args.warmup = 8 if args.workload == "c5" else 5

    if os.environ.get("DDGI_BENCH_WATCHDOG"):   # (diagnostics: every rank's Python stack on stderr every so many seconds — where a run that does not end is)
        import faulthandler

        faulthandler.dump_traceback_later(int(os.environ["DDGI_BENCH_WATCHDOG"]), repeat=True, file=sys.stderr)

    import torch
    import torch.distributed as dist

    import ddgi_amd

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product has no CPU path")
    # DDGI_BENCH_ONE_GPU=1: every rank uses device 0 (a functional run of the N > 1 path on a one-GPU box — not a measurement:
    # the ranks share the chip; RCCL refuses two ranks on one device, so this needs --exchange p2p)
    one_gpu = os.environ.get("DDGI_BENCH_ONE_GPU") == "1"
    if one_gpu:
        local_rank = 0
        if world > 1 and args.exchange == "rccl":
            raise SystemExit("DDGI_BENCH_ONE_GPU=1 needs --exchange p2p or auto (RCCL refuses two ranks on one device)")
    torch.cuda.set_device(local_rank)
    # DDGI_BENCH_FORCE_DIST=1: run the N > 1 code path (RCCL group, pipelined exchange) with a single rank — a smoke test of
    # that path on a one-GPU box
    sharded = world > 1 or os.environ.get("DDGI_BENCH_FORCE_DIST") == "1"
    if sharded:
        # torch.distributed is the CONTROL plane only — the 128-byte RCCL id / the p2p addresses, the barrier, the max over ranks —
        # and runs over gloo: the data path's RCCL communicator is the engine's own (ddgi_comm_create), so a problem with RCCL
        # cannot take the rendezvous down with it, and the fallback to the peer-to-peer transport can be agreed on
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        with _c_stdout_to_stderr():             # (gloo announces its connections on the C-level stdout; the line below must be the only one there)
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
            dist.barrier()

    def all_ok(flag):
        if not sharded:
            return bool(flag)
        t = torch.tensor([1 if flag else 0], dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item())

    setup_ms = {}
    w = WORKLOADS[args.workload]
    field = ddgi_amd.make_field(w["counts"], w["side"], w["s"], w["origin"])
    settings = ddgi_amd.make_settings(w["scene"], w["max_bounces"])
    t_setup = time.perf_counter()
    eng = ddgi_amd.ProbeEngine(field, settings, device=local_rank, rank=rank, world=world)
    if w["tile"] != (w["s"], w["s"]):
        eng.set_ray_tile(*w["tile"])
    stream = torch.cuda.current_stream()
    eng.set_stream(stream.cuda_stream)          # kernels + collectives share torch's stream
    if args.frames_in_flight is not None:
        eng.set_tuning("frames_in_flight", args.frames_in_flight)
    if sharded:
        eng.set_tuning("wait_timeout_ms", int(args.wait_timeout * 1000))   # every cross-rank wait of the timed loop is bounded (DDGI_ERR_TIMEOUT, not a hang)
    ddgi_mode = args.mode == "ddgi"
    setup_ms["create_handle_and_textures"] = (time.perf_counter() - t_setup) * 1e3
    t_setup = time.perf_counter()
    if ddgi_mode:
        eng.set_mode(ddgi_amd.MODE_DDGI)        # rays are generated in the kernel; tiles start zeroed
    else:
        eng.generate_probe_rays(seed=w["seed"])  # ray buffer resident in HBM from here on
        setup_ms["generate_and_upload_probe_rays"] = (time.perf_counter() - t_setup) * 1e3
    if w.get("lights"):
        eng.set_lights(w["scene"], np.array(w["lights"], dtype=ddgi_amd.LIGHT_DTYPE))  # animated per update from RenderSettings::time

    comm = None
    exchanging = False
    transport = "none"
    fallback = None
    bring_up_s = {}   # seconds per stage of a transport's (latest) bring-up on this rank: multi_gpu.bring_up_s

    def try_rccl():
        # the engine issues the RCCL all-gather itself (include/ddgi_probe.h: ddgi_exchange_*): rank 0 makes the
        # 128-byte RCCL id, torch.distributed carries it to the other ranks, every rank joins the communicator.
        # Bounded: an error or no answer within --rccl-timeout on ANY rank makes every rank give the transport up.
        t_up = time.perf_counter()

        def bring_up():
            with _c_stdout_to_stderr():
                c = ddgi_amd.comm_create(ids[0], world, rank, local_rank)
            eng.exchange_init(c, pipelined=True)   # pipelined: the exchange of an update overlaps the kernels of the next ones
            return c

        with _c_stdout_to_stderr():
            ok, res = _call_bounded(ddgi_amd.comm_unique_id, args.rccl_timeout) if rank == 0 else (True, None)
        ids = [res if (rank == 0 and ok) else None]
        dist.broadcast_object_list(ids, src=0)
        ok = ids[0] is not None
        if ok:
            ok, res = _call_bounded(bring_up, args.rccl_timeout)
        bring_up_s["rccl"] = {"unique_id_comm_init_rank_attach": time.perf_counter() - t_up}
        if all_ok(ok):
            return True, res, None
        if ok:
            eng.exchange_init(None)
        return False, None, "RCCL transport not available (%s on rank %d%s)" % ("ok" if ok else str(res)[:200], rank, "" if ok else ", this rank")

    def try_p2p():
        # peer-to-peer: every rank publishes its buffers (512 bytes), torch.distributed carries the addresses around, every rank
        # maps its peers' textures (IPC) and pushes its slab into them.  An error on ANY rank makes every rank give the transport up;
        # in REF mode (updates are idempotent) one exchanged update is checked before the transport is trusted: every rank must
        # hold the same gathered field (its bytes against the unsharded engine and the oracle are checked after the timed region).
        import hashlib

        why = None
        t_up = time.perf_counter()
        stages = bring_up_s.setdefault("p2p", {})
        # The largest buffer a peer would have to map.  On the stack measured a buffer of 2 GiB or more is not handed to another process reliably (round 6:
        # hipIpcOpenMemHandle that does not come back, bisected on ring sizes, tools/hunt/r06_probe4.sh; the library gives the call a deadline, but the helper thread it
        # leaves behind stands inside the HIP runtime).  A grid whose RING reaches that size (C5: one pair of depth tiles is 2^31 bytes) publishes LANDING ZONES
        # instead (csrc/ddgi_exchange.cpp: (world - 1) / world of one pair per texture and parity) — only a grid whose zones would reach it too goes straight to RCCL.
        per_pair = eng.num_probes * 2048 if ddgi_mode else eng.num_rays * 4
        zone = per_pair // world * (world - 1)
        if zone >= (2 << 30) and os.environ.get("DDGI_BENCH_P2P_ANY_SIZE") != "1":
            dist.barrier()
            return False, "peer-to-peer transport not attempted: a peer would map a landing zone of %.1f GB, and mappings of 2 GiB or more do not come back on this stack (profiles/r06_p2p_ring_size_bisection.txt)" % (zone / 1e9)
        try:
            if os.environ.get("DDGI_BENCH_FAIL_P2P") == "1":   # (fault injection: exercises the fallback's control flow)
                raise RuntimeError("DDGI_BENCH_FAIL_P2P")
            mine = eng.exchange_p2p_export(pipelined=True)
        except Exception as e:                      # noqa: BLE001 (whatever the binding raises: the transport is not available)
            mine, why = None, "export: " + str(e)[:160]
        stages["export_receive_ring"] = time.perf_counter() - t_up
        t_up = time.perf_counter()
        everyone = [None] * world
        dist.all_gather_object(everyone, mine)
        stages["addresses_over_gloo"] = time.perf_counter() - t_up
        t_up = time.perf_counter()
        ok = all(a is not None for a in everyone)
        if ok:
            # (bounded like RCCL's bring-up: mapping the peers' rings is a driver call per buffer — 4 ranks' rings of a C5-sized grid,
            # 12.8 GB each, did not come back within seven minutes on the one-GPU test box)
            # ONE RANK AT A TIME.  Round 6 placed the bring-up that never came back (C5's rings, four ranks): every rank stood in hipIpcOpenMemHandle ->
            # hsa_amd_ipc_memory_attach -> recvmsg, waiting for the EXPORTING process to hand the buffer's dmabuf over its socket — while that process
            # stood in the same call towards somebody else (profiles/r06_c5_bring_up_backtrace.txt).  Mapping is milliseconds per peer
            # (profiles/r06_ipc_open_cost.txt); taking turns costs nothing and no two ranks ever attach to each other at the same time.
            done, res = True, None
            for turn in range(world):
                if turn == rank:
                    done, res = _call_bounded(lambda: eng.exchange_p2p_init(everyone), args.p2p_timeout)
                dist.barrier()
            if not done:
                ok, why = False, "init: " + str(res)[:160]
        stages["map_peers"] = time.perf_counter() - t_up
        t_up = time.perf_counter()
        attached = ok
        ok = all_ok(ok)
        if ok and not ddgi_mode:
            digest = None
            try:
                for _ in range(2):
                    eng.probe_update()
                    eng.exchange()
                eng.exchange_finish()
                torch.cuda.synchronize()
                eng.synchronize()
                digest = hashlib.sha1(np.ascontiguousarray(eng.read_textures()[0]).tobytes()).hexdigest()
            except Exception as e:                  # noqa: BLE001
                why = "first exchange: " + str(e)[:160]
            digests = [None] * world
            dist.all_gather_object(digests, digest)
            ok = digest is not None and all(d == digests[0] for d in digests)
            if not ok and why is None:
                why = "the ranks' gathered fields differ after the first exchange"
            ok = all_ok(ok)
            stages["first_two_exchanges_checked"] = time.perf_counter() - t_up
        if ok:
            return True, None
        if attached:
            try:
                eng.exchange_init(None)
            except Exception:                       # noqa: BLE001
                pass
        return False, "peer-to-peer transport not available (%s, rank %d)" % (why or "another rank gave up", rank)

    def attach(cand):
        """Brings transport `cand` up on every rank (collective).  -> (ok, why)"""
        nonlocal comm, exchanging, transport
        if cand == "rccl":
            ok, c, why = try_rccl()
            if ok:
                comm = c
        else:
            ok, why = try_p2p()
        if ok:
            exchanging, transport = True, cand
        return ok, why

    def detach():
        """Every rank lets go of its transport (collective): nothing in flight, peers' mappings closed, communicator destroyed."""
        nonlocal comm, exchanging, transport
        fence()
        if sharded:
            dist.barrier()                      # every rank has stopped pushing before any rank unmaps / frees
        eng.exchange_init(None)
        if comm is not None:
            with _c_stdout_to_stderr():
                ddgi_amd.comm_destroy(comm)
            comm = None
        exchanging, transport = False, "none"
        if sharded:
            dist.barrier()

    pinned_split = bool(os.environ.get("DDGI_AQ_MARCH"))  # (profiling runs pin the split so that every launch is the steady-state kernel)
    frame_time = [0.0]
    frames_issued = [0]
    with_exchange = [True]

    def step():
        if ddgi_mode:
            frame_time[0] += 2.0                # RVPT::update: render_settings.time += 2 (rvpt.cpp:281)
            settings.time = frame_time[0]
            eng.probe_update(settings)
        else:
            eng.probe_update()
        frames_issued[0] += 1
        if exchanging and with_exchange[0]:
            eng.exchange()

    def fence():
        if exchanging:
            eng.exchange_finish()               # the stream waits for every exchange issued so far
        if sharded:
            dist.barrier()
        torch.cuda.synchronize()
        eng.synchronize()                       # (the stream is idle: only tells the handle so — the next update starts a group of frames in flight)

    def timed(steps, warmup):
        for _ in range(warmup):
            step()
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        fence()
        dt = time.perf_counter() - t0
        if sharded:
            t = torch.tensor([dt], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    # the first update of a handle: scene bake upload, the memoised noise lattice (host build + 70 MB upload), the light-feeler
    # classes (k_light_visibility) — all outside the timed region, and reported
    t_setup = time.perf_counter()
    step()
    fence()
    setup_ms["first_update_scene_noise_tables_light_classes"] = (time.perf_counter() - t_setup) * 1e3
    if not pinned_split:
        t_setup = time.perf_counter()
        eng.tune()                              # the march/event wave split of this configuration, measured once (blocks; outside the timed region)
        setup_ms["ddgi_tune"] = (time.perf_counter() - t_setup) * 1e3

    # ---- N > 1: which transport, decided by measurement ----------------------------------------------------------------------
    # Two transports serve the same all-gather (csrc/ddgi_exchange.cpp): peer-to-peer pushes into IPC-mapped textures (copies between
    # GPUs, no CU needed) and RCCL's ncclAllGather (kernels: the queue kernel's persistent workgroups fill every CU, so a collective
    # kernel finds a CU only between two launches — unless a few CUs are left free for it, tuning "reserve_cus").  `--exchange auto`
    # brings BOTH up, one after the other, times the same short run of updates + exchanges under each (at 0 / 2 / 4 reserved CUs;
    # the maximum over the ranks), and runs the timed region under the faster; the line reports both (multi_gpu.by_transport),
    # a transport that could not be brought up by name with the reason.
    by_transport = {}
    reserve_sweep = None
    if sharded:
        order = {"auto": ["p2p", "rccl"], "rccl": ["rccl"], "p2p": ["p2p"]}[args.exchange]
        if world == 1:
            order = ["rccl"]                        # (DDGI_BENCH_FORCE_DIST: the one-rank RCCL group)
        sweep_reserve = world > 1 and not ddgi_mode and os.environ.get("DDGI_RESERVE_CUS") is None
        for cand in order:
            ok, why = attach(cand)
            if not ok:
                by_transport[cand] = {"available": False, "reason": why, "bring_up_s": {k: round(v, 4) for k, v in bring_up_s.get(cand, {}).items()}}
                continue
            info = {"available": True, "ranks_in_communicator": eng.exchange_ranks() if cand == "rccl" else world,
                    "bring_up_s": {k: round(v, 4) for k, v in bring_up_s.get(cand, {}).items()}}
            if cand == "p2p":
                # what a peer maps of this rank: its rings (their pushes land where consumers read), or — a ring of 2 GiB and more — landing zones
                info.update(exported_mb_per_rank=eng.get_tuning("p2p_exported_mb"), landing_zones=eng.get_tuning("p2p_landing_zones") > 0)
            if world > 1:
                sweep = {}
                r0 = eng.get_tuning("reserve_cus")
                for r in ((0, 2, 4) if sweep_reserve else (r0,)):
                    eng.set_tuning("reserve_cus", r)
                    sweep[str(r)] = timed(16, 4) / 16 * 1e3
                best = min(sweep, key=lambda k: sweep[k])
                if "0" in sweep and sweep["0"] <= sweep[best] * 1.01:
                    best = "0"                      # (within a percent: the full machine)
                info.update(reserve_cus_ms_per_step=sweep, reserve_cus=int(best), ms_per_step=sweep[best])
                eng.set_tuning("reserve_cus", r0)
            by_transport[cand] = info
            if len(order) > 1:
                detach()
        usable = [c for c in order if by_transport[c].get("available")]
        verdict = torch.tensor([len(usable)], dtype=torch.int32)
        dist.all_reduce(verdict, op=dist.ReduceOp.MIN)
        if not usable or verdict.item() == 0:
            raise SystemExit("no exchange transport could be brought up: " + json.dumps(by_transport))
        winner = min(usable, key=lambda c: by_transport[c].get("ms_per_step", 0.0))
        pick = [winner]
        dist.broadcast_object_list(pick, src=0)     # (rank 0's clock decides; the times are maxima over the ranks already)
        winner = pick[0]
        if len(order) > 1:
            ok, why = attach(winner)
            if not ok:
                raise SystemExit("the chosen transport (%s) did not come up a second time: %s" % (winner, why))
        reserve_sweep = by_transport[winner].get("reserve_cus_ms_per_step")
        if reserve_sweep is not None:
            reserve_sweep = dict(reserve_sweep, chosen=by_transport[winner]["reserve_cus"])
            eng.set_tuning("reserve_cus", by_transport[winner]["reserve_cus"])
        refused = [c + ": " + str(by_transport[c].get("reason")) for c in order if not by_transport[c].get("available")]
        if refused:
            fallback = "; ".join(refused) + ": ran with " + winner

    elapsed = timed(args.steps, max(0, args.warmup - 1))   # (the first update above is the first warm-up step)

    # kernel durations of the timed steps: HIP events recorded on the launch stream by the engine.  With frames in flight a
    # continued update's own launch is empty (its predecessor's launch traced its rays): the MEAN over the steps is the time
    # per update, the individual launches are not
    trace_ms, blend_ms = eng.update_history_ms(min(args.steps, 64))
    kernel_ms = float(np.mean(trace_ms)) if len(trace_ms) else float("nan")
    fif = eng.get_tuning("frames_in_flight")
    # the same steps once more WITHOUT those events (tuning "timing" 0: what a caller who does not ask for per-update times gets);
    # reported beside the line's numbers, not instead of them
    untimed_ms = None
    if not sharded and not ddgi_mode:   # (DDGI mode: the lights move on with every step — later frames are not the same work)
        eng.set_tuning("timing", 0)
        untimed_ms = timed(args.steps, 1) / args.steps * 1e3
        eng.set_tuning("timing", 1)

    total_rays = eng.num_rays
    local_rays = total_rays // world
    ms_per_step = elapsed / args.steps * 1e3
    # REF: 48 B ProbeRay in + two 4 B texels out per ray.  DDGI: rays are generated in the kernel and the
    # ray records between trace and blend are an intermediate that does not count: 3072 B of tiles per probe (old tiles in +
    # new tiles out of the 8x8 rgba16f-equivalent irradiance and 16x16 rg16f-equivalent depth tiles), over trace + blend.
    probes_local = eng.num_probes // world
    bms = float(np.mean(blend_ms)) if len(blend_ms) else 0.0
    if ddgi_mode:
        algo_bytes = 3072 * probes_local
        kernel_ms = kernel_ms + bms
    else:
        algo_bytes = ALGO_BYTES_PER_RAY * local_rays
    achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9
    traffic, traffic_src = (None, None) if (ddgi_mode or world > 1) else _traffic_from_profiles(w["name"])
    issue = None if (ddgi_mode or args.workload != "c3" or world > 1) else _issue_from_profiles()
    valu = None
    if issue and issue.get("valu_wave_instructions_per_launch"):
        lane_ops = issue["valu_wave_instructions_per_launch"] * 64.0 * issue["valu_lane_use"]
        valu = {"useful_lane_ops_per_s": lane_ops / (kernel_ms * 1e-3), "lane_peak_per_s": VALU_LANE_PEAK,
                "frac_of_lane_peak": lane_ops / (kernel_ms * 1e-3) / VALU_LANE_PEAK, "lane_use": issue["valu_lane_use"], "valu_busy": issue["valu_busy"],
                "valu_issue_cycles_frac": issue["valu_wave_instructions_per_launch"] * 2.0 / (1024 * 2.4e9 * kernel_ms * 1e-3),
                "valu_issue_frac_of_measured_peak": issue["valu_wave_instructions_per_launch"] * 2.5 / (1024 * 2.4e9 * kernel_ms * 1e-3),
                "note": "instruction count and lane use REPLAYED from " + issue["replayed_from"] + " (the build those counters were taken on), time measured by this run; "
                        "lane peak = 256 CUs x 4 SIMDs x 32 lanes x 2.4 GHz (a wave64 VALU instruction issues over 2 cycles on CDNA4); valu_issue_cycles_frac = the share of "
                        "the SIMDs' cycles in which a VALU instruction issues at 2 cycles each; valu_issue_frac_of_measured_peak prices an instruction at the 2.5 cycles per "
                        "SIMD a stream of independent v_fma_f32 from 8 waves reaches on this chip (profiles/r04_exec_mask_rate.txt, r04_valu_latency.txt), before "
                        "the quarter-rate and packed instructions in the mix are counted: the kernel's time follows its instruction count — wave splits, "
                        "priorities, workgroup shapes and more ILP in the march waves all leave the count and the time where they are (docs/LAB_NOTES.md, Round 4)"}
    out = {
        "metric": "probe_rays_per_sec",
        "value": total_rays / (elapsed / args.steps),
        "unit": "rays/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "ms_per_step_at_reference_frames_in_flight": ms_per_step if fif == 2 else None,   # (filled in below at N = 1: the same loop with "frames_in_flight" 2)
        "value_at_reference_frames_in_flight": total_rays / (elapsed / args.steps) if fif == 2 else None,
        "ms_per_step_without_timing_events": untimed_ms,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": w["name"],
            "probes": list(w["counts"]),
            "rays_per_probe": w["tile"][0] * w["tile"][1],
            "probe_rays": total_rays,
            "scene": "minecraft_cave",
            "max_bounces": w["max_bounces"],
            "mode": "DDGI" if ddgi_mode else "REF",
            "frames_in_flight": fif,
            "parallelism": f"zslab{world}" + ((f"+allgather_{transport}" + ("_all_ranks_on_one_gpu" if one_gpu else "")) if world > 1 else ""),
        },
        "roofline": {
            "kernel": {"lane": "k_probe_trace_ref", "rounds": "k_probe_trace_wf"}.get(os.environ.get("DDGI_TRACE_KERNEL", ""), "k_probe_trace_aq"),
            "bound": "hbm",          # the roofline achieved / peak / frac are priced against (the contract's: "hbm" | "mfma")
            "limited_by": "valu",    # what the kernel really sits at (below)
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic,
            "traffic_replayed_from": traffic_src,
            "algorithmic_bytes_per_launch": algo_bytes,
            "kernel_ms": kernel_ms,
            "valu": valu,
            "issue": issue,
            "note": "achieved / peak / frac are the HBM figures the task defines (algorithmic bytes per update / mean launch duration of the timed updates, HIP events on the launch "
                    "stream; with frames in flight a launch traces up to `frames_in_flight` updates and the continued updates' own launches are empty — the mean is per update); "
                    "`limited_by` says what the kernel really sits at: its waves' dependent VALU instruction streams (voxel steps + hit shading, DESIGN.md section 4), `valu` prices it against the lane peak. "
                    "`traffic`, `issue` and the instruction count inside `valu` are REPLAYED from the committed rocprofv3 --pmc passes named beside them (counters need their own passes)",
        },
    }
    out["tuning"] = {"march_waves": eng.get_tuning("march_waves_measured"), "frames_in_flight": fif,
                     "note": "march_waves: waves of a 16-wave workgroup that march (the rest shade), measured by ddgi_tune() before the warm-up; frames_in_flight: the most updates one launch "
                             "may work on (the reference's host runs MAX_FRAMES_IN_FLIGHT = 2 ahead, src/rvpt/rvpt.h:23; the timed loop submits its steps back to back): an update submitted while its "
                             "predecessor runs is continued by the predecessor's workgroups"}
    if fallback:
        out["config"]["exchange_fallback"] = fallback
    if ddgi_mode:
        # the blend kernels on their own: what THEY must move is the ray records the trace left (20 B per ray: r, g, b, d, d*d;
        # an intermediate of the pass, so not part of `roofline`) + the f32 tiles in and out (6144 B per probe)
        bbytes = 20 * local_rays + 6144 * probes_local
        out["config"]["workload"] = w["name"].replace("_ref", "_ddgi")
        out["roofline"]["kernel"] += "+k_blend_weights+k_probe_blend"
        out["blend"] = {"kernel": "k_blend_weights+k_probe_blend", "kernel_ms": bms, "kernel_io_bytes_per_launch": bbytes,
                        "achieved_GBps": bbytes / (bms * 1e-3) / 1e9, "frac_of_hbm_peak": bbytes / (bms * 1e-3) / 1e9 / HBM_PEAK_GBS}
    if w.get("lights"):
        out["config"]["lights"] = f"{len(w['lights'])} (assets/shaders/structs.glsl:65-68), animated by update_lights(time), time += 2 per frame"
        out["config"]["hysteresis"] = 0.9
        out["config"]["frames_timed"] = f"{args.warmup}..{args.warmup + args.steps - 1}"

    # ---- N > 1: is the gathered field right, and where did the time go ---------------------------------------------------
    if world > 1:
        # per-rank kernel time of the timed steps; the exposed cost of the exchange = the same steps without it
        per_rank = [None] * world
        dist.all_gather_object(per_rank, {"rank": rank, "device": local_rank, "kernel_ms": kernel_ms, "transport": eng.exchange_transport()[0]})
        fence()
        with_exchange[0] = False
        no_x = timed(args.steps, 2) / args.steps * 1e3
        with_exchange[0] = True
        timed(0, 2)                             # (two updates WITH their exchanges again: the field below is a gathered one)
        out["multi_gpu"] = {
            "transport": transport, "pipelined": True, "ranks_seen": sorted(r["rank"] for r in per_rank), "per_rank": per_rank,
            "ms_per_step_without_exchange": no_x, "exchange_ms_exposed": ms_per_step - no_x,
            "reserve_cus_ms_per_step": reserve_sweep,
            "by_transport": by_transport,
            "note": "exchange_ms_exposed = ms_per_step minus the same timed loop without ddgi_exchange: what the pipelined all-gather costs the critical path; "
                    "by_transport: both transports brought up one after the other and timed on the same 16 updates + exchanges (maximum over the ranks) at 0 / 2 / 4 "
                    "CUs left free for the exchange's kernels — the timed region ran under the faster (`transport`); one that could not be brought up says why",
        }
        gathered = eng.read_textures() if not ddgi_mode else eng.read_tiles()   # a consumer: waits for the latest exchange by itself
        if rank == 0:
            import hashlib

            # (a) the same frames on ONE unsharded handle on this rank's GPU: every byte of the gathered field must equal it
            solo = ddgi_amd.ProbeEngine(field, ddgi_amd.make_settings(w["scene"], w["max_bounces"]), device=local_rank)
            if w["tile"] != (w["s"], w["s"]):
                solo.set_ray_tile(*w["tile"])
            if ddgi_mode:
                solo.set_mode(ddgi_amd.MODE_DDGI)
                if w.get("lights"):
                    solo.set_lights(w["scene"], np.array(w["lights"], dtype=ddgi_amd.LIGHT_DTYPE))
                st1 = ddgi_amd.make_settings(w["scene"], w["max_bounces"])
                for k in range(frames_issued[0]):
                    st1.time = 2.0 * (k + 1)
                    solo.probe_update(st1)
                want = solo.read_tiles()
            else:
                solo.generate_probe_rays(seed=w["seed"])
                solo.probe_update()
                want = solo.read_textures()
            # (c) the like-for-like denominator of the speed-up: the SAME box, ONE GPU, one unsharded handle, the same frames in
            # flight and timing loop (the other ranks wait at the barrier below; their GPUs idle)
            solo.set_tuning("frames_in_flight", fif)
            if not pinned_split:
                solo.tune()
            st1 = ddgi_amd.make_settings(w["scene"], w["max_bounces"])
            solo_frame = [frames_issued[0]]

            def solo_run(n):
                for _ in range(n):
                    if ddgi_mode:
                        solo_frame[0] += 1
                        st1.time = 2.0 * solo_frame[0]
                        solo.probe_update(st1)
                    else:
                        solo.probe_update()
                solo.synchronize()

            solo_run(max(2, args.warmup))
            t0 = time.perf_counter()
            solo_run(args.steps)
            one_gpu_ms = (time.perf_counter() - t0) / args.steps * 1e3
            out["multi_gpu"]["one_gpu_same_box_ms_per_step"] = one_gpu_ms
            out["multi_gpu"]["speedup_vs_one_gpu"] = one_gpu_ms / ms_per_step
            out["multi_gpu"]["speedup_note"] = ("one unsharded handle on rank 0's GPU of this box, frames_in_flight %d like the ranks, %d updates back to back after %d of warm-up, wall clock; "
                                               "the other ranks' GPUs idle meanwhile%s" % (fif, args.steps, max(2, args.warmup), " (DDGI_BENCH_ONE_GPU: every rank shares that GPU — not a measurement of scaling)" if one_gpu else ""))
            solo.close()
            sha = [hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest() for a in gathered]
            sha_want = [hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest() for a in want]
            differing = int(sum(int((np.asarray(g) != np.asarray(x)).reshape(-1, np.asarray(g).shape[-1]).any(axis=-1).sum()) for g, x in zip(gathered, want)))
            out["multi_gpu"]["field_sha1"] = sha
            out["multi_gpu"]["field_equals_one_gpu_handle"] = sha == sha_want
            out["parity_texels_differing"] = differing
            out["parity_checked"] = "gathered field of %d ranks vs one unsharded handle on the same box: %s (sha-1 of every texture; %d texels differ)" % (
                world, "equal" if sha == sha_want else "DIFFERENT", differing)
            # (b) a spread sample of probes against the CPU oracle (REF mode; what cpu_baseline does with the whole grid at N = 1)
            if not ddgi_mode and not args.no_cpu_baseline:
                cb, _ = cpu_baseline(args.cpu_probes, gpu_albedo=gathered[0], w=w, rays=None if args.workload == "c3" else eng.get_probe_rays(), sample_seconds=2.0)
                out["multi_gpu"]["oracle_sample"] = {k: cb[k] for k in ("parity_checked", "parity_texels_differing", "cores", "sample") if k in cb}
                out["parity_texels_differing"] = differing + cb.get("parity_texels_differing", 0)
        dist.barrier()                          # (the other ranks keep their handles — the peers' mapped buffers — alive until rank 0 has read)

    exact_albedo = eng.read_textures()[0] if (rank == 0 and world == 1 and not ddgi_mode) else None
    extras = world == 1 and not args.no_extras

    def hbm_rate(mode, kernel, seconds):
        """What the batch pulls through the L2's memory side per second: counter bytes of a committed rocprofv3 pass (replayed) over THIS run's time.  Beside
        algorithmic_GBps — bytes the points use over the time, which the caches can serve above any memory's rate: no field here is a fraction of a peak."""
        b, src = _sample_traffic_from_profiles(w["name"], mode, kernel)
        return None if b is None else {"GBps": b / seconds / 1e9, "bytes_per_batch": b, "replayed_from": src,
                                       "note": "FETCH_SIZE x 2 (gfx950: 128-byte requests tallied at 64) + WRITE_SIZE of the sample kernel, per batch of 1.44 M scattered points; Infinity-Cache hits are counted"}
    if extras and not ddgi_mode:
        # ---- the cage sampler on the same field (north_star's third kernel; assets/shaders/intersection.glsl:1306-1409) ----
        n_pts = 1600 * 900                       # one frame of shading points of the reference's window (src/rvpt/main.cpp:40-41)
        rng = np.random.default_rng(0)
        half = np.array(w["counts"], dtype=np.float64) * w["side"] * 0.47
        pos = torch.from_numpy((rng.uniform(-1, 1, size=(n_pts, 3)) * half + np.array(w["origin"])).astype(np.float32)).cuda()
        nrm = torch.from_numpy(rng.normal(size=(n_pts, 3)).astype(np.float32)).cuda()
        rgb = torch.empty((n_pts, 3), dtype=torch.float32, device="cuda")
        cage = torch.empty((n_pts, 8), dtype=torch.int32, device="cuda")

        def sample_batches(k):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(k):
                eng.sample_device(pos.data_ptr(), nrm.data_ptr(), n_pts, rgb.data_ptr(), cage.data_ptr())
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / k

        sample_batches(1)                       # (the handle's first large batch also ALLOCATES the per-texel table: once per handle, not per update)
        eng.probe_update()                      # REF, static rays: the same texels again (Q18) — and the table is stale, as after every update
        first = sample_batches(1)               # rebuilds the per-texel table of sample_probe (k_sample_box_filter) for the new textures: what a host pays per update
        sample_batches(3)
        steady = sample_batches(20)
        setup_ms["sample_box_table_first_batch_after_an_update"] = (first - steady) * 1e3
        io_bytes, table_bytes = 24 + 12 + 32, 8 * 16   # position + normal in, rgb + 8 cage indices out; one 16-byte table entry per cage corner
        # ... which lie in 1.5^3 = 3.4 lines of 128 bytes on average (the table's 2x2x2 bricks of probes: ddgi_types.h DDGI_BOX_LAYOUT)
        sector_bytes = 3.375 * 128
        out["sample"] = {
            "kernel": "k_probe_sample_ref (+ k_sample_box_filter once per update)", "points": n_pts, "ms": steady * 1e3, "points_per_s": n_pts / steady,
            "first_batch_after_update_ms": first * 1e3, "bytes_per_point": io_bytes + table_bytes,
            "algorithmic_GBps": n_pts * (io_bytes + table_bytes) / steady / 1e9,
            "l2_sector_GBps": n_pts * (io_bytes + sector_bytes) / steady / 1e9,
            "hbm_GBps": hbm_rate("ref", "k_probe_sample_ref", steady),
            "inside_grid": float((cage[:, 0] >= 0).float().mean()),
            "note": "1.44 M shading points scattered over the grid after the timed updates; bytes_per_point = 68 B of point I/O + 8 table entries of 16 B "
                    "(algorithmic); l2_sector_GBps counts the 128-byte lines those scattered entries cost (3.4 per point with the table in 2x2x2 bricks of "
                    "probes, 4.5 in slab order: round 5, 10.5 -> 14.8 G points/s; measured FETCH_SIZE x 2 = 0.58 GB per batch, profiles/r05_i_sample_layout_ab.txt) — "
                    "the kernel is bound by the lines it pulls through L2, DESIGN.md section 4",
        }
        # the same points in cage-cell order — what a frame's pixels are (neighbouring pixels shade neighbouring cells)
        cell = np.floor((pos.cpu().numpy() - np.array(w["origin"], dtype=np.float32)) / w["side"]).astype(np.int64)
        order = torch.from_numpy(np.lexsort((cell[:, 0], cell[:, 1], cell[:, 2]))).cuda()
        pos, nrm = pos[order].contiguous(), nrm[order].contiguous()
        sample_batches(3)
        ordered = sample_batches(20)
        out["sample"]["cell_ordered"] = {"ms": ordered * 1e3, "points_per_s": n_pts / ordered, "algorithmic_GBps": n_pts * (io_bytes + table_bytes) / ordered / 1e9,
                                         "note": "the same 1.44 M points sorted by the grid cell they lie in"}
        del pos, nrm, rgb, cage, order
    if extras and ddgi_mode:
        # ---- DDGI mode's cage sampler (Chebyshev-weighted: intersection.glsl:1306-1409 with the dormant visibility term on) on the tiles the timed updates left ----
        n_pts = 1600 * 900
        rng = np.random.default_rng(0)
        half = np.array(w["counts"], dtype=np.float64) * w["side"] * 0.47
        pos = torch.from_numpy((rng.uniform(-1, 1, size=(n_pts, 3)) * half + np.array(w["origin"])).astype(np.float32)).cuda()
        nrm = torch.from_numpy(rng.normal(size=(n_pts, 3)).astype(np.float32)).cuda()
        rgb = torch.empty((n_pts, 3), dtype=torch.float32, device="cuda")
        cage = torch.empty((n_pts, 8), dtype=torch.int32, device="cuda")

        def ddgi_batches(k):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(k):
                eng.sample_device(pos.data_ptr(), nrm.data_ptr(), n_pts, rgb.data_ptr(), cage.data_ptr())
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / k

        ddgi_batches(3)
        steady = ddgi_batches(20)
        # per point: position + normal in, rgb + 8 cage indices out (68 B) + per cage corner 4 irradiance texels of 16 B and 4 depth texels of 8 B (bilinear)
        bpp = 24 + 12 + 32 + 8 * (4 * 16 + 4 * 8)
        out["sample"] = {"kernel": "k_sample_* (grouping by cage) + k_probe_sample_ddgi", "points": n_pts, "ms": steady * 1e3, "points_per_s": n_pts / steady,
                         "bytes_per_point": bpp, "algorithmic_GBps": n_pts * bpp / steady / 1e9, "hbm_GBps": hbm_rate("ddgi", "k_probe_sample_ddgi", steady),
                         "inside_grid": float((cage[:, 0] >= 0).float().mean()),
                         "note": "1.44 M shading points scattered over the grid, DDGI mode (irradiance + depth tiles, Chebyshev visibility); bytes_per_point is algorithmic: "
                                 "68 B of point I/O + 8 corners x (4 irradiance texels of 16 B + 4 depth texels of 8 B); the batch is grouped by cage first "
                                 "(three small kernels, a fifth of the time: profiles/r05_k_sample_kernels.txt)"}
        # the same points in cage-cell order — what a frame's pixels are
        cell = np.floor((pos.cpu().numpy() - np.array(w["origin"], dtype=np.float32)) / w["side"]).astype(np.int64)
        order = torch.from_numpy(np.lexsort((cell[:, 0], cell[:, 1], cell[:, 2]))).cuda()
        pos, nrm = pos[order].contiguous(), nrm[order].contiguous()
        ddgi_batches(3)
        ordered = ddgi_batches(20)
        out["sample"]["cell_ordered"] = {"ms": ordered * 1e3, "points_per_s": n_pts / ordered, "algorithmic_GBps": n_pts * bpp / ordered / 1e9,
                                         "note": "the same 1.44 M points sorted by the grid cell they lie in (the grouping kernels still run: the library does not know)"}
        eng.set_tuning("sample_group", 0)   # what a host that knows its points are coherent (a G-buffer in pixel order) asks for
        ddgi_batches(3)
        ungrouped = ddgi_batches(20)
        eng.set_tuning("sample_group", 1)
        out["sample"]["cell_ordered"]["without_grouping"] = {"ms": ungrouped * 1e3, "points_per_s": n_pts / ungrouped,
                                                             "note": "tuning \"sample_group\" 0: the batch goes as it comes"}
        del pos, nrm, rgb, cage, order
    if extras and not ddgi_mode and not sharded:
        # ---- what frames in flight is worth: the same loop with every launch tracing its own update only, and with four ----
        sweep = {}
        for n in (1, 2, 4, 8):
            if n == fif:
                continue
            eng.set_tuning("frames_in_flight", n)
            dt = timed(args.steps, 2)
            sweep[str(n)] = {"ms_per_step": dt / args.steps * 1e3, "value": total_rays / (dt / args.steps)}
        eng.set_tuning("frames_in_flight", fif)
        sweep[str(fif)] = {"ms_per_step": ms_per_step, "value": out["value"], "headline": True}
        # the drop-in number: the reference's host runs MAX_FRAMES_IN_FLIGHT = 2 ahead (src/rvpt/rvpt.h:23) — the headline above needs a host that runs
        # `frames_in_flight` updates ahead
        if "2" in sweep:
            out["ms_per_step_at_reference_frames_in_flight"] = sweep["2"]["ms_per_step"]
            out["value_at_reference_frames_in_flight"] = sweep["2"]["value"]
        out["frames_in_flight"] = dict(sweep, note="tuning \"frames_in_flight\": 1 = every launch traces its own update and drains; n = a launch goes on with up to n - 1 "
                                                   "updates submitted behind it (the timed loop submits its steps back to back, as the contract asks). The headline uses the library's default")
    if extras and not ddgi_mode and not sharded:
        # ---- the boundary handing over HOST buffers: the reference re-uploads its rays every frame (src/rvpt/rvpt.cpp:285) ----
        # ddgi_upload_probe_rays checks every ray's probe_info against the grid, keeps a host copy and copies 48 B/ray over PCIe; then the update.
        # Never `value` (the timed region above starts with the rays resident in HBM): the PCIe-inclusive rate of a host that does what the reference's does.
        host_rays = eng.get_probe_rays()
        other_rays = host_rays.copy()
        other_rays["direction"] = -other_rays["direction"]   # (a second, different ray set: every chunk of the buffer changes between two frames)
        n_host = max(2, min(6, args.steps)) & ~1

        def host_loop(buffers):
            eng.upload_probe_rays(buffers[0]); step(); fence()
            t_up, t0 = 0.0, time.perf_counter()
            for k in range(n_host):
                t1 = time.perf_counter()
                eng.upload_probe_rays(buffers[(k + 1) % len(buffers)])
                t_up += time.perf_counter() - t1
                step()
            fence()
            dt = (time.perf_counter() - t0) / n_host
            return {"ms_per_step": dt * 1e3, "value": total_rays / dt, "unit": "rays/s", "upload_ms": t_up / n_host * 1e3, "upload_GBps": host_rays.nbytes / (t_up / n_host) / 1e9}

        new_rays = host_loop([host_rays, other_rays])
        same_rays = host_loop([host_rays])
        out["host_buffers"] = {"bytes_per_step": int(host_rays.nbytes), "steps": n_host,
                               "same_rays_every_frame": dict(same_rays, note="the reference's host: one ray set, generated at start-up (main.cpp:47), the whole buffer handed over every frame — chunks equal to "
                                                                              "the handle's host copy are recognised (two reads of the buffer on host threads) and nothing crosses PCIe; upload_GBps is the buffer's size over the call's time"),
                               "new_rays_every_frame": dict(new_rays, note="two ray sets in turn: every chunk is checked, copied into the page-locked host copy and sent (48 B/ray over PCIe, synchronous)"),
                               "note": "every step = ddgi_upload_probe_rays + ddgi_probe_update, as the reference's host does per frame (probe_buffer.copy_to of the whole buffer, rvpt.cpp:285); "
                                       "PCIe-inclusive, reported beside `value`, never as it"}
        eng.upload_probe_rays(host_rays)
        del other_rays
        del host_rays
    fast_albedo = None
    if world == 1 and not ddgi_mode and not args.no_fast_march:
        # the opt-in tolerance-mode march on the same workload, timed the same way (the headline `value` above is the exact march)
        eng.set_tuning("fast_march", 1)
        if not pinned_split:
            eng.tune()
        fast_elapsed = timed(args.steps, args.warmup)
        ftr, _ = eng.update_history_ms(min(args.steps, 64))
        out["fast_march"] = {
            "ms_per_step": fast_elapsed / args.steps * 1e3, "value": total_rays / (fast_elapsed / args.steps), "unit": "rays/s",
            "kernel_ms": float(np.mean(ftr)), "active": bool(eng.get_tuning("fast_march_active")), "march_waves": eng.get_tuning("march_waves_measured"),
            "speedup_vs_exact": kernel_ms / float(np.mean(ftr)),
            "note": "ddgi_set_tuning(h, \"fast_march\", 1): empty-space skipping through a 2-bit per-voxel skip field in LDS; NOT bit-exact — tolerance below and in tests/test_gpu_fast_march.py",
        }
        fast_albedo = eng.read_textures()[0]
        eng.set_tuning("fast_march", 0)
    out["setup_ms"] = dict({k: round(v, 3) for k, v in setup_ms.items()},
                           note="host wall clock of what happens ONCE per handle / configuration, outside the timed region: the first update uploads the baked scene, builds and "
                                "uploads the memoised noise lattice (70 MB) and runs k_light_visibility (the light is static: once); ddgi_tune measures the wave split; "
                                "the sampler's per-texel table is rebuilt by the first large batch after every update")
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not ddgi_mode:
        out["cpu_baseline"], fast_tol = cpu_baseline(args.cpu_probes, gpu_albedo=exact_albedo, w=w,
                                                     rays=None if args.workload == "c3" else eng.get_probe_rays(), fast_albedo=fast_albedo)
        if fast_tol and "fast_march" in out:
            out["fast_march"]["tolerance"] = fast_tol
    if exchanging:
        if sharded:
            dist.barrier()                      # every rank has stopped pushing before any rank unmaps / frees
        eng.exchange_init(None)
    eng.close()
    if comm is not None:
        ddgi_amd.comm_destroy(comm)
    if sharded:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))
    if fallback:
        sys.stdout.flush()
        os._exit(0)                             # (a helper thread may still sit inside the RCCL call that never answered)


if __name__ == "__main__":
    main()
