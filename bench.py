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


def _tile_view(raster, w, mask):
    """The tiles of the masked probes out of a reference-layout raster: [n_probes, ty, tx, 4]."""
    tx, ty = w["tile"]
    cxz = w["counts"][0] * w["counts"][2]
    return raster.reshape(w["counts"][1], ty, cxz, tx, 4).transpose(0, 2, 1, 3, 4)[mask]


def _tolerance(a, b):
    d = np.abs(a[..., :3].astype(np.int32) - b[..., :3].astype(np.int32))
    return {"within_1_255": float((d <= 1).mean()), "mean_abs_diff_255": float(d.mean()), "texels_differing": int((d.max(axis=-1) > 0).sum())}


def cpu_baseline(n_probes=96, gpu_albedo=None, w=WORKLOAD, rays=None, fast_albedo=None):
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
    n = int(min(total, max(len(calib), rate * 12.0)))
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
    ap.add_argument("--mode", choices=["ref", "ddgi"], default="ref",
                    help="ref (default): the reference's live behaviour, the headline metric; ddgi: in-kernel Fibonacci rays + "
                         "octahedral irradiance/depth blend with hysteresis (trace + blend per step)")
    ap.add_argument("--exchange", choices=["rccl", "p2p"], default="rccl",
                    help="N > 1: how the ranks' slabs are exchanged — rccl (default): one in-place ncclAllGather per texture; "
                         "p2p: every rank pushes its slab into its peers' textures (ddgi_exchange_p2p_*, IPC-mapped buffers)")
    args = ap.parse_args()
    if args.workload == "c5" and args.mode != "ddgi":
        raise SystemExit("--workload c5 is S-Dyn (4 dynamic lights + temporal hysteresis): run it with --mode ddgi")
    if args.warmup is None:
        args.warmup = 8 if args.workload == "c5" else 5

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
    # the ranks share the chip; RCCL refuses two ranks on one device, so this needs --exchange p2p and the gloo backend)
    one_gpu = os.environ.get("DDGI_BENCH_ONE_GPU") == "1"
    if one_gpu:
        local_rank = 0
        if world > 1 and args.exchange != "p2p":
            raise SystemExit("DDGI_BENCH_ONE_GPU=1 needs --exchange p2p (RCCL refuses two ranks on one device)")
    torch.cuda.set_device(local_rank)
    # DDGI_BENCH_FORCE_DIST=1: run the N > 1 code path (RCCL group, pipelined exchange) with a single rank — a smoke test of
    # that path on a one-GPU box
    sharded = world > 1 or os.environ.get("DDGI_BENCH_FORCE_DIST") == "1"
    if sharded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        with _c_stdout_to_stderr():
            if one_gpu:
                dist.init_process_group(backend="gloo", rank=rank, world_size=world)
            else:
                dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    w = WORKLOADS[args.workload]
    field = ddgi_amd.make_field(w["counts"], w["side"], w["s"], w["origin"])
    settings = ddgi_amd.make_settings(w["scene"], w["max_bounces"])
    eng = ddgi_amd.ProbeEngine(field, settings, device=local_rank, rank=rank, world=world)
    if w["tile"] != (w["s"], w["s"]):
        eng.set_ray_tile(*w["tile"])
    stream = torch.cuda.current_stream()
    eng.set_stream(stream.cuda_stream)          # kernels + collectives share torch's stream
    ddgi_mode = args.mode == "ddgi"
    if ddgi_mode:
        eng.set_mode(ddgi_amd.MODE_DDGI)        # rays are generated in the kernel; tiles start zeroed
    else:
        eng.generate_probe_rays(seed=w["seed"])  # ray buffer resident in HBM from here on
    if w.get("lights"):
        eng.set_lights(w["scene"], np.array(w["lights"], dtype=ddgi_amd.LIGHT_DTYPE))  # animated per update from RenderSettings::time

    comm = None
    exchanging = False
    if sharded and (args.exchange == "rccl" or world == 1):
        # the engine issues the RCCL all-gather itself (include/ddgi_probe.h: ddgi_exchange_*): rank 0 makes the
        # 128-byte RCCL id, torch.distributed carries it to the other ranks, every rank joins the communicator
        with _c_stdout_to_stderr():
            ids = [ddgi_amd.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(ids, src=0)
            comm = ddgi_amd.comm_create(ids[0], world, rank, local_rank)
        # pipelined: the exchange of update k overlaps the kernels of update k+1 (two texture pairs inside the engine)
        eng.exchange_init(comm, pipelined=True)
        exchanging = True
    elif sharded:
        # peer-to-peer: every rank publishes its buffers (512 bytes), torch.distributed carries the addresses around
        mine = eng.exchange_p2p_export(pipelined=True)
        everyone = [None] * world
        dist.all_gather_object(everyone, mine)
        eng.exchange_p2p_init(everyone)
        exchanging = True

    pinned_split = bool(os.environ.get("DDGI_AQ_MARCH"))  # (profiling runs pin the split so that every launch is the steady-state kernel)
    if not pinned_split:
        eng.tune()                              # the march/event wave split of this configuration, measured once (blocks; outside the timed region)
    frame_time = [0.0]

    def step():
        if ddgi_mode:
            frame_time[0] += 2.0                # RVPT::update: render_settings.time += 2 (rvpt.cpp:281)
            settings.time = frame_time[0]
            eng.probe_update(settings)
        else:
            eng.probe_update()
        if exchanging:
            eng.exchange()

    def fence():
        if exchanging:
            eng.exchange_finish()               # the stream waits for every exchange issued so far
        if sharded:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if sharded:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if one_gpu else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # kernel durations of the timed steps: HIP events recorded on the launch stream by the engine
    trace_ms, blend_ms = eng.update_history_ms(min(args.steps, 64))
    # the same steps once more WITHOUT those events (tuning "timing" 0: what a caller who does not ask for per-update times gets);
    # reported beside the line's numbers, not instead of them
    untimed_ms = None
    if not sharded and not ddgi_mode:   # (DDGI mode: the lights move on with every step — later frames are not the same work)
        eng.set_tuning("timing", 0)
        step()
        fence()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        untimed_ms = (time.perf_counter() - t1) / args.steps * 1e3
        eng.set_tuning("timing", 1)
    kernel_ms = float(np.mean(trace_ms)) if len(trace_ms) else float("nan")

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
    out = {
        "metric": "probe_rays_per_sec",
        "value": total_rays / (elapsed / args.steps),
        "unit": "rays/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
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
            "parallelism": f"zslab{world}" + ((f"+allgather_{args.exchange}" + ("_all_ranks_on_one_gpu" if one_gpu else "")) if world > 1 else ""),
        },
        "roofline": {
            "kernel": {"lane": "k_probe_trace_ref", "rounds": "k_probe_trace_wf"}.get(os.environ.get("DDGI_TRACE_KERNEL", ""), "k_probe_trace_aq"),
            "bound": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic,
            "traffic_replayed_from": traffic_src,
            "algorithmic_bytes_per_launch": algo_bytes,
            "kernel_ms": kernel_ms,
            "issue": None if (ddgi_mode or args.workload != "c3") else _issue_from_profiles(),
            "note": "achieved / kernel_ms are measured by this run (HIP events on the launch stream); `traffic` and `issue` are REPLAYED from the committed rocprofv3 --pmc passes named beside them (counters need their own passes). The trace kernel is VALU-issue bound (dependent voxel steps + hit shading), not HBM bound, see DESIGN.md section 4",
        },
    }
    out["tuning"] = {"march_waves": eng.get_tuning("march_waves_measured"), "note": "waves of a 16-wave workgroup that march (the rest shade); measured by ddgi_tune() before the warm-up"}
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
    exact_albedo = eng.read_textures()[0] if (rank == 0 and world == 1 and not ddgi_mode) else None
    fast_albedo = None
    if world == 1 and not ddgi_mode and not args.no_fast_march:
        # the opt-in tolerance-mode march on the same workload, timed the same way (the headline `value` above is the exact march)
        eng.set_tuning("fast_march", 1)
        if not pinned_split:
            eng.tune()
        for _ in range(args.warmup):
            step()
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        fast_elapsed = time.perf_counter() - t0
        ftr, _ = eng.update_history_ms(min(args.steps, 64))
        out["fast_march"] = {
            "ms_per_step": fast_elapsed / args.steps * 1e3, "value": total_rays / (fast_elapsed / args.steps), "unit": "rays/s",
            "kernel_ms": float(np.mean(ftr)), "active": bool(eng.get_tuning("fast_march_active")), "march_waves": eng.get_tuning("march_waves_measured"),
            "speedup_vs_exact": kernel_ms / float(np.mean(ftr)),
            "note": "ddgi_set_tuning(h, \"fast_march\", 1): empty-space skipping through a 2-bit per-voxel skip field in LDS; NOT bit-exact — tolerance below and in tests/test_gpu_fast_march.py",
        }
        fast_albedo = eng.read_textures()[0]
        eng.set_tuning("fast_march", 0)
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


if __name__ == "__main__":
    main()
