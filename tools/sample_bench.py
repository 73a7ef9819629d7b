#!/usr/bin/env python3
"""Device-resident throughput of the cage-sample kernels (REF and DDGI mode) on the bench grid."""
import os
os.environ.setdefault("DDGI_AUTOTUNE", "1")  # tools measure the steady state: let the first update of a configuration pick the wave split
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import ddgi_amd
from bench import WORKLOADS

w = WORKLOADS[os.environ.get("SAMPLE_WORKLOAD", "c3")]   # SAMPLE_WORKLOAD=c5 SAMPLE_MODES=1: the DDGI sampler beyond the caches (3.2 GB of tiles)
n = 1600 * 900  # one full-HD-ish frame of shading points (the reference's window, main.cpp:40-41)
rng = np.random.default_rng(0)
half = np.array(w["counts"], dtype=np.float64) * w["side"] * 0.47
pos = torch.from_numpy((rng.uniform(-1, 1, size=(n, 3)) * half + np.array(w["origin"])).astype(np.float32)).cuda()
nrm = torch.from_numpy(rng.normal(size=(n, 3)).astype(np.float32)).cuda()
rgb = torch.empty((n, 3), dtype=torch.float32, device="cuda")
cage = torch.empty((n, 8), dtype=torch.int32, device="cuda")
for mode in [int(m) for m in os.environ.get("SAMPLE_MODES", "0,1").split(",")]:
    eng = ddgi_amd.ProbeEngine(ddgi_amd.make_field(w["counts"], w["side"], w["s"], w["origin"]), ddgi_amd.make_settings(w["scene"], 8))
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    if w["tile"] != (w["s"], w["s"]):
        eng.set_ray_tile(*w["tile"])
    eng.set_mode(mode)
    if mode == ddgi_amd.MODE_REF:
        eng.generate_probe_rays(seed=1)
    eng.probe_update()
    for it in range(3):
        eng.sample_device(pos.data_ptr(), nrm.data_ptr(), n, rgb.data_ptr(), cage.data_ptr())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 20
    for it in range(K):
        eng.sample_device(pos.data_ptr(), nrm.data_ptr(), n, rgb.data_ptr(), cage.data_ptr())
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    inside = float((cage[:, 0] >= 0).float().mean())
    # REF mode: a large batch reads sample_probe from a per-texel table that the FIRST batch after a probe update has to build
    # (k_sample_box_filter) — time that first batch too: update, synchronise, one batch
    first = []
    for it in range(5):
        eng.probe_update()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        eng.sample_device(pos.data_ptr(), nrm.data_ptr(), n, rgb.data_ptr(), cage.data_ptr())
        torch.cuda.synchronize()
        first.append(time.perf_counter() - t1)
    print("mode %d: %.3f ms for %d points (%.0f Mpoints/s), %.1f GB/s of point I/O (68 B/point), %.0f%% inside the grid; the first batch after a probe update: %.3f ms (%.0f Mpoints/s)" % (
        mode, dt * 1e3, n, n / dt / 1e6, n * 68 / dt / 1e9, inside * 100, min(first) * 1e3, n / min(first) / 1e6))
    eng.close()
