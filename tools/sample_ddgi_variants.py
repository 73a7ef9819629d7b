#!/usr/bin/env python3
"""The DDGI-mode cage sampler on 1.44 M points over the C3 grid (GPU box): scattered or in cell order, grouped by cage or not."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import ddgi_amd
from bench import WORKLOAD as w

n = 1600 * 900
rng = np.random.default_rng(0)
half = np.array(w["counts"], dtype=np.float64) * w["side"] * 0.47
p = (rng.uniform(-1, 1, size=(n, 3)) * half + np.array(w["origin"])).astype(np.float32)
nr = rng.normal(size=(n, 3)).astype(np.float32)
order = np.lexsort((p[:, 0], p[:, 1], p[:, 2]))
eng = ddgi_amd.ProbeEngine(ddgi_amd.make_field(w["counts"], w["side"], w["s"], w["origin"]), ddgi_amd.make_settings(w["scene"], 8))
eng.set_stream(torch.cuda.current_stream().cuda_stream)
eng.set_mode(ddgi_amd.MODE_DDGI)
for f in range(3):
    eng.probe_update(ddgi_amd.make_settings(w["scene"], 8, time=2.0 * f))
rgb = torch.empty((n, 3), dtype=torch.float32, device="cuda")
cage = torch.empty((n, 8), dtype=torch.int32, device="cuda")
for label, idx in (("scattered", np.arange(n)), ("sorted by cell", order)):
    pos, nrm = torch.from_numpy(p[idx]).cuda(), torch.from_numpy(nr[idx]).cuda()
    for group in (1, 0):
        eng.set_tuning("sample_group", group)
        for _ in range(3):
            eng.sample_device(pos.data_ptr(), nrm.data_ptr(), n, rgb.data_ptr(), cage.data_ptr())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            eng.sample_device(pos.data_ptr(), nrm.data_ptr(), n, rgb.data_ptr(), cage.data_ptr())
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20
        print("%-15s grouped %d: %.3f ms  %.2f G points/s" % (label, group, dt * 1e3, n / dt / 1e9), flush=True)
eng.close()
