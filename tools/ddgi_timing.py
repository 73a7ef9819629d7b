#!/usr/bin/env python3
"""DDGI-mode kernel times for the bench workload (GPU box)."""
import os, sys
os.environ.setdefault("DDGI_AUTOTUNE", "1")  # tools measure the steady state: let the first update of a configuration pick the wave split
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ddgi_amd
from bench import WORKLOAD as w

if os.environ.get("DDGI_TIMING_COUNTS"):   # another grid size for the same scene, e.g. 32,32,32
    w = dict(w, counts=tuple(int(v) for v in os.environ["DDGI_TIMING_COUNTS"].split(",")))

eng = ddgi_amd.ProbeEngine(ddgi_amd.make_field(w["counts"], w["side"], w["s"], w["origin"]),
                           ddgi_amd.make_settings(w["scene"], w["max_bounces"]))
eng.set_mode(ddgi_amd.MODE_DDGI)
for f in range(8):
    eng.probe_update(ddgi_amd.make_settings(w["scene"], w["max_bounces"], time=2.0 * f))
tr, bl = eng.update_history_ms(8)
print("trace ms", np.round(tr, 3), "blend ms", np.round(bl, 3))
P = eng.num_probes
print("blend: %.1f MB algorithmic (2 x 3 KB tiles/probe) + %.1f MB ray records -> %.0f GB/s" % (
    P * 6144 / 1e6, eng.num_rays * 20 / 1e6, (P * 6144 + eng.num_rays * 20) / (bl[-1] * 1e-3) / 1e9))
rng = np.random.default_rng(0)
import time
n = 1 << 20
pos = (rng.uniform(-30, 30, size=(n, 3))).astype(np.float32); nrm = rng.normal(size=(n, 3)).astype(np.float32)
t0 = time.perf_counter(); rgb, cage = eng.sample(pos, nrm); dt = time.perf_counter() - t0
print("sample (host round trip, %d points): %.1f ms" % (n, dt * 1e3))
