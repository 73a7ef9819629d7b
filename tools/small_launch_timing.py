#!/usr/bin/env python3
"""Trace-kernel time of one rank's slab of the bench workload for large world sizes (GPU box, one GPU), and of the smallest
slab with fewer bounces: the fixed part of the kernel's time — the latency of one ray's chain of marches and events."""
import os, sys
os.environ.setdefault("DDGI_AUTOTUNE", "1")  # tools measure the steady state: let the first update of a configuration pick the wave split
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, ddgi_amd
from bench import WORKLOAD as w

def run(world, bounces):
    eng = ddgi_amd.ProbeEngine(ddgi_amd.make_field(w["counts"], w["side"], w["s"], w["origin"]), ddgi_amd.make_settings(w["scene"], bounces), rank=world // 2, world=world)
    eng.generate_probe_rays(seed=1)
    for _ in range(8): eng.probe_update()
    eng.synchronize()
    tr, bl = eng.update_history_ms(5)
    print("world %2d  rays %7d  max_bounces %d  trace %.3f ms  (march waves %d)" % (world, eng.num_rays // world, bounces, float(np.mean(tr)), eng.get_tuning("march_waves_measured")))
    eng.close()

for world in (8, 16, 32):
    run(world, w["max_bounces"])
for b in (1, 2, 4):
    run(32, b)
