#!/usr/bin/env python3
"""Trace-kernel time against launch size (GPU box): the fill/drain latency of the round structure."""
import os, sys
os.environ.setdefault("DDGI_AUTOTUNE", "1")  # tools measure the steady state: let the first update of a configuration pick the wave split
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ddgi_amd
from bench import WORKLOAD as w

for counts in [(32, 16, 32), (32, 16, 16), (32, 16, 8), (32, 16, 4), (32, 16, 2), (32, 16, 1), (32, 4, 1), (8, 4, 1), (2, 1, 1)]:
    eng = ddgi_amd.ProbeEngine(ddgi_amd.make_field(counts, w["side"], w["s"], w["origin"]), ddgi_amd.make_settings(w["scene"], w["max_bounces"]))
    eng.generate_probe_rays(seed=1)
    for _ in range(6):
        eng.probe_update()
    eng.synchronize()
    tr, _ = eng.update_history_ms(4)
    eng.trace_stats(True)
    eng.probe_update(); eng.synchronize()
    st = eng.trace_stats(False)
    wgs = st["waves"] // 16
    print("%-14s rays %8d  wgs %3d rays/wg %6d  %.3f ms  rounds/wg %.0f" % (counts, eng.num_rays, wgs, eng.num_rays // max(wgs, 1), float(np.mean(tr)), st["rounds"] / max(wgs, 1)))
    eng.close()
