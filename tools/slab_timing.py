#!/usr/bin/env python3
"""Kernel time of ONE rank's slab of the bench workload at world = 1, 2, 4, 8 (GPU box, one GPU):
what strong scaling can reach before any exchange cost."""
import os, sys
os.environ.setdefault("DDGI_AUTOTUNE", "1")  # tools measure the steady state: let the first update of a configuration pick the wave split
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ddgi_amd
from bench import WORKLOAD as w

base = None
for world in (1, 2, 4, 8):
    for mode in ("ref", "ddgi", "ref-fast", "ddgi-fast"):
        fast = mode.endswith("-fast")
        mode = mode.split("-")[0]
        eng = ddgi_amd.ProbeEngine(ddgi_amd.make_field(w["counts"], w["side"], w["s"], w["origin"]),
                                   ddgi_amd.make_settings(w["scene"], w["max_bounces"]), rank=world // 2, world=world)
        eng.set_tuning("fast_march", 1 if fast else 0)
        if mode == "ddgi":
            eng.set_mode(ddgi_amd.MODE_DDGI)
        else:
            eng.generate_probe_rays(seed=1)
        label = mode + ("-fast" if fast else "")
        for _ in range(8):
            eng.probe_update()
        eng.synchronize()
        tr, bl = eng.update_history_ms(5)
        t = float(np.mean(tr)) + (float(np.mean(bl)) if mode == "ddgi" else 0.0)
        if world == 1:
            base = base or {}
            base[label] = t
        print("world %d %-9s rank %d: trace %.3f ms blend %.3f ms  -> speed-up of the slab kernels %.2fx" % (
            world, label, world // 2, float(np.mean(tr)), float(np.mean(bl)) if mode == "ddgi" else 0.0, base[label] / t))
        eng.close()
