#!/usr/bin/env python3
"""Would a second stream hide the drain?  Two handles (two streams) on one GPU tracing the same workload back to back, against one
handle tracing as many updates alone: if the dispatcher starts the next launch's workgroups on the CUs the previous launch's
draining workgroups free one by one, the pair's aggregate time per update drops below the single handle's (GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ddgi_amd
from bench import WORKLOAD as w

N = 40


def make(world, fif):
    e = ddgi_amd.ProbeEngine(ddgi_amd.make_field(w["counts"], w["side"], w["s"], w["origin"]), ddgi_amd.make_settings(w["scene"], w["max_bounces"]), rank=world // 2, world=world)
    e.set_tuning("frames_in_flight", fif)
    e.generate_probe_rays(seed=1)
    e.tune()
    e.set_tuning("timing", 0)
    return e


for world in (1, 8):
    for fif in (1, 2):
        a, b = make(world, fif), make(world, fif)
        best1 = best2 = 1e9
        for rep in range(3):
            for e in (a, b):
                for _ in range(4):
                    e.probe_update()
                e.synchronize()
            t0 = time.perf_counter()
            for _ in range(2 * N):
                a.probe_update()
            a.synchronize()
            best1 = min(best1, (time.perf_counter() - t0) / (2 * N) * 1e3)
            t0 = time.perf_counter()
            for k in range(N // fif):
                for _ in range(fif):
                    a.probe_update()
                for _ in range(fif):
                    b.probe_update()
            a.synchronize(), b.synchronize()
            best2 = min(best2, (time.perf_counter() - t0) / (2 * (N // fif) * fif) * 1e3)
        print("world %d frames_in_flight %d: one handle %.4f ms per update, two handles on two streams %.4f ms per update (%.1f %%)" % (
            world, fif, best1, best2, 100.0 * (best2 / best1 - 1.0)), flush=True)
        a.close(), b.close()
