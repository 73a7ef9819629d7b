#!/usr/bin/env python3
"""Steady-state time per probe update of ONE rank's slab of the bench workload (C3, REF) at world = 1, 2, 4, 8 and
frames_in_flight = 1, 2, 4, 8 (GPU box, one GPU): wall clock over a run of back-to-back updates, no per-update events
(a continued update has no kernel time of its own).  What strong scaling can reach before any exchange cost."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ddgi_amd
from bench import WORKLOAD as w

N = int(os.environ.get("FIF_UPDATES", "48"))
fast = os.environ.get("FIF_FAST", "0") == "1"
ddgi_mode = os.environ.get("FIF_MODE", "ref") == "ddgi"   # DDGI mode: in-kernel rays, time += 2 per update (the light moves with it), trace + blend per update
clock = [0.0]
base = {}
WORLDS = tuple(int(x) for x in os.environ.get("FIF_WORLDS", "1,2,4,8").split(","))   # (a subset: e.g. FIF_WORLDS=8 FIF_FIFS=8 under rocprofv3 for a timeline)
FIFS = tuple(int(x) for x in os.environ.get("FIF_FIFS", "1,2,4,8").split(","))
for world in WORLDS:
    for fif in FIFS:
        eng = ddgi_amd.ProbeEngine(ddgi_amd.make_field(w["counts"], w["side"], w["s"], w["origin"]),
                                   ddgi_amd.make_settings(w["scene"], w["max_bounces"]), rank=world // 2, world=world)
        eng.set_tuning("frames_in_flight", fif)
        eng.set_tuning("fast_march", 1 if fast else 0)
        if ddgi_mode:
            eng.set_mode(ddgi_amd.MODE_DDGI)
        else:
            eng.generate_probe_rays(seed=1)

        def update():
            if ddgi_mode:
                clock[0] += 2.0
                eng.probe_update(ddgi_amd.make_settings(w["scene"], w["max_bounces"], time=clock[0]))
            else:
                eng.probe_update()

        update()
        eng.tune()
        eng.set_tuning("timing", 0)
        best = None
        for rep in range(3):
            for _ in range(8):
                update()
            eng.synchronize()
            before = eng.get_tuning("continued_workgroups")
            t0 = time.perf_counter()
            for _ in range(N):
                update()
            eng.synchronize()
            dt = (time.perf_counter() - t0) / N * 1e3
            cont = eng.get_tuning("continued_workgroups") - before
            best = dt if best is None else min(best, dt)
        if world == 1 or 1 not in base:
            base.setdefault(fif, best), base.setdefault(1, best)
        print("%sworld %d frames_in_flight %d%s: %.4f ms per update (march waves %d, %d workgroup continuations in %d updates)  -> %.2fx of one GPU's %.3f ms (fif 1)" % (
            "DDGI " if ddgi_mode else "", world, fif, " fast-march" if fast else "", best, eng.get_tuning("march_waves_measured"), cont, N, base[1] / best, base[1]), flush=True)
        eng.close()
