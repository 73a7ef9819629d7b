#!/usr/bin/env python3
"""tools/aq_stats.py for S-Dyn's four animated lights on a C3-sized grid (DDGI mode): events and marches per ray with one light and with four."""
import os, sys
os.environ.setdefault("DDGI_AUTOTUNE", "1")
os.environ["DDGI_TRACE_KERNEL"] = "queues"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ddgi_amd
from bench import WORKLOAD as w, WORKLOAD_C5 as w5

for four in (False, True):
    eng = ddgi_amd.ProbeEngine(ddgi_amd.make_field(w["counts"], w["side"], w["s"], w["origin"]), ddgi_amd.make_settings(w["scene"], w["max_bounces"]))
    eng.set_mode(ddgi_amd.MODE_DDGI)
    if four:
        eng.set_lights(w["scene"], np.array(w5["lights"], dtype=ddgi_amd.LIGHT_DTYPE))
    st0 = ddgi_amd.make_settings(w["scene"], w["max_bounces"])
    eng.tune()
    for f in range(9):
        st0.time = 2.0 * (f + 1)
        eng.probe_update(st0)
    eng.synchronize()
    ms = eng.last_update_ms()
    eng.trace_stats(True)
    st0.time = 20.0
    eng.probe_update(st0); eng.synchronize()
    st = eng.trace_stats(False)
    rays = eng.num_rays
    trips, lanes, groups, glanes = st["trips"], st["lane_steps"], st["event_rounds"], st["lane_events"]
    print("lights %d: production trace %.3f ms;  march: %.1f lane-trips/ray, %.1f of 64 lanes busy per trip;  events: %.2f per ray, %.1f of 64 lanes per group" % (
        4 if four else 1, ms["trace_ms"], lanes / rays, lanes / max(trips, 1), glanes / rays, glanes / max(groups, 1)))
    for nm, (visits, lns) in st["sections"].items():
        if visits:
            print("  section %-34s visits %8d  lanes/visit %5.1f  lane-visits/ray %6.2f" % (nm, visits, lns / visits, lns / rays))
    eng.close()
