#!/usr/bin/env python3
"""Prints the trace kernel's lane-utilisation counters for the bench workload (GPU box)."""
import os, sys
os.environ.setdefault("DDGI_AUTOTUNE", "1")  # tools measure the steady state: let the first update of a configuration pick the wave split
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ddgi_amd
from bench import WORKLOAD as w

world = int(os.environ.get("STATS_WORLD", "1"))   # STATS_WORLD=8: one rank's slab of the sharded grid
eng = ddgi_amd.ProbeEngine(ddgi_amd.make_field(w["counts"], w["side"], w["s"], w["origin"]),
                           ddgi_amd.make_settings(w["scene"], w["max_bounces"]), rank=world // 2, world=world)
if len(sys.argv) > 1 and sys.argv[1] == "ddgi":
    eng.set_mode(ddgi_amd.MODE_DDGI)
else:
    eng.generate_probe_rays(seed=1)
eng.probe_update(); eng.synchronize()
eng.trace_stats(True)
eng.probe_update(); eng.synchronize()
st = eng.trace_stats(False)
ms = eng.last_update_ms()["trace_ms"]
rays = eng.num_rays // world
print(st, "kernel_ms", ms)
cyc = [st[k] for k in ("cyc_scan", "cyc_march", "cyc_march_wait", "cyc_list", "cyc_events", "cyc_events_wait")]
if sum(cyc):
    print("phase share of wave time: sort %.3f march %.3f march-barrier %.3f (unused %.3f) events %.3f events-barrier %.3f | fetches/wave %.0f rounds %d" % (
        *[c / sum(cyc) for c in cyc], st["fetches"] / st["waves"], st["rounds"]))
print("steps/ray %.1f  march-lane-utilisation %.3f  events/ray %.2f  lanes/event-round %.1f  trips/wave %.0f rounds/wave %.0f" % (
    st["lane_steps"] / rays, st["lane_steps"] / (64.0 * st["trips"]), st["lane_events"] / rays,
    st["lane_events"] / st["event_rounds"], st["trips"] / st["waves"], st["event_rounds"] / st["waves"]))
names = ("stem", "wall", "ground/moss/mold", "caps+flat", "light-or-miss", "feeler", "dead-primary", "refill")
tot = sum(st["bucket_cycles"]) or 1
for nm, cyc, n in zip(names, st["bucket_cycles"], st["bucket_groups"]):
    print("  bucket %-18s groups %8d  cycles/group %8.0f  share of event time %.3f" % (nm, n, cyc / max(n, 1), cyc / tot))
