#!/usr/bin/env python3
"""Lane utilisation of k_probe_trace_aq's march trips and event groups on the bench workload (GPU box)."""
import os, sys
os.environ.setdefault("DDGI_AUTOTUNE", "1")  # tools measure the steady state: let the first update of a configuration pick the wave split
os.environ["DDGI_TRACE_KERNEL"] = "queues"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ddgi_amd
from bench import WORKLOAD as w

eng = ddgi_amd.ProbeEngine(ddgi_amd.make_field(w["counts"], w["side"], w["s"], w["origin"]), ddgi_amd.make_settings(w["scene"], w["max_bounces"]))
eng.generate_probe_rays(seed=1)
eng.probe_update(); eng.synchronize()
if os.environ.get("DDGI_LIB", "").endswith("_prof.so"):
    eng.set_tuning("ablate", 16)  # profiling build: count the feeler classes too (slow: global atomics per event)
eng.trace_stats(True)
eng.probe_update(); eng.synchronize()
st = eng.trace_stats(False)
rays = eng.num_rays
trips, lanes, groups, glanes = st["trips"], st["lane_steps"], st["event_rounds"], st["lane_events"]
print("march: %.1f lane-trips/ray, %.1f of 64 lanes busy per trip;  events: %.2f per ray, %.1f of 64 lanes per group;  kernel %.3f ms" % (
    lanes / rays, lanes / max(trips, 1), glanes / rays, glanes / max(groups, 1), eng.last_update_ms()["trace_ms"]))
q = [st[k] for k in ("cyc_scan", "cyc_march", "cyc_march_wait", "cyc_list", "cyc_events", "cyc_events_wait")] + [st["_14"], st["_15"]]
if q[0]:
    print("queues at an event wave's poll: MQ %.0f  FQ %.0f  EQ (all buckets) %.0f slots;  idle polls %.2f of %d;  march bursts: %.2f per ray, "
          "%.1f lanes in flight at the start, %.2f of the fetches found MQ short" % (q[1] / q[0], q[2] / q[0], q[3] / q[0], q[4] / q[0], q[0],
          q[5] / rays, q[7] / max(q[5], 1), q[6] / max(q[5], 1)))
if st["rounds"]:
    steps = int(os.environ.get("DDGI_AQ_STEPS_BUILT", "24"))
    print("march bursts: %.2f useful lane-steps per ray = %.1f per burst of %d x 64 = %.3f of the lane-steps a burst issues; %.1f steps per lane-trip" % (
        st["rounds"] / rays, st["rounds"] / max(q[5], 1), steps, st["rounds"] / max(q[5], 1) / (steps * 64.0), st["rounds"] / max(lanes, 1)))
print("counts: march_bursts %d event_groups %d rays %d" % (q[5], groups, rays))
print("feelers per ray (profiling build, ablate 16):", {k: round(v / rays, 3) for k, v in st["feeler_classes"].items()})
for nm, (visits, lanes) in st["sections"].items():
    if visits:
        print("  section %-34s visits %8d  lanes/visit %5.1f  lane-visits/ray %6.2f" % (nm, visits, lanes / visits, lanes / rays))
