#!/usr/bin/env python3
"""Where an event wave's cycles go (GPU box; needs the lap-timer build: make alt ALTNAME=lap ALTFLAGS="-DDDGI_PROFILING=1 -DDDGI_LAP=1",
DDGI_LIB=.../libddgi_probe_lap.so).  Cycles are wall cycles of the wave between two probes of its instruction stream, so they
include the turns of the other waves on the SIMD: read them as shares."""
import os, sys
os.environ.setdefault("DDGI_AUTOTUNE", "1")  # tools measure the steady state: let the first update of a configuration pick the wave split
os.environ["DDGI_TRACE_KERNEL"] = "queues"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ddgi_amd
from bench import WORKLOAD as w

eng = ddgi_amd.ProbeEngine(ddgi_amd.make_field(w["counts"], w["side"], w["s"], w["origin"]), ddgi_amd.make_settings(w["scene"], w["max_bounces"]))
if len(sys.argv) > 1 and sys.argv[1] == "ddgi":
    eng.set_mode(ddgi_amd.MODE_DDGI)
else:
    eng.generate_probe_rays(seed=1)
eng.probe_update(); eng.synchronize()
eng.trace_stats(True)
eng.probe_update(); eng.synchronize()
st = eng.trace_stats(False)
rays = eng.num_rays
tot = sum(c for c, _ in st["sections"].values()) or 1
print("kernel %.3f ms; event waves' cycles by section (share, lanes active at the section's probe per ray):" % eng.last_update_ms()["trace_ms"])
for nm, (cyc, lanes) in st["sections"].items():
    if cyc:
        print("  %-34s %6.3f   lane-visits/ray %6.2f" % (nm, cyc / tot, lanes / rays))
