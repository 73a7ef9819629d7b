#!/usr/bin/env python3
"""C3 update time of k_probe_trace_aq against the march/event wave split (GPU box): wall clock over back-to-back updates,
frames_in_flight as given by DDGI_FRAMES_IN_FLIGHT (default: the library's).  DDGI_LIB selects the library (A/B builds)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ddgi_amd
from bench import WORKLOAD as w

splits = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "3,4,5,6,7,8,9").split(",")]
eng = ddgi_amd.ProbeEngine(ddgi_amd.make_field(w["counts"], w["side"], w["s"], w["origin"]), ddgi_amd.make_settings(w["scene"], w["max_bounces"]))
eng.generate_probe_rays(seed=1)
eng.set_tuning("timing", 0)
want = None
row = []
for mw in splits:
    eng.set_tuning("march_waves", mw)
    best = 1e9
    for rep in range(3):
        for _ in range(4):
            eng.probe_update()
        eng.synchronize()
        t0 = time.perf_counter()
        for _ in range(24):
            eng.probe_update()
        eng.synchronize()
        best = min(best, (time.perf_counter() - t0) / 24 * 1e3)
    a = eng.read_textures()[0]
    if want is None:
        want = a
    row.append("%d:%.4f%s" % (mw, best, "" if np.array_equal(a, want) else "(DIFFERS)"))
print(os.path.basename(os.environ.get("DDGI_LIB", "libddgi_probe.so")), "fif", eng.get_tuning("frames_in_flight"), " ".join(row), flush=True)
eng.close()
