import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import ddgi_amd
from bench import WORKLOAD as w
eng = ddgi_amd.ProbeEngine(ddgi_amd.make_field(w["counts"], w["side"], w["s"], w["origin"]), ddgi_amd.make_settings(w["scene"], w["max_bounces"]))
eng.generate_probe_rays(seed=1)
eng.set_tuning("timing", 0)
eng.set_tuning("fast_march", 1)
row = []
for mw in (4, 5, 6, 7):
    eng.set_tuning("march_waves", mw)
    best = 1e9
    for rep in range(3):
        for _ in range(4): eng.probe_update()
        eng.synchronize()
        t0 = time.perf_counter()
        for _ in range(24): eng.probe_update()
        eng.synchronize()
        best = min(best, (time.perf_counter() - t0) / 24 * 1e3)
    row.append("%d:%.4f" % (mw, best))
print(os.path.basename(os.environ.get("DDGI_LIB", "libddgi_probe.so")), "fast march", " ".join(row), flush=True)
