"""DDGI mode: what continuing an update in its predecessor's launch costs or saves on a grid of COUNTS probes (C5 by default), 1 or 4 lights."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import ddgi_amd
from bench import WORKLOADS
w = WORKLOADS["c5"]
counts = tuple(int(x) for x in os.environ.get("COUNTS", "128,64,128").split(","))
for fif, sync in ((1, False), (8, False), (8, True), (1, False), (8, False)):
    eng = ddgi_amd.ProbeEngine(ddgi_amd.make_field(counts, w["side"], w["s"], w["origin"]), ddgi_amd.make_settings(w["scene"], 8))
    eng.set_mode(ddgi_amd.MODE_DDGI)
    if os.environ.get("LIGHTS", "4") == "4":
        eng.set_lights(w["scene"], np.array(w["lights"], dtype=ddgi_amd.LIGHT_DTYPE))
    eng.set_tuning("frames_in_flight", fif)
    eng.set_tuning("march_waves", int(os.environ.get("MW", "4")))
    if os.environ.get("LIGHT_VIS"):
        eng.set_tuning("light_vis", int(os.environ["LIGHT_VIS"]))
    t = [0.0]
    def upd():
        t[0] += 2.0
        eng.probe_update(ddgi_amd.make_settings(w["scene"], 8, time=t[0]))
        if sync: eng.synchronize()
    for _ in range(4): upd()
    eng.synchronize()
    c0 = eng.get_tuning("continued_workgroups")
    t0 = time.perf_counter()
    N = 8
    for _ in range(N): upd()
    eng.synchronize()
    dt = (time.perf_counter() - t0) / N * 1e3
    print("fif %d sync-every-update %s: %.3f ms per update, %d continuations" % (fif, sync, dt, eng.get_tuning("continued_workgroups") - c0), flush=True)
    tr, bl = eng.update_history_ms(N)
    print("   launches (ms):", " ".join("%.1f" % x for x in tr), "| blends:", " ".join("%.2f" % x for x in bl), flush=True)
    eng.close()
