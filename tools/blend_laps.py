#!/usr/bin/env python3
"""Where a stage of k_probe_blend_depth_res goes (GPU box; library built with -DDDGI_BLEND_LAPS: `make alt ALTFLAGS=-DDDGI_BLEND_LAPS ALTNAME=laps`):
workgroup 0's twelve waves stamp s_memtime before and after every barrier.  Prints, per wave, the cycles from the previous barrier's
release to its arrival at the next one (its own work) — the stage lasts as long as the slowest."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("DDGI_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dynamic-diffuse-global-illumination-minecraft_amd", "libddgi_probe_laps.so"))
import numpy as np
import ddgi_amd
from bench import WORKLOAD as w

eng = ddgi_amd.ProbeEngine(ddgi_amd.make_field(w["counts"], w["side"], w["s"], w["origin"]), ddgi_amd.make_settings(w["scene"], w["max_bounces"]))
eng.set_mode(ddgi_amd.MODE_DDGI)
for f in range(4):
    eng.probe_update(ddgi_amd.make_settings(w["scene"], w["max_bounces"], time=2.0 * f))
eng.synchronize()
lib = ddgi_amd.load_library()
laps = np.zeros((14, 24), dtype=np.uint64)
assert lib.ddgi_debug_blend_laps(laps.ctypes.data_as(C.c_void_p)) == 0
t0 = laps[:12, 22].min()   # the first wave's entry
names = ["tile %d" % i for i in range(7)] + ["texels q0 q1 + fetch", "(last stage: q1)", "(last stage: q3)", "idle", "texels q2 q3 + fetch"]
print("cycles (s_memtime); columns: entry and arrival at the first barrier since the first wave's entry | per stage: own work until it reaches the barrier, [stage length]")
for wv in range(12):
    row = laps[wv].astype(np.int64)
    out = ["%-22s" % names[wv], "entry %6d  at the first barrier %6d |" % (row[22] - t0, row[0] - t0)]
    k = 1
    while 2 + 2 * (k - 1) < 22 and row[2 + 2 * (k - 1)] != 0:
        work = row[2 + 2 * (k - 1)] - row[1 + 2 * (k - 1)]
        stage = row[3 + 2 * (k - 1)] - row[1 + 2 * (k - 1)]
        out.append("%6d [%6d]" % (work, stage))
        k += 1
    print(" ".join(out))

print("k_probe_blend_irr, workgroup 0 (cycles since entry): slots written | barrier | contraction done | staged + barrier | texels stored")
for wv in (12, 13):
    row = laps[wv].astype(np.int64)
    print("wave %d: " % (wv - 12), "  ".join("%6d" % (row[i] - row[22]) for i in range(5)), "  | second step of every eight of the contraction:", "  ".join("%6d" % (row[i] - row[22]) for i in range(5, 9)))
