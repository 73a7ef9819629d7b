#!/usr/bin/env python3
"""Where a task of k_probe_blend_irr12 goes (GPU box; library built with -DDDGI_BLEND_LAPS: `make alt ALTFLAGS=-DDDGI_BLEND_LAPS ALTNAME=laps`):
workgroup 0's twelve waves stamp the cycle counter at entry, around the barriers, at the second step of every eight of the contraction, after it and after the texels."""
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
t0 = laps[:12, 22].min()
print("k_probe_blend_irr12, workgroup 0, cycles since the first wave's entry: entry | tiles + slots written | barrier | contraction: second step of 8, 16, 24, 32 | done | staged + barrier | texels stored")
for wv in range(12):
    row = laps[wv].astype(np.int64) - int(t0)
    print("wave %2d (group %d tile %d channel %d): %6d | %6d | %6d | %6d %6d %6d %6d | %6d | %6d | %6d" % (wv, wv // 6, (wv % 6) // 3, wv % 3, row[22], row[0], row[1], row[5], row[6], row[7], row[8], row[2], row[3], row[4]))
