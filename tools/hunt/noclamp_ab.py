#!/usr/bin/env python3
"""A/B of the DDGI_EXP_NOCLAMP builds on a grid whose probes all stand INSIDE the cave's baked box (30 x 16 x 30, side 2: C3 has its last x and z columns in the
rock outside the box, whose rays the clamp ends at once and a build without it marches for 125 steps — that, not the three instructions, is what
profiles/r06_f_noclamp_ab.txt's first table measured).  Usage: DDGI_LIB=... python tools/hunt/noclamp_ab.py"""
import os, sys, time
os.environ.setdefault("DDGI_AUTOTUNE", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import ddgi_amd
counts = tuple(int(v) for v in os.environ.get("COUNTS", "30,16,30").split(","))
eng = ddgi_amd.ProbeEngine(ddgi_amd.make_field(counts, 2, 16, (1.4, 0.0, 1.0)), ddgi_amd.make_settings(0, 8))
eng.generate_probe_rays(seed=1)
eng.probe_update(); eng.synchronize()
eng.tune()
for _ in range(5): eng.probe_update()
eng.synchronize()
t = time.perf_counter()
n = 40
for _ in range(n): eng.probe_update()
eng.synchronize()
dt = (time.perf_counter() - t) / n
a = eng.read_textures()[0]
import hashlib
print("%s counts %s: %.4f ms per update (%.3f G rays/s), albedo sha1 %s" % (os.path.basename(os.environ.get("DDGI_LIB", "libddgi_probe.so")), counts, dt * 1e3, eng.num_rays / dt / 1e9, hashlib.sha1(a.tobytes()).hexdigest()[:12]))
