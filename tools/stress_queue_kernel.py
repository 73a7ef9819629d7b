#!/usr/bin/env python3
"""Repeats the bench workload's update many times and checks that the output never changes (GPU box):
the queue kernel's execution order differs from run to run, its result must not."""
import os
os.environ.setdefault("DDGI_AUTOTUNE", "1")  # tools measure the steady state: let the first update of a configuration pick the wave split
import hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ddgi_amd
from bench import WORKLOAD as w

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
eng = ddgi_amd.ProbeEngine(ddgi_amd.make_field(w["counts"], w["side"], w["s"], w["origin"]), ddgi_amd.make_settings(w["scene"], w["max_bounces"]))
eng.generate_probe_rays(seed=1)
ref = None
t0 = time.time()
for k in range(n):
    eng.probe_update()
    h = hashlib.sha256(eng.read_textures()[0].tobytes()).hexdigest()
    if ref is None:
        ref = h
    elif h != ref:
        print("MISMATCH at update", k, h, ref)
        sys.exit(1)
print("REF mode: %d updates, one digest %s (%.1f s)" % (n, ref[:16], time.time() - t0))
# frames in flight: runs of back-to-back updates of random length with random host delays in between — some submissions reach a
# predecessor that still runs (and are continued by its workgroups), some find the stream idle, some arrive as it drains
rng = np.random.default_rng(7)
t0 = time.time()
total = cont0 = 0
for fif in (8, 4, 2, 3):
    eng.set_tuning("frames_in_flight", fif)
    for rep in range(max(1, n // 4)):
        run = int(rng.integers(1, 21))
        for _ in range(run):
            eng.probe_update()
            if rng.random() < 0.3:
                time.sleep(float(rng.uniform(0.0, 0.003)))
        total += run
        h = hashlib.sha256(eng.read_textures()[0].tobytes()).hexdigest()
        if h != ref:
            print("MISMATCH after a run of", run, "updates at frames_in_flight", fif, h, ref)
            sys.exit(1)
print("REF mode, frames in flight 8 / 4 / 2 / 3: %d updates in random runs, %d workgroup continuations, one digest (%.1f s)" % (total, eng.get_tuning("continued_workgroups"), time.time() - t0))
eng.set_mode(ddgi_amd.MODE_DDGI)
ref = None
for k in range(n // 4):
    eng.configure()           # zero tiles, frame 0: every repetition is the same update
    eng.probe_update()
    irr, dep = eng.read_tiles()
    h = hashlib.sha256(irr.tobytes() + dep.tobytes()).hexdigest()
    if ref is None:
        ref = h
    elif h != ref:
        print("DDGI MISMATCH at", k)
        sys.exit(1)
print("DDGI mode: %d updates, one digest %s" % (n // 4, ref[:16]))
