#!/usr/bin/env python3
"""Randomised parity sweep (GPU box): random grids / scenes / bounce counts / light sets, REF and DDGI
mode, HIP (default kernels) against the pinned oracle, bit for bit.  Not part of the test suite (the
oracle makes it slow); run after changes to the trace kernels.  The oracle is the checker here, as in tests/.
Usage: tools/fuzz_parity.py [cases] [seed]"""
import os
os.environ.setdefault("DDGI_AUTOTUNE", "1")  # tools measure the steady state: let the first update of a configuration pick the wave split
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ddgi_amd
from oracle import oracle_py as O

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
O.set_arith(1)
bad = 0
t0 = time.time()
for case in range(n_cases):
    scene = int(rng.integers(0, 3))
    counts = tuple(int(v) for v in rng.integers(1, 6, size=3))
    side = int(rng.integers(1, 12))
    s = int(rng.integers(1, 9))
    centre = {0: (1.4, 0.0, 1.0), 1: (0.0, 0.0, 15.0), 2: (0.0, 0.0, 0.0)}[scene]
    origin = tuple(float(np.float32(c + rng.choice([0.0, 0.5, rng.uniform(-6, 6)]))) for c in centre)   # incl. probes on voxel boundaries
    bounces = int(rng.integers(1, 11))
    lights = None
    if rng.random() < 0.4:
        nl = int(rng.integers(1, 5))
        lights = np.array([(float(rng.uniform(3, 20)), tuple(rng.uniform(0.1, 1.0, 3)), tuple(np.float32(np.array(centre) + rng.uniform(-12, 12, 3))))
                           for _ in range(nl)], dtype=ddgi_amd.LIGHT_DTYPE)
    ddgi_mode = rng.random() < 0.35
    tag = f"case {case}: scene {scene} counts {counts} side {side} s {s} origin {origin} bounces {bounces} lights {'shipped' if lights is None else len(lights)} {'DDGI' if ddgi_mode else 'REF'}"
    f = O.make_field(counts, side, s, origin, hysteresis=0.8)
    st = O.make_settings(scene, bounces)
    with ddgi_amd.ProbeEngine(ddgi_amd.make_field(counts, side, s, origin, hysteresis=0.8), ddgi_amd.make_settings(scene, bounces)) as eng:
        if lights is not None:
            eng.set_lights(scene, lights)
        if ddgi_mode:
            eng.set_mode(ddgi_amd.MODE_DDGI)
            P = counts[0] * counts[1] * counts[2]
            o_irr = np.zeros((P, 8, 8, 4), dtype=np.float32)
            o_dep = np.zeros((P, 16, 16, 2), dtype=np.float32)
            ok = True
            for frame in range(2):
                eng.probe_update()
                irr, dep = eng.read_tiles()
                O.ddgi_update(f, st, frame, o_irr, o_dep, lights=None if lights is None else np.array(lights, dtype=O.LIGHT_DTYPE))
                ok &= np.array_equal(irr.view(np.uint32), o_irr.view(np.uint32)) and np.array_equal(dep.view(np.uint32), o_dep.view(np.uint32))
        else:
            # (frames in flight: a random ring length and a random run of back-to-back updates of the same work — the raster read
            # behind the last one must be the oracle's whichever launch traced it)
            fif = int(rng.choice([1, 2, 3, 4, 8]))
            n_up = int(rng.integers(1, 12))
            eng.set_tuning("frames_in_flight", fif)
            eng.generate_probe_rays(seed=1)
            for _ in range(n_up):
                eng.probe_update()
            got, _ = eng.read_textures()
            tag += f" fif {fif} x {n_up} updates"
            rays = O.generate_probe_rays(f, O.new_rand_state(1))
            want, _ = O.probe_update(f, st, rays, lights=None if lights is None else np.array(lights, dtype=O.LIGHT_DTYPE))
            ok = np.array_equal(got, want)
    if not ok:
        bad += 1
        print("MISMATCH", tag)
    elif case % 10 == 0:
        print("ok", tag)
print("%d cases, %d mismatches, %.0f s" % (n_cases, bad, time.time() - t0))
sys.exit(1 if bad else 0)
