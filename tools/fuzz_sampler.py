#!/usr/bin/env python3
"""Randomised parity sweep of the REF cage sampler's per-texel table (GPU box): random grids — odd and even probe counts, 1 .. 9 per axis —,
spacings, ray tiles (square and not), batches beyond the table's threshold with points everywhere, crowded into one cage, on the last cage
layers (wrapped corner indices) and outside; the table path (2x2x2 bricks of probes) against the per-point path and the pinned oracle, bit
for bit, over two updates (the table must follow the textures).  The oracle is the checker here, as in tests/.
Usage: tools/fuzz_sampler.py [cases] [seed]"""
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
    counts = tuple(int(v) for v in rng.integers(1, 10, size=3))
    side = int(rng.integers(1, 6))
    tile = (int(rng.integers(2, 9)), int(rng.integers(2, 9))) if rng.random() < 0.5 else None
    s = int(rng.integers(2, 9))
    centre = {0: (1.4, 0.0, 1.0), 1: (0.0, 0.0, 15.0), 2: (0.0, 0.0, 0.0)}[scene]
    origin = tuple(float(np.float32(c + rng.choice([0.0, 0.5, rng.uniform(-3, 3)]))) for c in centre)
    tag = f"case {case}: scene {scene} counts {counts} side {side} tile {tile or (s, s)} origin {origin}"
    n = 66000 + int(rng.integers(0, 9000))
    c = np.asarray(counts, dtype=np.float32)
    o = np.asarray(origin, dtype=np.float32)
    spread = (rng.uniform(-0.6, 0.6, size=(n - 9000, 3)) * c * side + o).astype(np.float32)
    crowd = (rng.uniform(0.05, 0.95, size=(3000, 3)) * side + o).astype(np.float32)
    edge = (rng.uniform(-0.5, 0.5, size=(6000, 3)) * c * side + o).astype(np.float32)
    ax = int(rng.integers(0, 3))
    edge[:, ax] = o[ax] + side * (counts[ax] // 2 - 0.5)        # the last cage layer along one axis
    pos = np.concatenate([spread, crowd, edge]).astype(np.float32)
    nrm = rng.normal(size=(len(pos), 3)).astype(np.float32)
    nrm[:4] = [[0, 0, 1], [0, 0, -1], [1, 0, 0], [0, 1, 0]]
    O.set_ray_tile(*(tile or (0, 0)))
    f = O.make_field(counts, side, s, origin)
    ok = True
    with ddgi_amd.ProbeEngine(ddgi_amd.make_field(counts, side, s, origin), ddgi_amd.make_settings(scene, 4)) as eng:
        if tile:
            eng.set_ray_tile(*tile)
        for seed in (1, 2):
            eng.generate_probe_rays(seed=seed, reseed=True)
            eng.probe_update()
            eng.set_tuning("sample_box", 1)
            rgb_t, cage_t = eng.sample(pos, nrm)
            eng.set_tuning("sample_box", 0)
            rgb_d, cage_d = eng.sample(pos, nrm)
            albedo, distance = eng.read_textures()
            want_rgb, want_cage = O.sample(f, albedo, distance, pos, nrm)
            ok &= np.array_equal(cage_t, want_cage) and np.array_equal(rgb_t.view(np.uint32), want_rgb.view(np.uint32))
            ok &= np.array_equal(cage_d, want_cage) and np.array_equal(rgb_d.view(np.uint32), want_rgb.view(np.uint32))
    inside = float((want_cage[:, 0] >= 0).mean())
    if not ok:
        bad += 1
        print("MISMATCH", tag)
    elif case % 8 == 0:
        print("ok", tag, "inside %.2f" % inside)
O.set_ray_tile(0, 0)
print("%d cases, %d mismatches, %.0f s" % (n_cases, bad, time.time() - t0))
sys.exit(1 if bad else 0)
