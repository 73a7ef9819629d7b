#!/usr/bin/env python3
"""Generates the golden fixtures under tests/golden/ from the CPU oracle (PINNED arithmetic).

The reference ships no fixtures and cannot be run in the build container, so these
are frozen outputs of this repo's own restatement of its algorithm: they pin the oracle (and the HIP
path) against silent drift between rounds.  Data only: inputs and expected outputs.
    python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle_py as O  # noqa: E402
from tests.common import CONFIGS, shading_points  # noqa: E402


def main():
    O.build()
    O.set_arith(True)
    out = {}
    for name in ("c1_cornell", "cave_small", "house_small"):
        counts, side, s, origin, scene = CONFIGS[name]
        f = O.make_field(counts, side, s, origin)
        rays = O.generate_probe_rays(f, O.new_rand_state(1))
        albedo, distance = O.probe_update(f, O.make_settings(scene, 8), rays)
        pos, nrm = shading_points(np.random.default_rng(21), counts, side, origin, 256)
        rgb, cage = O.sample(f, albedo, distance, pos, nrm)
        out[name + "_rays"] = rays.view(np.uint8).reshape(len(rays), 48)
        out[name + "_albedo"] = albedo
        out[name + "_pos"] = pos
        out[name + "_nrm"] = nrm
        out[name + "_rgb"] = rgb
        out[name + "_cage"] = cage
    np.savez_compressed(os.path.join(HERE, "probe_path_golden.npz"), **out)
    kat = {"rng": {str(p): O.rng_kat(p) for p in (0, 1, 2, 255, 12345, 4194303)},
           "glibc_rand_seed1_first8": [O.glibc_rand(st) for st in [O.new_rand_state(1)] for _ in range(8)],
           "cornell_march": []}
    for d in [(1, 0.25, 0.125), (0.1, 1, 0.2), (0.2, -0.1, 1), (0.3, -1, -0.2), (-0.3, -0.2, -1), (0.9, 0.3, 0.31)]:
        b, it, o = O.grid_march((0, 0, 15), d, 1)
        kat["cornell_march"].append({"dir": d, "block": b, "iters": it, "out_hex": [float(x).hex() for x in o]})
    with open(os.path.join(HERE, "kat.json"), "w") as fh:
        json.dump(kat, fh, indent=1)
    print("wrote", sorted(os.listdir(HERE)))


if __name__ == "__main__":
    main()
