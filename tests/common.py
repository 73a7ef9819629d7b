"""Shared configurations for the parity tests (BASELINE.json `configs`, SURVEY.md §8 sizes)."""
import numpy as np

# name -> (counts, side_length, s, origin, scene)
CONFIGS = {
    # C1: Cornell box, 2x2x2 probes x 64 rays
    "c1_cornell": ((2, 2, 2), 6, 8, (0.0, 0.0, 15.0), 1),
    # C2: Cornell box, 8x8x8 probes x 256 rays
    "c2_cornell": ((8, 8, 8), 2, 16, (0.0, 0.0, 15.0), 1),
    # small cave / house grids for the oracle-sized parity cases
    "cave_small": ((4, 3, 4), 8, 8, (1.4, 0.0, 1.0), 0),
    "cave_odd": ((3, 3, 3), 9, 6, (1.4, 0.0, 1.0), 0),
    "house_small": ((4, 2, 2), 10, 8, (0.0, 0.0, 0.0), 2),
    # C3: Minecraft cave, 32x16x32 probes x 256 rays (the roofline configuration)
    "c3_cave": ((32, 16, 32), 2, 16, (1.4, 0.0, 1.0), 0),
}


def shading_points(rng, field_counts, side, origin, n):
    """Random shading points/normals spread over (and a little beyond) the probe grid."""
    counts = np.asarray(field_counts, dtype=np.float32)
    ext = counts * side * 0.6
    pos = (rng.uniform(-1, 1, size=(n, 3)).astype(np.float32) * ext + np.asarray(origin, dtype=np.float32))
    nrm = rng.normal(size=(n, 3)).astype(np.float32)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    return pos.astype(np.float32), nrm.astype(np.float32)
