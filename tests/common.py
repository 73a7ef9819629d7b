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


_C3_ORACLE = {}


def c3_oracle_albedo(oracle, arith, seed=1):
    """The oracle's albedo raster of the full C3 grid (4 194 304 texels; seconds on the GPU box's host threads), computed
    once per test session and arithmetic: "pinned" (what the kernels implement bit for bit) or "literal" (one IEEE operation
    per GLSL operator, libm — the closest this repository gets to the reference's own semantics)."""
    key = (arith, seed)
    if key not in _C3_ORACLE:
        counts, side, s, origin, scene = CONFIGS["c3_cave"]
        oracle.set_arith(arith == "pinned")
        try:
            f = oracle.make_field(counts, side, s, origin)
            rays = oracle.generate_probe_rays(f, oracle.new_rand_state(seed))
            _C3_ORACLE[key] = oracle.probe_update(f, oracle.make_settings(scene, 8), rays)[0]
        finally:
            oracle.set_arith(True)
    return _C3_ORACLE[key]


def texel_tolerance_stats(got, want):
    """(fraction of rgb channels within one unorm8 step, mean |difference| in unorm8 steps, texels that differ at all)"""
    d = np.abs(got[..., :3].astype(np.int32) - want[..., :3].astype(np.int32))
    return float((d <= 1).mean()), float(d.mean()), int((d.max(axis=-1) > 0).sum())
