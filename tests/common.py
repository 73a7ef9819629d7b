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
_CACHE_DIR = "/tmp/ddgi_oracle_cache"


def c3_oracle_albedo(oracle, arith, seed=1, lights=None, fraction=1):
    """The oracle's albedo raster of the C3 grid (4 194 304 texels; from seconds to a minute and a half on the GPU box's host
    threads, depending on the box), computed once per (arithmetic, ray seed, light table, fraction) — per test session in memory, and
    per box on disk (/tmp), so that the tests and a second run of the suite share it.
      arith     "pinned" (what the kernels implement bit for bit) or "literal" (one IEEE operation per GLSL operator, libm — the
                closest this repository gets to the reference's own semantics)
      lights    None = the scene's shipped table, else an array of oracle.LIGHT_DTYPE
      fraction  1 = every probe; n > 1 = one n-th of the probes, in 64 runs of consecutive probes spread evenly over the grid
                (the other tiles stay zero: compare where alpha == 255)
    One full-grid comparison per distinct kernel path is kept (tests/test_gpu_parity.py, test_gpu_fast_march.py,
    test_gpu_edge_cases.py); further ray sets and light tables use a fraction."""
    import hashlib
    import os

    lkey = "shipped" if lights is None else hashlib.sha1(np.ascontiguousarray(lights).tobytes()).hexdigest()[:16]
    key = (arith, seed, lkey, fraction)
    if key in _C3_ORACLE:
        return _C3_ORACLE[key]
    # (the cache is only as good as the oracle it came from: its library's bytes are part of the key)
    from oracle import oracle_py

    with open(oracle_py.build(), "rb") as fh:
        okey = hashlib.sha1(fh.read()).hexdigest()[:16]
    path = os.path.join(_CACHE_DIR, "c3_%s_%d_%s_%d_%s.npy" % (arith, seed, lkey, fraction, okey))
    if os.path.exists(path):
        try:
            _C3_ORACLE[key] = np.load(path)
            return _C3_ORACLE[key]
        except Exception:
            pass
    counts, side, s, origin, scene = CONFIGS["c3_cave"]
    oracle.set_arith(arith == "pinned")
    try:
        f = oracle.make_field(counts, side, s, origin)
        rays = oracle.generate_probe_rays(f, oracle.new_rand_state(seed))
        st = oracle.make_settings(scene, 8)
        if fraction == 1:
            out = oracle.probe_update(f, st, rays, lights=lights)[0]
        else:
            n_probes, n = counts[0] * counts[1] * counts[2], s * s
            runs = 64
            per = max(1, n_probes // fraction // runs)
            out = None
            for r in range(runs):
                first = (r * n_probes // runs + (seed * 37) % max(1, n_probes // runs - per)) * n
                part = oracle.probe_update(f, st, rays, first=first, count=per * n, lights=lights)[0]
                out = part if out is None else np.maximum(out, part)
    finally:
        oracle.set_arith(True)
    _C3_ORACLE[key] = out
    try:
        os.makedirs(_CACHE_DIR, exist_ok=True)
        np.save(path + ".tmp.npy", out)
        os.replace(path + ".tmp.npy", path)
    except Exception:
        pass
    return out


def texel_tolerance_stats(got, want):
    """(fraction of rgb channels within one unorm8 step, mean |difference| in unorm8 steps, texels that differ at all)"""
    d = np.abs(got[..., :3].astype(np.int32) - want[..., :3].astype(np.int32))
    return float((d <= 1).mean()), float(d.mean()), int((d.max(axis=-1) > 0).sum())
