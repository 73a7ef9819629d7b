"""A SECOND, independently written restatement of the two chaotic pieces of the probe path, checked against the oracle.

Parity with the reference is unpinned by the reference itself (it ships no vectors and its GLSL cannot run here,
DESIGN.md section 2), so a misreading of the GLSL in oracle/ddgi_oracle.c would go unnoticed by every HIP-vs-oracle
test.  This file restates, from the GLSL text alone and in a different language and style (numpy binary32 scalars,
one IEEE operation per GLSL operator, libm's sinf — i.e. the oracle's LITERAL arithmetic), the two functions where a
misreading would do the most damage:

  grid_march      assets/shaders/intersection.glsl:1051-1100   (the voxel traversal: step rule, voxel id, normal pick)
  fbm             assets/shaders/intersection.glsl:400-435     (noise2D -> interpNoise2D -> 8 octaves)
  getBlockAt      assets/shaders/intersection.glsl:758-791     (scene 1, the Cornell box, to drive the march)

and compares bit for bit with the oracle in LITERAL mode on random inputs.  Test infrastructure only."""
import ctypes as C

import numpy as np
import pytest

f32 = np.float32
_libm = C.CDLL("libm.so.6")
_libm.sinf.restype = C.c_float
_libm.sinf.argtypes = [C.c_float]
_libm.powf.restype = C.c_float
_libm.powf.argtypes = [C.c_float, C.c_float]


def _sin(x):
    return f32(_libm.sinf(float(x)))


def _fract(x):  # GLSL fract: x - floor(x)
    return f32(x - f32(np.floor(x)))


def _mix(a, b, t):  # GLSL mix: a * (1 - t) + b * t
    return f32(f32(a * f32(f32(1.0) - t)) + f32(b * t))


# ---- intersection.glsl:400-435 ------------------------------------------------------------------------
def noise2d(px, py):
    d = f32(f32(px * f32(127.1)) + f32(py * f32(311.7)))          # dot(p, vec2(127.1, 311.7))
    return _fract(f32(_sin(d) * f32(43758.5453)))


def interp_noise2d(x, y):
    ix, iy = int(np.floor(x)), int(np.floor(y))
    fx, fy = _fract(x), _fract(y)
    v1, v2 = noise2d(f32(ix), f32(iy)), noise2d(f32(ix + 1), f32(iy))
    v3, v4 = noise2d(f32(ix), f32(iy + 1)), noise2d(f32(ix + 1), f32(iy + 1))
    return _mix(_mix(v1, v2, fx), _mix(v3, v4, fx), fy)


def fbm(x, y):
    total = f32(0.0)
    for i in range(1, 9):
        freq = f32(_libm.powf(2.0, float(i)))
        amp = f32(_libm.powf(0.5, float(i)))
        total = f32(total + f32(interp_noise2d(f32(x * freq), f32(y * freq)) * amp))
    return total


# ---- intersection.glsl:758-791 ------------------------------------------------------------------------
def cornell_block(c):
    x, y, z = (float(v) for v in c)
    if x == -10 and abs(y) < 10 and abs(z - 15) < 10:
        return 2
    if x == 10 and abs(y) < 10 and abs(z - 15) < 10:
        return 3
    if abs(y) == 10 and abs(x) < 10 and abs(z - 15) < 10:
        return 5
    if z == 25 and abs(x) < 10 and abs(y) < 10:
        return 5
    if abs(x + 3) < 3 and abs(y + 7) < 3 and abs(z - 13) < 3:
        return 5
    if abs(x - 4) < 3 and abs(y + 4) < 6 and abs(z - 16) < 3:
        return 5
    return 0


# ---- intersection.glsl:1051-1100 ----------------------------------------------------------------------
def _gl_max(a, b):  # GLSL max(x, y) = x < y ? y : x
    return b if a < b else a


def _gl_min(a, b):  # GLSL min(x, y) = y < x ? y : x
    return b if b < a else a


def grid_march(origin, direction, block_at):
    o = [f32(v) for v in origin]
    d = [f32(v) for v in direction]
    length = f32(np.sqrt(f32(f32(f32(d[0] * d[0]) + f32(d[1] * d[1])) + f32(d[2] * d[2]))))
    rd = [f32(v / length) for v in d]                                # normalize(ray.direction)
    p = list(o)
    t = f32(0.0)
    with np.errstate(divide="ignore", invalid="ignore"):
        for it in range(1, 126):
            t2 = []
            for k in range(3):
                fr = _fract(p[k])
                t2.append(_gl_max(f32(f32(-fr) / rd[k]), f32(f32(f32(1.0) - fr) / rd[k])))
            step = f32(_gl_min(_gl_min(t2[0], t2[1]), t2[2]) + f32(0.0001))
            t = f32(t + step)
            p = [f32(o[k] + f32(rd[k] * t)) for k in range(3)]       # ray.origin + ray_dir * curr_t
            cell = [f32(np.ceil(v)) for v in p]
            block = block_at(cell)
            if block > 0:
                pi = [f32(c - f32(0.5)) for c in cell]
                diff = [f32(p[k] - pi[k]) for k in range(3)]
                dl = f32(np.sqrt(f32(f32(f32(diff[0] * diff[0]) + f32(diff[1] * diff[1])) + f32(diff[2] * diff[2]))))
                diff = [f32(v / dl) for v in diff]
                normal, best = [0.0, 0.0, 0.0], f32(0.0)
                for k in range(3):
                    if abs(diff[k]) > best:
                        best = abs(diff[k])
                        normal = [0.0, 0.0, 0.0]
                        normal[k] = float(np.sign(diff[k]))
                return block, it, t, normal, p
    return 0, 125, t, None, p


def _bits(x):
    return np.float32(x).view(np.uint32)


def test_fbm_restated_twice_agrees_bitwise(oracle):
    lib = oracle.lib()
    lib.oracle_fbm.restype = C.c_float
    lib.oracle_fbm.argtypes = [C.c_float, C.c_float]
    lib.oracle_interp_noise2d.restype = C.c_float
    lib.oracle_interp_noise2d.argtypes = [C.c_float, C.c_float]
    oracle.set_arith(False)  # LITERAL
    rng = np.random.default_rng(41)
    pts = np.concatenate([rng.uniform(-40, 40, (300, 2)), rng.uniform(-3, 3, (200, 2)), [[0.0, 0.0], [-0.5, 7.25], [12.0, -3.0]]]).astype(np.float32)
    for x, y in pts:
        assert _bits(interp_noise2d(f32(x), f32(y))) == _bits(lib.oracle_interp_noise2d(float(x), float(y))), (x, y)
        assert _bits(fbm(f32(x), f32(y))) == _bits(lib.oracle_fbm(float(x), float(y))), (x, y)
    # the cave floor rule that consumes it (intersection.glsl:726-742) sees the same values
    vals = np.array([fbm(f32(x * f32(0.058)), f32(y * f32(0.058))) for x, y in pts[:200]])
    assert 0.2 < vals.mean() < 0.8 and vals.std() > 0.02


@pytest.mark.parametrize("scene", [1, 0])
def test_grid_march_restated_twice_agrees_bitwise(oracle, scene):
    """Scene 1: this file's own Cornell getBlockAt.  Scene 0: the cave through the oracle's getBlockAt (the march logic
    on rough geometry; the cave's voxel function is checked against the bake in tests/test_host_parity.py)."""
    lib = oracle.lib()
    oracle.set_arith(False)  # LITERAL
    rng = np.random.default_rng(43 + scene)
    block_at = cornell_block if scene == 1 else (lambda c: lib.oracle_get_block_at(float(c[0]), float(c[1]), float(c[2]), 0))
    hits = misses = 0
    for _ in range(160 if scene == 1 else 120):
        if scene == 1:
            o = rng.uniform([-8.5, -8.5, 6.5], [8.5, 8.5, 23.5]).astype(np.float32)
        else:
            o = (np.array([0, 0, 0]) + rng.normal(size=3) * 9.0).astype(np.float32)
        d = rng.normal(size=3).astype(np.float32) * np.float32(rng.uniform(0.3, 3.0))   # not unit length: grid_march normalises
        block, it, t, normal, p = grid_march(o, d, block_at)
        o_block, o_it, out = oracle.grid_march(o, d, scene)
        assert (block, it) == (o_block, o_it), (o, d)
        if block:
            hits += 1
            assert _bits(t) == _bits(out[0]), (o, d)
            assert normal == [float(v) for v in out[1:4]], (o, d)
            assert [int(_bits(v)) for v in p] == [int(_bits(v)) for v in out[7:10]], (o, d)
        else:
            misses += 1
    assert hits > 50 and (scene == 0 or misses > 0)   # Cornell: the open front lets rays out


# ========================================================================================================
# The whole path a second time (tests/glsl_restated.py), against the oracle's LITERAL arithmetic, bit for bit.
# The checks are pure functions of a seed, run in worker processes (spawned: libgomp does not survive a fork).
# ========================================================================================================
import os
from concurrent.futures import ProcessPoolExecutor
import multiprocessing as mp

N_WORKERS = max(1, min(16, os.cpu_count() or 1))
N_POINT = 100_000                                               # inputs per per-point function
N_SCENE = 100_000 if (os.cpu_count() or 1) >= 8 else 24_000    # intersect_scene marches up to 125 voxels per input


def _oracle_literal():
    from oracle import oracle_py

    lib = oracle_py.lib()
    oracle_py.set_arith(False)
    lib.oracle_get_color_at.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    lib.oracle_intersect_sphere.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_void_p]
    lib.oracle_intersect_scene.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    return oracle_py, lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _u(x):
    return np.asarray(x, dtype=np.float32).view(np.uint32)


def _same(a, b):  # bit-equal, with any NaN equal to any NaN
    a, b = np.asarray(a, dtype=np.float32).ravel(), np.asarray(b, dtype=np.float32).ravel()
    return bool(np.all((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))))


def _check_block_at(args):
    seed, n = args
    from tests import glsl_restated as G

    _, lib = _oracle_literal()
    rng = np.random.default_rng(seed)
    seen = set()
    for k in range(n):
        scene = (0, 0, 0, 0, 0, 0, 1, 2)[k % 8]
        if scene == 0:   # the hollow, the rock around it, the fbm floor band below y = -15, the mushrooms' quadrants
            c = np.floor(rng.uniform([-48, -30, -44], [40, 22, 38])).astype(np.float32)
            if k % 5 == 0:
                c[1] = np.float32(rng.integers(-24, -14))
        elif scene == 1:
            c = np.floor(rng.uniform([-13, -13, 2], [13, 13, 28])).astype(np.float32)
        else:
            c = np.floor(rng.uniform([-28, -8, -18], [28, 8, 18])).astype(np.float32)
        got = G.get_block_at((c[0], c[1], c[2]), scene)
        want = lib.oracle_get_block_at(float(c[0]), float(c[1]), float(c[2]), scene)
        assert got == want, (scene, c, got, want)
        seen.add((scene, got))
    return sorted(seen)


_AXES = [(1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1)]


def _check_color_at(args):
    seed, n = args
    from tests import glsl_restated as G

    _, lib = _oracle_literal()
    rng = np.random.default_rng(seed)
    out = np.zeros(3, dtype=np.float32)
    for k in range(n):
        t = 1 + k % 13
        p = rng.uniform([-45, -24, -40], [36, 18, 34]).astype(np.float32)
        if k % 7 == 0:
            p = np.round(p * 2) / np.float32(2)         # lattice points and half-way points: the branches on exact values
        nrm = np.array(_AXES[rng.integers(0, 6)], dtype=np.float32)
        got = G.get_color_at((p[0], p[1], p[2]), t, (nrm[0], nrm[1], nrm[2]))
        lib.oracle_get_color_at(_p(p), t, _p(nrm), _p(out))
        assert _same(got, out), (t, p, nrm, got, out)
    return n


def _check_intersect_sphere(args):
    seed, n = args
    from tests import glsl_restated as G

    _, lib = _oracle_literal()
    rng = np.random.default_rng(seed)
    out = np.zeros(4, dtype=np.float32)
    hits = 0
    for k in range(n):
        o = (rng.normal(size=3) * rng.choice([0.5, 3.0, 40.0])).astype(np.float32)
        d = rng.normal(size=3).astype(np.float32) * np.float32(rng.choice([0.1, 1.0, 10.0]))
        if k % 3 == 0:
            d = (-o + rng.normal(size=3).astype(np.float32) * np.float32(0.7)).astype(np.float32)   # aimed at the sphere
        maxt = np.float32(np.inf) if k % 4 else np.float32(rng.uniform(0.1, 5.0))
        t, pos = G.intersect_sphere(tuple(o), tuple(d), np.float32(0.0), maxt)
        hit = lib.oracle_intersect_sphere(_p(o), _p(d), 0.0, float(maxt), _p(out))
        assert bool(hit) == bool(t < np.inf) and _same([t], out[:1]), (o, d, maxt, t, out)
        if hit:
            hits += 1
            assert _same(pos, out[1:4]), (o, d, pos, out)
    return hits


def _check_intersect_scene(args):
    seed, n = args
    from tests import glsl_restated as G

    oracle_py, lib = _oracle_literal()
    rng = np.random.default_rng(seed)
    out = np.zeros(10, dtype=np.float32)
    kinds = {0: 0, 2: 0, 3: 0}
    for k in range(n):
        scene = (0, 0, 0, 1, 2)[k % 5]
        st = oracle_py.make_settings(scene, 8)
        centre, spread = {0: ((-4, -2, -2), (14, 8, 14)), 1: ((0, 0, 15), (8, 8, 8)), 2: ((0, 0, 0), (20, 4, 12))}[scene]
        o = (np.array(centre) + rng.uniform(-1, 1, 3) * np.array(spread)).astype(np.float32)
        d = rng.normal(size=3).astype(np.float32) * np.float32(rng.choice([0.3, 1.0, 2.5]))   # not unit: info.t assumes it, grid_march normalises (Q16)
        if k % 4 == 0:   # at a light, from nearby: the sphere branch
            lp = np.array(G.LIGHTS[scene][0][2], dtype=np.float32)
            o = (lp + rng.normal(size=3) * 3.0).astype(np.float32)
            d = (lp - o + rng.normal(size=3) * 0.05).astype(np.float32)
        got = G.intersect_scene(tuple(o), tuple(d), scene)
        typ = lib.oracle_intersect_scene(C.byref(st), _p(o), _p(d), _p(out))
        assert (0 if got is None else got["type"]) == typ, (scene, o, d, got, typ)
        kinds[typ] += 1
        if got is not None:
            assert _same([got["t"]], out[0:1]) and _same(got["pos"], out[1:4]) and _same(got["normal"], out[4:7]) and _same(got["color"], out[7:10]), (scene, o, d, got, out)
    return kinds


def _check_ray_colors(args):
    """probe_pass.comp:main for arbitrary ray records: intersect_scene + get_direct_lighting + the hemisphere sample + the RNG,
    8 bounces — the per-ray float colour against the oracle's."""
    seed, n, scene = args
    from tests import glsl_restated as G

    oracle_py, _ = _oracle_literal()
    rng = np.random.default_rng(seed)
    counts, s = (2, 2, 2), 8
    f = oracle_py.make_field(counts, 6, s, (0.0, 0.0, 15.0))
    rays = np.zeros(counts[0] * counts[1] * counts[2] * s * s, dtype=oracle_py.RAY_DTYPE)
    centre, spread = {0: ((-4, -2, -2), (14, 8, 14)), 1: ((0, 0, 15), (8, 8, 8)), 2: ((0, 0, 0), (20, 4, 12))}[scene]
    first = int(rng.integers(0, len(rays) - n))
    for i in range(first, first + n):
        rays[i]["origin"][:3] = np.array(centre) + rng.uniform(-1, 1, 3) * np.array(spread)
        d = rng.normal(size=3)
        rays[i]["direction"][:3] = d / np.linalg.norm(d)
        rays[i]["probe_info"][:3] = (i // (s * s), i % s, (i // s) % s)
    _, _, colors = oracle_py.probe_update(f, oracle_py.make_settings(scene, 8), rays, first=first, count=n, want_float=True, nthreads=1)
    lit = 0
    for i in range(first, first + n):
        o = tuple(np.float32(v) for v in rays[i]["origin"][:3])
        d = tuple(np.float32(v) for v in rays[i]["direction"][:3])
        got = G.probe_ray_color(i, o, d, scene, 8)
        assert _same(got, colors[i - first]), (scene, i, o, d, got, colors[i - first])
        lit += int(any(c > 0 for c in got))
    return lit


def _check_sample(args):
    seed, n = args
    from tests import glsl_restated as G

    oracle_py, _ = _oracle_literal()
    rng = np.random.default_rng(seed)
    counts, side, s, origin = (5, 3, 4), 3, 6, (1.4, 0.0, 1.0)     # odd and even counts: the sampler's shift is counts / 2 (Q4)
    f = oracle_py.make_field(counts, side, s, origin)
    W, H = counts[0] * counts[2] * s, counts[1] * s
    albedo = rng.integers(0, 256, size=(H, W, 4), dtype=np.uint8)
    distance = rng.integers(0, 256, size=(H, W, 4), dtype=np.uint8)
    pos = (np.array(origin) + rng.uniform(-1, 1, (n, 3)) * np.array(counts) * side * 0.62).astype(np.float32)
    nrm = rng.normal(size=(n, 3)).astype(np.float32)
    nrm[::11] = np.array(_AXES, dtype=np.float32)[rng.integers(0, 6, size=len(nrm[::11]))]   # axis normals: |z| = 1 makes the texel inversion NaN (Q8)
    rgb, cage = oracle_py.sample(f, albedo, distance, pos, nrm)
    inside = 0
    for i in range(n):
        got, got_cage = G.get_diffuse_gi(tuple(pos[i]), tuple(nrm[i]), albedo, distance, counts, side, tuple(np.float32(v) for v in origin), s)
        assert list(cage[i]) == got_cage, (i, pos[i], cage[i], got_cage)
        assert _same(got, rgb[i]), (i, pos[i], nrm[i], got, rgb[i])
        inside += int(got_cage[0] >= 0)
    return inside


def _spread(fn, total, extra=()):
    """fn over `total` inputs in N_WORKERS processes; the per-chunk results."""
    per = (total + N_WORKERS - 1) // N_WORKERS
    jobs = [(1000 + w, per) + tuple(extra) for w in range(N_WORKERS)]
    with ProcessPoolExecutor(max_workers=N_WORKERS, mp_context=mp.get_context("spawn")) as pool:
        return list(pool.map(fn, jobs))


def test_get_block_at_all_scenes_restated_twice(oracle):
    seen = set()
    for chunk in _spread(_check_block_at, N_POINT):
        seen.update(tuple(x) for x in chunk)
    # every block type of the cave was met (10 wall, 11-13 floor band, 6-9 mushrooms), and both other scenes' walls
    assert {t for sc, t in seen if sc == 0} >= {0, 6, 7, 8, 9, 10, 11, 12, 13}, sorted(seen)
    assert {t for sc, t in seen if sc == 1} >= {0, 2, 3, 5} and {t for sc, t in seen if sc == 2} >= {0, 1, 2, 3, 5}


def test_get_color_at_all_13_types_restated_twice(oracle):
    assert sum(_spread(_check_color_at, N_POINT)) >= N_POINT


def test_intersect_sphere_restated_twice(oracle):
    assert sum(_spread(_check_intersect_sphere, N_POINT)) > N_POINT // 20


def test_intersect_scene_restated_twice(oracle):
    kinds = {0: 0, 2: 0, 3: 0}
    for chunk in _spread(_check_intersect_scene, N_SCENE):
        for k, v in chunk.items():
            kinds[k] += v
    assert kinds[3] > N_SCENE // 3 and kinds[2] > N_SCENE // 50 and kinds[0] > 0, kinds


@pytest.mark.parametrize("scene", [0, 1])
def test_whole_rays_direct_lighting_hemisphere_rng_restated_twice(oracle, scene):
    per = 24 if N_WORKERS >= 8 else 8
    jobs = [(2000 + 17 * w + scene, per, scene) for w in range(N_WORKERS)]
    with ProcessPoolExecutor(max_workers=N_WORKERS, mp_context=mp.get_context("spawn")) as pool:
        lit = sum(pool.map(_check_ray_colors, jobs))
    assert lit > per * N_WORKERS // 4


def test_sample_probe_and_get_diffuse_gi_restated_twice(oracle):
    inside = sum(_spread(_check_sample, N_POINT))
    assert N_POINT // 10 < inside < N_POINT


def test_one_whole_c1_update_through_the_second_restatement(oracle):
    """BASELINE config C1 (Cornell box, 2x2x2 probes x 64 rays, 8 bounces): all 512 rays of the update through the numpy path,
    rows of probes in parallel — the rgba8 raster must equal the oracle's LITERAL one byte for byte."""
    oracle.set_arith(False)
    counts, side, s, origin, scene = (2, 2, 2), 6, 8, (0.0, 0.0, 15.0), 1
    f = oracle.make_field(counts, side, s, origin)
    rays = oracle.generate_probe_rays(f, oracle.new_rand_state(1))
    want = oracle.probe_update(f, oracle.make_settings(scene, 8), rays)[0]
    chunks = [(rays, k, min(k + 32, len(rays))) for k in range(0, len(rays), 32)]
    with ProcessPoolExecutor(max_workers=N_WORKERS, mp_context=mp.get_context("spawn")) as pool:
        parts = list(pool.map(_c1_chunk, chunks))
    got = np.zeros_like(want)
    for part in parts:
        mask = part[..., 3] == 255
        got[mask] = part[mask]
    assert (got[..., 3] == 255).all() and got[..., :3].any()
    assert np.array_equal(got, want), f"{int((got != want).any(axis=-1).sum())} of 512 texels differ"


def _c1_chunk(args):
    rays, lo, hi = args
    from tests import glsl_restated as G

    counts, s, scene = (2, 2, 2), 8, 1
    width = counts[0] * counts[2] * s
    albedo = np.zeros((counts[1] * s, width, 4), dtype=np.uint8)
    for index in range(lo, hi):
        r = rays[index]
        o = tuple(np.float32(v) for v in r["origin"][:3])
        d = tuple(np.float32(v) for v in r["direction"][:3])
        info = r["probe_info"]
        probe = int(info[0])
        y_probe = probe // (counts[0] * counts[2])
        x_probe = probe - y_probe * counts[0] * counts[2]
        c = G.probe_ray_color(index, o, d, scene, 8)
        albedo[y_probe * s + int(info[2]), x_probe * s + int(info[1])] = (G.unorm8(c[0]), G.unorm8(c[1]), G.unorm8(c[2]), 255)
    return albedo


# ---- the dormant pieces DDGI mode switches on (rows a18-a20) ---------------------------------------------------------------
f32 = np.float32


def test_update_lights_restated_twice(oracle):
    """probe_pass.comp:217-251 for the three shipped tables, the commented four-light cave table (structs.glsl:65-68: light
    indices up to 3 exercise (i + 1) * 2 and (i / 2) * 4) and times from a frame's +2 steps to hours of them."""
    from tests import glsl_restated as G
    oracle.set_arith(False)
    try:
        rng = np.random.default_rng(31)
        times = np.concatenate([np.arange(0.0, 400.0, 2.0), rng.uniform(0.0, 1.0e6, 1500), [1.0e7, 3.0e8]]).astype(np.float32)
        four = np.zeros(4, dtype=oracle.LIGHT_DTYPE)
        four["intensity"] = 40.0
        four["col"] = 1.0
        four["pos"] = [(4, 17.5, 8.5), (-14, 12, 5), (20, 10, -18), (-2, 22, 30)]
        n = 0
        for scene in (0, 1, 2):
            for base in (oracle.shipped_lights(scene), four):
                table = [(f32(b["intensity"]), tuple(f32(c) for c in b["col"]), tuple(f32(c) for c in b["pos"])) for b in base]
                for t in times:
                    want = oracle.update_lights(scene, float(t), base)["pos"]
                    got = np.array(G.update_lights(scene, t, table), dtype=np.float32)
                    assert np.array_equal(want.view(np.uint32), got.view(np.uint32)), (scene, float(t), want, got)
                    n += len(base)
        assert n > 10000
    finally:
        oracle.set_arith(True)


def test_oct_encode_decode_restated_twice(oracle):
    """octahedral.glsl:16-34 on 10^5 unit vectors / points of the [-1, 1] square (axes, octant borders, the folded half)."""
    from tests import glsl_restated as G
    oracle.set_arith(False)
    try:
        rng = np.random.default_rng(37)
        v = rng.normal(size=(N_POINT, 3))
        v /= np.linalg.norm(v, axis=1, keepdims=True)
        v = v.astype(np.float32)
        v[:6] = [(1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1)]
        v[6:200, 2] *= 0.0           # the fold line z = 0
        v[200:400, 0] = 0.0          # signNotZero(+-0)
        v[300:400, 0] = -0.0
        enc = oracle.oct_encode(v)
        uv = rng.uniform(-1.0, 1.0, size=(N_POINT, 2)).astype(np.float32)
        uv[:4] = [(1, 1), (-1, 1), (0, 0), (1, -1)]
        uv[4:300, 0] = 0.0
        dec = oracle.oct_decode(uv)
        step = max(1, N_POINT // 20000)   # (python scalars: a spread 20 000 of each, the special rows all)
        rows = sorted(set(range(0, 400)) | set(range(0, N_POINT, step)))
        for i in rows:
            e = G.oct_encode(tuple(f32(c) for c in v[i]))
            assert _bits(e[0]) == _bits(enc[i, 0]) and _bits(e[1]) == _bits(enc[i, 1]), (i, v[i], e, enc[i])
            d = G.oct_decode(tuple(f32(c) for c in uv[i]))
            assert all(_bits(d[k]) == _bits(dec[i, k]) for k in range(3)), (i, uv[i], d, dec[i])
        # and the pair is what it claims to be: decode(encode(v)) == v to rounding
        back = oracle.oct_decode(enc)
        assert np.abs(back[400:] - v[400:]).max() < 1e-6   # (the rows above were bent off the unit sphere on purpose)
    finally:
        oracle.set_arith(True)
