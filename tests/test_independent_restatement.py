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
