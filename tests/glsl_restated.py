"""A SECOND restatement of the reference's probe path, written from the GLSL text alone (test infrastructure).

oracle/ddgi_oracle.c is the first; it is what every HIP-vs-oracle test trusts, and the reference can pin neither (it ships
no vectors and its GLSL cannot run here — DESIGN.md section 2).  This module restates the same functions a second time, in
another language and another style — numpy binary32 SCALARS, one IEEE operation per GLSL operator in source order, libm's
sinf / cosf / acosf / powf through ctypes (i.e. the oracle's LITERAL arithmetic) — so that a misreading of the GLSL has to be
made twice, independently, to go unnoticed.  tests/test_independent_restatement.py holds the two against each other bit for bit.

GLSL built-ins are taken by their definitions in the GLSL 4.50 specification (8.1-8.5):
    fract(x) = x - floor(x)          mod(x, y) = x - y * floor(x / y)     mix(a, b, t) = a * (1 - t) + b * t
    clamp(x, lo, hi) = min(max(x, lo), hi)      min(x, y) = y < x ? y : x      max(x, y) = x < y ? y : x
    dot(a, b) = a.x * b.x + a.y * b.y + ...     length(v) = sqrt(dot(v, v))    distance(a, b) = length(a - b)
    normalize(v) = v / length(v)                sign(x) in {-1, 0, 1}
Where the reference leaves a value undefined the same pin as everywhere in this repository applies: the
material of a light-sphere hit is zero, int(NaN) = 0, rgba8 stores round to nearest-even and map NaN to 0.

Covered (file:line of /root/reference/assets/shaders):
    intersection.glsl:78-121     intersect_sphere          :321, :538-542   sdSphere, sdRoundBox
    intersection.glsl:400-499    random1, noise2D, interpNoise2D, fbm, noise, interpNoise1D, fbm1D, generate_point, worleyNoise
    intersection.glsl:544-697    tiny / small / medium / large_mushroom, all_mushrooms
    intersection.glsl:699-826    getBlockAt (cave, Cornell, house)           :828-870  getUVs, dotsPattern
    intersection.glsl:872-1047   getColorAt, all 13 block types              :1051-1100 grid_march
    intersection.glsl:1152-1240  get_text_coord_from_probe_number, sample_probe
    intersection.glsl:1244-1301  intersect_scene                             :1306-1409 get_diffuse_gi
    probe_pass.comp:45-71        wang_hash, rand_xorshift, rand              :150-178  calculate_random_dir_hemisphere
    probe_pass.comp:180-215      get_direct_lighting                         :253-303  main
    structs.glsl:54-89           the shipped light tables
The dormant pieces DDGI mode switches on, restated the same way:
    probe_pass.comp:217-251      update_lights             :298-299  mix(old, new, hysteresis)
    octahedral.glsl:16-34        octEncode, octDecode (signNotZero: g3dmath's, +1 for x >= 0 else -1)
"""
import ctypes as C

import numpy as np

f32 = np.float32
_libm = C.CDLL("libm.so.6")
for _name, _n in (("sinf", 1), ("cosf", 1), ("acosf", 1), ("powf", 2)):
    getattr(_libm, _name).restype = C.c_float
    getattr(_libm, _name).argtypes = [C.c_float] * _n

INF = f32(np.inf)
ZERO, ONE, HALF = f32(0.0), f32(1.0), f32(0.5)


def sin(x):
    return f32(_libm.sinf(float(x)))


def cos(x):
    return f32(_libm.cosf(float(x)))


def acos(x):
    return f32(_libm.acosf(float(x)))


def powf(x, y):
    return f32(_libm.powf(float(x), float(y)))


def floor(x):
    return f32(np.floor(x))


def ceil(x):
    return f32(np.ceil(x))


def sqrt(x):
    with np.errstate(invalid="ignore"):
        return f32(np.sqrt(f32(x)))


def fract(x):
    return f32(x - floor(x))


def gmin(x, y):
    return y if y < x else x


def gmax(x, y):
    return y if x < y else x


def clamp(x, lo, hi):
    return gmin(gmax(x, lo), hi)


def mix(a, b, t):
    return f32(f32(a * f32(ONE - t)) + f32(b * t))


def sign(x):
    return f32(1.0) if x > 0 else (f32(-1.0) if x < 0 else f32(0.0))


def mod(x, y):
    return f32(x - f32(y * floor(f32(x / y))))


def to_int(x):  # int(float): truncation; int(NaN) pinned to 0
    if np.isnan(x):
        return 0
    if np.isinf(x):
        return 2147483647 if x > 0 else -2147483648
    return int(x)


# ---- vectors as tuples of binary32 scalars -------------------------------------------------------------
def v3(x, y, z):
    return (f32(x), f32(y), f32(z))


def vadd(a, b):
    return tuple(f32(p + q) for p, q in zip(a, b))


def vsub(a, b):
    return tuple(f32(p - q) for p, q in zip(a, b))


def vmul(a, b):
    return tuple(f32(p * q) for p, q in zip(a, b))


def vscale(a, s):
    return tuple(f32(p * s) for p in a)


def vdiv(a, s):
    with np.errstate(divide="ignore", invalid="ignore"):
        return tuple(f32(p / s) for p in a)


def dot(a, b):
    acc = f32(a[0] * b[0])
    for p, q in zip(a[1:], b[1:]):
        acc = f32(acc + f32(p * q))
    return acc


def length(a):
    return sqrt(dot(a, a))


def normalize(a):
    return vdiv(a, length(a))


def cross(a, b):
    return (f32(f32(a[1] * b[2]) - f32(b[1] * a[2])), f32(f32(a[2] * b[0]) - f32(b[2] * a[0])), f32(f32(a[0] * b[1]) - f32(b[0] * a[1])))


def vmix(a, b, t):
    return tuple(mix(p, q, t) for p, q in zip(a, b))


# ---- intersection.glsl:321, 538-542 ---------------------------------------------------------------------
def sd_sphere(p, s):
    return f32(length(p) - f32(s))


def sd_round_box(p, b, r):
    q = tuple(f32(abs(pc) - f32(bc)) for pc, bc in zip(p, b))
    outside = length(tuple(gmax(c, ZERO) for c in q))
    inside = gmin(gmax(q[0], gmax(q[1], q[2])), ZERO)
    return f32(f32(outside + inside) - f32(r))


# ---- intersection.glsl:400-499: the hash noises ------------------------------------------------------------
K = f32(43758.5453)


def random1(p):
    return fract(f32(sin(dot(p, v3(127.1, 311.7, 191.999))) * K))


def noise2d(p):
    return fract(f32(sin(dot(p, (f32(127.1), f32(311.7)))) * K))


def interp_noise2d(x, y):
    ix, fx = to_int(floor(x)), fract(x)
    iy, fy = to_int(floor(y)), fract(y)
    v1 = noise2d((f32(ix), f32(iy)))
    v2 = noise2d((f32(ix + 1), f32(iy)))
    v3_ = noise2d((f32(ix), f32(iy + 1)))
    v4 = noise2d((f32(ix + 1), f32(iy + 1)))
    return mix(mix(v1, v2, fx), mix(v3_, v4, fx), fy)


def fbm(x, y):
    total = ZERO
    for i in range(1, 9):
        freq = powf(2.0, i)
        amp = powf(0.5, i)
        total = f32(total + f32(interp_noise2d(f32(x * freq), f32(y * freq)) * amp))
    return total


def noise1(i):
    # fract(sin(vec2(203.311 * i, i * sin(0.324 + 140.0 * i)))).x — only the x component is kept
    return fract(sin(f32(f32(203.311) * i)))


def interp_noise1d(x):
    ix, fx = floor(x), fract(x)
    return mix(noise1(ix), noise1(f32(ix + ONE)), fx)


def fbm1d(x):
    total = ZERO
    for i in range(8):
        freq = powf(2.0, float(i))
        amp = powf(0.5, float(i))
        total = f32(total + f32(interp_noise1d(f32(x * freq)) * amp))
    return total


CELL = f32(5.0)


def generate_point(cell):
    p = (f32(cell[0]), f32(cell[1]))
    a = dot(p, (f32(127.1), f32(311.7)))
    b = f32(dot(p, (f32(269.5), f32(183.3))) * K)   # the factor sits INSIDE the sine's argument for y, and nowhere for x (:467)
    p = (f32(p[0] + fract(sin(a))), f32(p[1] + fract(sin(b))))
    return (f32(p[0] * CELL), f32(p[1] * CELL))


def worley(pixel):
    cell = (floor(f32(pixel[0] / CELL)), floor(f32(pixel[1] / CELL)))
    shortest = length(vsub(pixel, generate_point(cell)))
    for i in (-1.0, 0.0, 1.0):
        nx = f32(cell[0] + f32(i))
        for j in (-1.0, 0.0, 1.0):
            ny = f32(cell[1] + f32(j))
            d = length(vsub(pixel, generate_point((nx, ny))))
            if d < shortest:
                shortest = d
    return f32(shortest / CELL)


# ---- intersection.glsl:544-697: the mushrooms ---------------------------------------------------------------
def tiny_mushroom(p):
    if sd_round_box(p, (1.0, 0.5, 1.0), 0.0) <= 0:
        return 7
    if p[0] == 0 and p[2] == 0 and p[1] < 0:
        return 9
    return 0


def small_mushroom(p):
    if sd_round_box(p, (1.0, 0.5, 1.0), 1.0) <= 0:
        if p[1] > 0:
            return 8
        if p[1] == 0:
            return 7
        if p[1] < 0:
            return 6
    if p[0] == 0 and p[2] == 0 and p[1] < 0:
        return 9
    return 0


def medium_mushroom(p):
    if sd_round_box(p, (2.0, 0.5, 2.0), 1.0) <= 0:
        if p[1] > 0:
            return 6
        if p[1] == 0:
            return 7
        if p[1] < 0:
            return 8
    if p[0] == 0 and p[2] == 0 and p[1] < 0 and p[1] > -7:
        return 9
    if p[0] == 1 and p[2] == 0 and p[1] < -5 and p[1] > -12:
        return 9
    if p[0] == 2 and p[2] == 0 and p[1] < -10:
        return 9
    return 0


def large_mushroom(p, direction):
    if sd_round_box(p, (3.0, 0.5, 3.0), 1.5) <= 0:
        if p[1] > 0:
            return 6
        if p[1] == 0:
            return 8
        if p[1] < 0:
            return 7
    if p[0] == 0 and p[2] == 0 and p[1] < 0 and p[1] > -9:
        return 9
    if p[0] == 0 and p[2] == direction and p[1] < -7 and p[1] > -18:
        return 9
    if p[0] == 0 and p[2] == 2 * direction and p[1] < -16:
        return 9
    return 0


def all_mushrooms(c):
    def at(x, y, z):
        return vsub(c, v3(x, y, z))

    x, z = c[0], c[2]
    if x < 0 and z > 0:
        if x < -16:
            if z > 20:
                return tiny_mushroom(at(-19, -12, 22))
            if z < 4:
                return tiny_mushroom(at(-18, -12, 2))
            check = large_mushroom(at(-22, 3, 8), -1)
            if check != 0:
                return check
            check = medium_mushroom(at(-27, -4, 16))
            if check != 0:
                return check
            return 0
        if z > 10 and x > -6:
            return tiny_mushroom(at(-4, -14, 12))
        if z < 14:
            return medium_mushroom(at(-4, -1, 6))
        return small_mushroom(at(-10, -8, 18))
    if x < 0 and z < 0:
        if x < -16:
            if x < -28:
                if z < -16:
                    return tiny_mushroom(at(-32, -14, -20))
                return tiny_mushroom(at(-30, -12, -12))
            if z > -10:
                return small_mushroom(at(-25, -7, -4))
            return medium_mushroom(at(-20, -3, -20))
        if x < -12 and z > -12:
            return tiny_mushroom(at(-14, -15, -10))
        if z > -10 and x > -4:
            return tiny_mushroom(at(-2, -12, -2))
        if z < -10:
            return small_mushroom(at(-5, -9, -14))
        return large_mushroom(at(-8, 8, -6), 1)
    if x > 0 and z < 0:
        if z > -5:
            return tiny_mushroom(at(6, -14, -3))
        if z < -14:
            if x > 18:
                return tiny_mushroom(at(20, -7, -16))
            return large_mushroom(at(14, 10, -20), -1)
        return medium_mushroom(at(6, -6, -10))
    return 0


# ---- intersection.glsl:699-826 ----------------------------------------------------------------------------
def get_block_at(c, scene):
    x, y, z = c
    if scene == 0:
        if y > 17.0:
            return 0
        if y < -15:
            if y < -18:
                r = fbm(f32(x * f32(0.3)), f32(z * f32(0.3)))
                if to_int(floor(f32(r * f32(2.0)))) == 0:
                    return 12
            r = fbm(f32(x * f32(0.058)), f32(z * f32(0.058)))
            d = to_int(floor(f32(r * f32(5.0))))
            if f32(-21 + d) >= y:
                if y == -18:
                    return 13
                return 11
        if sd_sphere(c, 20.0) > 0.0:
            if sd_sphere(vadd(c, v3(16, 8, -10)), 20.0) > 0.0:
                if sd_sphere(vadd(c, v3(-13, -1, 19)), 18.0) > 0.0:
                    if sd_sphere(vadd(c, v3(20, 15, 15)), 21.0) > 0.0:
                        return 10
        return all_mushrooms(c)
    if scene == 1:
        if x == -10 and abs(y) < 10 and abs(f32(z - f32(15))) < 10:
            return 2
        if x == 10 and abs(y) < 10 and abs(f32(z - f32(15))) < 10:
            return 3
        if abs(y) == 10 and abs(x) < 10 and abs(f32(z - f32(15))) < 10:
            return 5
        if z == 25 and abs(x) < 10 and abs(y) < 10:
            return 5
        if abs(f32(x + f32(3))) < 3 and abs(f32(y + f32(7))) < 3 and abs(f32(z - f32(13))) < 3:
            return 5
        if abs(f32(x - f32(4))) < 3 and abs(f32(y + f32(4))) < 6 and abs(f32(z - f32(16))) < 3:
            return 5
        return 0
    if scene == 2:
        if y == -5:
            return 1
        if abs(x) == 25 and abs(y) < 5 and abs(z) < 15:
            return 2
        if y == 5 and abs(x) < 25 and abs(z) < 15:
            return 5
        if z == -15 and abs(x) < 25 and abs(y) < 5:
            return 3
        if z == 15:
            if abs(f32(x - f32(10))) < 2 and abs(f32(y + f32(1))) < 4:
                return 0
            if abs(x) < 25 and abs(y) < 5:
                return 3
        return 0
    return 0


# ---- intersection.glsl:828-870 ----------------------------------------------------------------------------
def get_uvs(p, n):
    if n[1] == 0:
        if n[0] == 0:
            if sign(n[2]) > 0:
                return (f32(ceil(p[0]) - p[0]), f32(p[1] - floor(p[1])))
            return (f32(p[0] - floor(p[0])), f32(p[1] - floor(p[1])))
        if sign(n[0]) < 1:
            return (f32(ceil(p[2]) - p[2]), f32(p[1] - floor(p[1])))
        return (f32(p[2] - floor(p[2])), f32(p[1] - floor(p[1])))
    if sign(n[1]) < 0:
        return (f32(p[0] - floor(p[0])), f32(ceil(p[2]) - p[2]))
    return (f32(p[0] - floor(p[0])), f32(p[2] - floor(p[2])))


def dots_pattern(point, radius, cell_size):
    c = f32(f32(f32(4.0) * radius) * cell_size)
    h = f32(c / f32(2.0))
    q = tuple(f32(mod(f32(pc + h), c) - h) for pc in point)
    return f32(length(q) - radius)


# ---- intersection.glsl:872-1047 ---------------------------------------------------------------------------
def get_color_at(point, block_type, normal):
    px, py, pz = point
    t = block_type
    if t == 1:
        r = f32(0.3)   # (the random1 value is overwritten on the next line, :892-893)
        if px < 0 and pz > 0:
            if px < -16:
                return v3(0.8, 0.4, 0.2)
            return (f32(0.1), r, f32(0.2))
        if px < 0 and pz < 0:
            if px < -16:
                return v3(0.4, 0.8, 0.2)
            return (f32(0.99), r, r)
        if px > 0 and pz < 0:
            return (f32(0.1), r, f32(0.5))
        return (f32(0.99), r, r)
    if t == 2:
        return v3(0.95, 0, 0)
    if t == 3:
        return v3(0, 0.95, 0)
    if t == 4:
        return v3(0, 0, 0.95)
    if t == 5:
        return v3(0.95, 0.95, 0.95)
    if t == 6:
        w = worley((px, pz))
        if w < f32(0.35):
            return v3(1, 0, 0.223)
        return v3(1, 0.2, 0)
    if t == 7:
        green = v3(0.8, 1, 0)
        w = worley((f32(px + f32(5)), f32(pz + f32(5))))
        if w < f32(0.25):
            return vsub(green, vscale(vsub(v3(0.5, 0.5, 0.5), green), w))
        return v3(1, 0, 0.011)
    if t == 8:
        light_orange, dark_purple = v3(1, 0.313, 0), v3(1, 0, 0.223)
        g = get_uvs(point, normal)
        m = f32(0.707)
        # mat2(0.707, -0.707, 0.707, 0.707) is column major: columns (0.707, -0.707) and (0.707, 0.707)
        uv = (f32(f32(m * g[0]) + f32(m * g[1])), f32(f32(f32(-0.707) * g[0]) + f32(m * g[1])))
        radius = f32(0.05)
        dist = dots_pattern(uv, radius, f32(1.8))
        circle = f32(f32(radius - dist) * f32(100.0))
        alpha = clamp(circle, ZERO, ONE)
        return vmix(light_orange, dark_purple, alpha)
    if t == 9:
        uvs = get_uvs(point, normal)
        val = fbm(f32(uvs[0] * f32(5)), pz)
        val = f32(val + f32(HALF * fbm1d(px)))
        val = clamp(val, ZERO, ONE)
        return vmix(v3(0.3, 0.1, 0.3), v3(0.9, 0.9, 0.9), val)
    if t == 10:
        color = v3(0.568, 0.133, 0.439)
        if py < -8:
            color = v3(0.349, 0.133, 0.427)
        elif py < -6:
            color = v3(0.568, 0.133, 0.439)
        elif py < -5:
            color = v3(0.639, 0.176, 0.725)
        elif py < 0:
            color = v3(0.274, 0.188, 0.772)
        elif py < 4:
            color = v3(0.341, 0.270, 0.768)
        elif py < 6:
            color = v3(0.368, 0.203, 0.415)
        elif py < 11:
            color = v3(0.470, 0.270, 0.729)
        uv = get_uvs(point, normal)
        r = fbm(f32(0.05), f32(f32(uv[1] + py) * f32(0.3)))
        wall = v3(0, 0.666, 1)
        if px < -1:
            wall = v3(0.294, 0.007, 0.152)
        elif px < 6 and px >= -1:
            gradient = f32(px / f32(7.0))
            rr = random1((ceil(px), ceil(py), ceil(pz)))
            wall = v3(0, 0.666, 1) if rr < gradient else v3(0.294, 0.007, 0.152)
        return vmix(wall, color, r)
    if t == 11:
        color, mold = v3(0.294, 0.007, 0.152), v3(0.901, 0.992, 0.427)
        r = f32(random1((ceil(px), ceil(py), ceil(pz))) / f32(3))
        combined = vmix(color, mold, r)
        uv = get_uvs(point, normal)
        r = fbm(f32(uv[0] * f32(2.0)), f32(uv[1] * f32(2.0)))
        return vmix(combined, v3(0.294, 0.007, 0.152), f32(r / f32(2.0)))
    if t in (12, 13):
        uv = get_uvs(point, normal)
        base_green = v3(0.356, 1, 0.101) if t == 12 else v3(0.803, 1, 0.341)
        base_purple = v3(0.619, 1, 0.278)
        off = (f32(uv[0] - HALF), f32(uv[1] - HALF))
        axis = normalize(off)
        r = interp_noise2d(axis[0], axis[1])
        tt = f32(f32(f32(2.0) * length(off)) + f32(r * f32(0.3)))   # distance(uv, vec2(0.5)) = length(uv - 0.5)
        return vmix(base_green, base_purple, tt)
    raise ValueError(block_type)


# ---- intersection.glsl:1051-1100 --------------------------------------------------------------------------
def grid_march(origin, direction, scene):
    """-> None | dict(t, normal, color, pos_march)"""
    rd = normalize(direction)
    p = origin
    t = ZERO
    with np.errstate(divide="ignore", invalid="ignore"):
        for _ in range(125):
            t2 = []
            for k in range(3):
                fr = fract(p[k])
                t2.append(gmax(f32(f32(-fr) / rd[k]), f32(f32(ONE - fr) / rd[k])))
            t = f32(t + f32(gmin(gmin(t2[0], t2[1]), t2[2]) + f32(0.0001)))
            p = vadd(origin, vscale(rd, t))
            cell = tuple(ceil(v) for v in p)
            block = get_block_at(cell, scene)
            if block > 0:
                pi = tuple(f32(c - HALF) for c in cell)
                diff = normalize(vsub(p, pi))
                normal, best = (ZERO, ZERO, ZERO), ZERO
                for k in range(3):
                    if abs(diff[k]) > best:
                        best = abs(diff[k])
                        n = [ZERO, ZERO, ZERO]
                        n[k] = f32(sign(diff[k]) * ONE)
                        normal = tuple(n)
                nn = normalize(normal)
                return {"t": t, "normal": nn, "color": get_color_at(p, block, nn), "block": block, "pos_march": p}
    return None


# ---- intersection.glsl:78-121 -----------------------------------------------------------------------------
def intersect_sphere(o, d, mint, maxt):
    """-> (t, pos) of the unit sphere at the origin; t = INF: none"""
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        a = dot(d, d)
        b = f32(-dot(d, o))
        c = f32(dot(o, o) - ONE)
        disc = f32(f32(b * b) - f32(a * c))
        disc = sqrt(disc) if disc > 0 else INF
        t1 = f32(f32(b - disc) / a)
        t2 = f32(f32(b + disc) / a)
        t1 = t1 if (mint < t1 and t1 < maxt) else INF
        t2 = t2 if (mint < t2 and t2 < maxt) else INF
        t = gmin(t1, t2)
        pos = vadd(o, vscale(d, t))
    return t, pos


# ---- structs.glsl:54-89 -----------------------------------------------------------------------------------
LIGHTS = {
    0: [(f32(100.0), v3(1, 1, 1), v3(4, 17.5, 8.5))],
    1: [(f32(15.0), v3(1, 1, 1), v3(0, 8, 13))],
    2: [(f32(1.0), v3(1, 1, 1), v3(5, 9.3, 36.5)), (f32(1.0), v3(1, 1, 1), v3(0, 0, 0))],
}


# ---- intersection.glsl:1244-1301 --------------------------------------------------------------------------
def intersect_scene(o, d, scene, mint=ZERO, maxt=INF):
    """-> None | dict(t, pos, normal, color, type)"""
    closest = INF
    info = None
    tenth = f32(0.1)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        for (_, _, lpos) in LIGHTS[scene]:
            to = vdiv(vsub(o, lpos), tenth)
            td = vdiv(d, tenth)
            t, pos = intersect_sphere(to, td, mint, closest)
            if t < closest:
                info = {"t": t, "normal": pos, "color": (ZERO, ZERO, ZERO), "type": 2}   # Q12: unassigned Material, pinned to zero
            closest = gmin(t, closest)
        hit = grid_march(o, d, scene)
        if hit is not None and hit["t"] < closest:
            info = {"t": hit["t"], "normal": hit["normal"], "color": hit["color"], "type": 3}
            closest = hit["t"]
        if not closest < INF:
            return None
        info["normal"] = normalize(info["normal"])
        info["pos"] = vadd(vadd(o, vscale(d, info["t"])), vscale(info["normal"], f32(0.001)))
    return info


# ---- probe_pass.comp:45-71 --------------------------------------------------------------------------------
M32 = 0xFFFFFFFF


def wang_hash(seed):
    seed = ((seed ^ 61) ^ (seed >> 16)) & M32
    seed = (seed * 9) & M32
    seed = (seed ^ (seed >> 4)) & M32
    seed = (seed * 0x27D4EB2D) & M32
    return (seed ^ (seed >> 15)) & M32


class Rng:
    def __init__(self, p_idx):
        self.state = wang_hash(p_idx & M32)

    def rand(self):
        s = self.state
        s ^= (s << 13) & M32
        s ^= s >> 17
        s ^= (s << 5) & M32
        self.state = s
        return f32(f32(s) / f32(4294967296.0))   # uint -> float (round to nearest), then the division


# ---- probe_pass.comp:150-178 ------------------------------------------------------------------------------
TWO_PI = f32(6.2831853071795864769252867665590057683943)
SQRT_THIRD = f32(0.5773502691896257645091487805019574556476)


def random_dir_hemisphere(normal, rng):
    with np.errstate(invalid="ignore", divide="ignore"):
        up = sqrt(rng.rand())
        over = sqrt(f32(ONE - f32(up * up)))
        around = f32(rng.rand() * TWO_PI)
        if abs(normal[0]) < SQRT_THIRD:
            not_normal = v3(1, 0, 0)
        elif abs(normal[1]) < SQRT_THIRD:
            not_normal = v3(0, 1, 0)
        else:
            not_normal = v3(0, 0, 1)
        p1 = normalize(cross(normal, not_normal))
        p2 = normalize(cross(normal, p1))
        a = vscale(normal, up)
        b = vscale(p1, f32(cos(around) * over))
        c = vscale(p2, f32(sin(around) * over))
        return vadd(vadd(a, b), c)


# ---- probe_pass.comp:180-215 ------------------------------------------------------------------------------
def get_direct_lighting(info, scene):
    direct = (ZERO, ZERO, ZERO)
    n_visible = 0
    with np.errstate(invalid="ignore", divide="ignore", over="ignore"):
        for (intensity, col, lpos) in LIGHTS[scene]:
            to_light = normalize(vsub(lpos, info["pos"]))
            temp = intersect_scene(info["pos"], to_light, scene)
            if temp is not None:
                lambert = clamp(dot(normalize(info["normal"]), normalize(vsub(lpos, info["pos"]))), ZERO, ONE)
                if temp["type"] == 2:
                    dist = length(vsub(lpos, info["pos"]))
                    direct = vadd(direct, vdiv(vscale(vscale(col, lambert), intensity), dist))   # lambert * l.col * l.intensity / dist, left to right
                else:
                    return vscale(vscale(info["color"], f32(0.2)), lambert)
                n_visible += 1
        if n_visible != 0:
            return vdiv(vmul(info["color"], direct), f32(n_visible))
    return (ZERO, ZERO, ZERO)


# ---- probe_pass.comp:253-303 ------------------------------------------------------------------------------
def unorm8(x):
    if np.isnan(x):
        return 0
    return int(np.rint(f32(clamp(x, ZERO, ONE) * f32(255.0))))   # round to nearest-even


def probe_ray_color(index, origin, direction, scene, max_bounces):
    rng = Rng(index)
    color = (ZERO, ZERO, ZERO)
    o, d = origin, direction
    for _ in range(max_bounces):
        hit = intersect_scene(o, d, scene)
        if hit is None:
            break
        color = vadd(color, get_direct_lighting(hit, scene))
        o = vadd(hit["pos"], vscale(hit["normal"], f32(0.0001)))
        d = random_dir_hemisphere(hit["normal"], rng)
    return vdiv(color, f32(max_bounces))


def probe_update(counts, s, rays, scene, max_bounces):
    """The whole probe pass over a ray array (the engine's / oracle's structured dtype) -> albedo raster [H, W, 4] u8."""
    cx, cy, cz = counts
    width = cx * cz * s
    albedo = np.zeros((cy * s, width, 4), dtype=np.uint8)
    for index in range(len(rays)):
        r = rays[index]
        o = tuple(f32(v) for v in r["origin"])
        d = tuple(f32(v) for v in r["direction"])
        info = r["probe_info"]
        probe = to_int(f32(info[0]))
        y_probe = probe // (cx * cz)
        x_probe = probe - y_probe * (cx * cz)
        tx, ty = x_probe * s + to_int(f32(info[1])), y_probe * s + to_int(f32(info[2]))
        c = probe_ray_color(index, o, d, scene, max_bounces)
        albedo[ty, tx] = (unorm8(c[0]), unorm8(c[1]), unorm8(c[2]), 255)
    return albedo


# ---- intersection.glsl:1152-1240 --------------------------------------------------------------------------
PI = f32(3.1415926535897932384626433832795)


def tex_coord_of_probe(probe, counts, s):
    x_dim = counts[0] * counts[2]
    if probe >= x_dim * counts[1]:
        return None
    if probe < 0 or x_dim < 0:
        return None
    rx = to_int(mod(f32(probe), f32(x_dim)))           # int(mod(probe_number, x_dim)): float mod
    ry = to_int(floor(f32(probe // x_dim)))            # int(floor(probe_number / x_dim)): integer division first
    if ry >= counts[1]:
        return None
    return rx * s, ry * s


def _load(tex, x, y):
    px = tex[y, x]
    return tuple(f32(f32(int(px[k])) / f32(255.0)) for k in range(3))


def sample_probe(probe, direction, which, albedo, distance, counts, s):
    top = tex_coord_of_probe(probe, counts, s)
    if top is None:
        return v3(1, 0, 1)
    with np.errstate(invalid="ignore", divide="ignore"):
        d = normalize(direction)
        rx = to_int(f32(f32(f32(f32(-1.0) * f32(d[2] - ONE)) / f32(2.0)) * f32(s)))
        if rx == s:
            rx = 0
        sqrt_z = sqrt(f32(ONE - f32(d[2] * d[2])))
        ry = to_int(f32(f32(acos(f32(d[0] / sqrt_z)) / f32(f32(2.0) * PI)) * f32(s)))
    sx, sy = top[0] + rx, top[1] + ry
    result = _load(albedo, sx, sy)     # (the centre texel of the ALBEDO image whichever texture is asked for, Q9)
    count = 0
    for dx in range(-2, 3):
        x = sx + dx
        if x < top[0] or x >= top[0] + s:
            continue
        for dy in range(-2, 3):
            y = sy + dy
            if y < top[1] or y >= top[1] + s:
                continue
            count += 1
            result = vadd(result, _load(albedo if which == 0 else distance, x, y))
    with np.errstate(invalid="ignore", divide="ignore"):
        return vdiv(result, f32(count))


# ---- intersection.glsl:1306-1409 --------------------------------------------------------------------------
def get_diffuse_gi(pos, normal, albedo, distance, counts, side, origin, s):
    """-> (rgb, the 8 probe indices of the cage; all -1 when the shader returns magenta)"""
    cage = [-1] * 8
    magenta = [-1] * 8
    with np.errstate(invalid="ignore", divide="ignore", over="ignore"):
        n = normalize(normal)
        fside = f32(side)
        base = tuple(to_int(floor(f32(f32(p - f32(o)) / fside))) for p, o in zip(pos, origin))
        half_x = f32(np.floor(f32(f32(counts[0]) / f32(2.0))))   # int(vec3) takes .x: the x count bounds all three axes (Q6)
        for k in range(3):
            if base[k] < to_int(f32(-half_x)) or base[k] > to_int(f32(half_x - ONE)):
                return v3(1, 0, 1), magenta
        base_world = tuple(f32(f32(f32(b) * fside) + f32(o)) for b, o in zip(base, origin))   # (ivec3 * int) converted, + origin
        irradiance = (ZERO, ZERO, ZERO)
        sum_weight = ZERO
        alpha = tuple(clamp(f32(f32(p - bw) / fside), ZERO, ONE) for p, bw in zip(pos, base_world))
        for i in range(8):
            offset = ((i >> 2) & 1, (i >> 1) & 1, i & 1)
            cur = tuple(b + o for b, o in zip(base, offset))
            shifted = tuple(c + (cnt // 2) for c, cnt in zip(cur, counts))   # ivec3(floor(probe_counts / 2)): an ivec3 divided by an int
            idx = shifted[1] * counts[0] * counts[2] + shifted[2] * counts[0] + shifted[0]
            if idx < 0 or idx >= counts[0] * counts[1] * counts[2]:
                return v3(1, 0, 1), magenta
            cage[i] = idx
            tri = tuple(mix(f32(ONE - a), a, f32(o)) for a, o in zip(alpha, offset))
            probe_pos = tuple(f32(bw + f32(f32(o) * fside)) for bw, o in zip(base_world, offset))
            direction = normalize(vsub(probe_pos, pos))
            temp = gmax(f32(0.0001), f32(f32(dot(direction, n) + ONE) * HALF))
            weight = f32(f32(temp * temp) + f32(0.2))
            # (the Chebyshev term is computed and not applied, :1363-1383; its sample_probe call has no effect on the result)
            weight = gmax(f32(0.000001), weight)
            if weight < f32(0.2):
                weight = f32(weight * f32(f32(weight * weight) * f32(ONE / f32(f32(0.2) * f32(0.2)))))
            weight = f32(weight * f32(f32(tri[0] * tri[1]) * tri[2]))
            sample = sample_probe(idx, n, 0, albedo, distance, counts, s)
            irradiance = vadd(irradiance, vscale(sample, weight))
            sum_weight = f32(sum_weight + weight)
        return vdiv(irradiance, sum_weight), cage


# ---- the dormant pieces (DDGI mode) ------------------------------------------------------------------------
def update_lights(scene, time, lights):
    """probe_pass.comp:217-251 on one scene's table [(intensity, col, pos)] -> the moved positions.  GLSL globals are
    initialised per invocation: the offsets apply to the table as shipped / as given, once.  Literals without a suffix are
    floats (GLSL has no implicit double), `(i + 1) * 2` and `(i / 2) * 4` are integer expressions converted when they meet
    the float, and a * b * c groups from the left."""
    time = f32(time)
    out = []
    for i, (_, _, pos) in enumerate(lights):
        x, y, z = pos
        if scene == 0:
            t = f32(f32(0.05) * time)
            if i == 0:
                z = f32(z + f32(f32(10) * cos(f32(t * f32(0.1)))))
            else:
                x = f32(x + f32(f32((i + 1) * 2) * sin(f32(t * f32(0.5)))))
                y = f32(y + f32(f32((i // 2) * 4) * sin(f32(t * f32(0.5)))))
                z = f32(z + f32(f32((i + 1) * 2) * cos(f32(t * f32(0.5)))))
        elif scene == 1:
            t = f32(f32(0.005) * time)
            x = f32(x + f32(f32(i + 1) * sin(t)))
            y = f32(y + f32(f32((i // 2) * 4) * sin(t)))
            z = f32(z + f32(f32(i + 1) * cos(t)))
        elif scene == 2:
            d = f32(f32(0.00005) * time)
            x, y, z = f32(x + d), f32(y + d), f32(z + d)
        out.append((x, y, z))
    return out


def sign_not_zero(x):
    return ONE if x >= ZERO else f32(-1.0)


def oct_encode(v):
    """octahedral.glsl:16-23"""
    l1norm = f32(f32(abs(v[0]) + abs(v[1])) + abs(v[2]))
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        inv = f32(ONE / l1norm)
        rx, ry = f32(v[0] * inv), f32(v[1] * inv)
    if v[2] < ZERO:
        rx, ry = f32(f32(ONE - abs(ry)) * sign_not_zero(rx)), f32(f32(ONE - abs(rx)) * sign_not_zero(ry))
    return rx, ry


def oct_decode(o):
    """octahedral.glsl:28-34"""
    x, y = f32(o[0]), f32(o[1])
    z = f32(f32(ONE - abs(x)) - abs(y))
    if z < ZERO:
        x, y = f32(f32(ONE - abs(y)) * sign_not_zero(x)), f32(f32(ONE - abs(x)) * sign_not_zero(y))
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        return normalize(v3(x, y, z))


def hysteresis_mix(old, new, hysteresis):
    """probe_pass.comp:298-299: color = mix(old, new, hysteresis) — GLSL's mix puts the weight `hysteresis` on the NEW value"""
    return mix(f32(old), f32(new), f32(hysteresis))
