"""ctypes binding of the CPU oracle (oracle/libddgi_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; never by the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libddgi_oracle.so")


class Field(C.Structure):
    _fields_ = [
        ("probe_count", C.c_int32 * 3),
        ("side_length", C.c_int32),
        ("hysteresis", C.c_float),
        ("sqrt_rays_per_probe", C.c_int32),
        ("_pad0", C.c_int32 * 2),
        ("field_origin", C.c_float * 3),
        ("visualize", C.c_uint8),
        ("_pad1", C.c_uint8 * 3),
    ]


class Settings(C.Structure):
    _fields_ = [
        ("screen_width", C.c_int32),
        ("screen_height", C.c_int32),
        ("max_bounces", C.c_int32),
        ("camera_mode", C.c_int32),
        ("render_mode", C.c_int32),
        ("scene", C.c_int32),
        ("time", C.c_float),
        ("visualize_probes", C.c_int32),
    ]


class RandState(C.Structure):
    _fields_ = [("ring", C.c_uint32 * 31), ("k", C.c_uint32), ("seeded", C.c_int32)]


RAY_DTYPE = np.dtype(
    [
        ("origin", "<f4", 3),
        ("_p0", "<f4"),
        ("direction", "<f4", 3),
        ("_p1", "<f4"),
        ("probe_info", "<f4", 3),
        ("_p2", "<f4"),
    ]
)
LIGHT_DTYPE = np.dtype([("intensity", "<f4"), ("col", "<f4", 3), ("pos", "<f4", 3)])
assert RAY_DTYPE.itemsize == 48 and LIGHT_DTYPE.itemsize == 28
assert C.sizeof(Field) == 48 and C.sizeof(Settings) == 32


def build(force=False):
    src = [os.path.join(_HERE, n) for n in ("ddgi_oracle.c", "pinned_math.h", "Makefile")]
    stale = (not os.path.exists(_SO)) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in src)
    if force or stale:
        subprocess.run(["make", "-C", _HERE, "-B" if force else "-s", "libddgi_oracle.so"],
                       check=True, stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.oracle_wang_hash.restype = C.c_uint32
        L.oracle_wang_hash.argtypes = [C.c_uint32]
        L.oracle_glibc_rand.restype = C.c_int32
        L.oracle_get_block_at.argtypes = [C.c_float, C.c_float, C.c_float, C.c_int]
        L.oracle_probe_update.argtypes = [
            C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_int,
            C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.oracle_sample.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_uint64, C.c_void_p, C.c_void_p]
        for n in ("oracle_sinf", "oracle_cosf", "oracle_acosf"):
            getattr(L, n).restype = C.c_float
            getattr(L, n).argtypes = [C.c_float]
        _lib = L
    return _lib


def make_field(counts, side, s, origin, hysteresis=0.9):
    f = Field()
    f.probe_count[:] = list(counts)
    f.side_length = int(side)
    f.hysteresis = hysteresis
    f.sqrt_rays_per_probe = int(s)
    f.field_origin[:] = list(origin)
    f.visualize = 1
    return f


def make_settings(scene, max_bounces=8, time=0.0):
    st = Settings()
    st.screen_width, st.screen_height = 1600, 900
    st.max_bounces = max_bounces
    st.scene = scene
    st.time = time
    return st


def set_arith(pinned):
    lib().oracle_set_arith(1 if pinned else 0)


def new_rand_state(seed=1):
    st = RandState()
    lib().oracle_glibc_srand(C.byref(st), C.c_uint32(seed))
    return st


def glibc_rand(st):
    return lib().oracle_glibc_rand(C.byref(st))


_tile = [0, 0]


def set_ray_tile(tile_x=0, tile_y=0):
    """Non-square ray tile (tile_x strata along z x tile_y strata along phi); (0, 0) = square, from the field."""
    _tile[0], _tile[1] = int(tile_x), int(tile_y)
    lib().oracle_set_ray_tile(int(tile_x), int(tile_y))


def ray_tile(field):
    s = field.sqrt_rays_per_probe
    return (_tile[0] or s, _tile[1] or s)


def rays_per_probe(field):
    tx, ty = ray_tile(field)
    return tx * ty


def generate_probe_rays(field, rand_state):
    P = field.probe_count[0] * field.probe_count[1] * field.probe_count[2]
    n = P * rays_per_probe(field)
    rays = np.zeros(n, dtype=RAY_DTYPE)
    lib().oracle_generate_probe_rays(C.byref(field), C.byref(rand_state), rays.ctypes.data_as(C.c_void_p))
    return rays


def texture_size(field):
    tx, ty = ray_tile(field)
    return field.probe_count[0] * field.probe_count[2] * tx, field.probe_count[1] * ty


def probe_update(field, settings, rays, first=0, count=None, lights=None, nthreads=0, want_float=False):
    """REF-mode probe update -> (albedo[H,W,4] u8, distance[H,W,4] u8[, colors f32])."""
    W, H = texture_size(field)
    n = len(rays) if count is None else count
    albedo = np.zeros((H, W, 4), dtype=np.uint8)
    dist = np.zeros((H, W, 4), dtype=np.uint8)
    colors = np.zeros((n, 3), dtype=np.float32) if want_float else None
    lp, nl = None, 0
    if lights is not None:
        lights = np.ascontiguousarray(lights, dtype=LIGHT_DTYPE)
        lp, nl = lights.ctypes.data_as(C.c_void_p), len(lights)
    lib().oracle_probe_update(
        C.byref(field), C.byref(settings), rays.ctypes.data_as(C.c_void_p), first, n, lp, nl,
        albedo.ctypes.data_as(C.c_void_p), dist.ctypes.data_as(C.c_void_p),
        colors.ctypes.data_as(C.c_void_p) if want_float else None, nthreads)
    return (albedo, dist, colors) if want_float else (albedo, dist)


def probe_update_probes(field, settings, rays, probes, nthreads=0, albedo=None):
    """REF-mode update of the listed probes only (one parallel loop) -> albedo[H,W,4] u8."""
    W, H = texture_size(field)
    if albedo is None:
        albedo = np.zeros((H, W, 4), dtype=np.uint8)
    probes = np.ascontiguousarray(probes, dtype=np.int32)
    lib().oracle_probe_update_probes(
        C.byref(field), C.byref(settings), rays.ctypes.data_as(C.c_void_p),
        probes.ctypes.data_as(C.c_void_p), len(probes), albedo.ctypes.data_as(C.c_void_p), nthreads)
    return albedo


def sample(field, albedo, distance, pos, nrm):
    pos = np.ascontiguousarray(pos, dtype=np.float32)
    nrm = np.ascontiguousarray(nrm, dtype=np.float32)
    n = pos.shape[0]
    rgb = np.zeros((n, 3), dtype=np.float32)
    cage = np.zeros((n, 8), dtype=np.int32)
    albedo = np.ascontiguousarray(albedo)
    distance = np.ascontiguousarray(distance)
    lib().oracle_sample(C.byref(field), albedo.ctypes.data_as(C.c_void_p),
                        distance.ctypes.data_as(C.c_void_p), pos.ctypes.data_as(C.c_void_p),
                        nrm.ctypes.data_as(C.c_void_p), n, rgb.ctypes.data_as(C.c_void_p),
                        cage.ctypes.data_as(C.c_void_p))
    return rgb, cage


def grid_march(o, d, scene):
    o = np.asarray(o, dtype=np.float32)
    d = np.asarray(d, dtype=np.float32)
    out = np.zeros(10, dtype=np.float32)
    it = C.c_int(0)
    block = lib().oracle_grid_march(o.ctypes.data_as(C.c_void_p), d.ctypes.data_as(C.c_void_p),
                                    scene, out.ctypes.data_as(C.c_void_p), C.byref(it))
    return block, it.value, out


def rng_kat(p_idx):
    u = (C.c_uint32 * 3)()
    f = (C.c_float * 2)()
    lib().oracle_rng_kat(C.c_uint32(p_idx), u, f)
    return list(u), list(f)


def num_threads():
    return lib().oracle_num_threads()


# ---- DDGI mode ---------------------------------------------------------------------------------

IRR_TILE, DEP_TILE = 8, 16


def shipped_lights(scene):
    out = np.zeros(8, dtype=LIGHT_DTYPE)
    n = C.c_int(0)
    lib().oracle_shipped_lights(scene, out.ctypes.data_as(C.c_void_p), C.byref(n))
    return out[: n.value].copy()


def update_lights(scene, time, base):
    base = np.ascontiguousarray(base, dtype=LIGHT_DTYPE)
    out = np.zeros(len(base), dtype=LIGHT_DTYPE)
    lib().oracle_update_lights(scene, C.c_float(time), base.ctypes.data_as(C.c_void_p), len(base),
                               out.ctypes.data_as(C.c_void_p))
    return out


def oct_encode(v):
    """a20 (octahedral.glsl:16-23) on an (n, 3) array of vectors -> (n, 2)"""
    v = np.ascontiguousarray(v, dtype=np.float32).reshape(-1, 3)
    out = np.zeros((len(v), 2), dtype=np.float32)
    lib().oracle_oct_encode(v.ctypes.data_as(C.c_void_p), len(v), out.ctypes.data_as(C.c_void_p))
    return out


def oct_decode(uv):
    """a20 (octahedral.glsl:28-34) on an (n, 2) array -> (n, 3)"""
    uv = np.ascontiguousarray(uv, dtype=np.float32).reshape(-1, 2)
    out = np.zeros((len(uv), 3), dtype=np.float32)
    lib().oracle_oct_decode(uv.ctypes.data_as(C.c_void_p), len(uv), out.ctypes.data_as(C.c_void_p))
    return out


def frame_rotation(frame):
    m = np.zeros(9, dtype=np.float32)
    lib().oracle_frame_rotation(C.c_uint32(frame), m.ctypes.data_as(C.c_void_p))
    return m.reshape(3, 3)


def new_tiles(field):
    P = field.probe_count[0] * field.probe_count[1] * field.probe_count[2]
    return (np.zeros((P, IRR_TILE, IRR_TILE, 4), dtype=np.float32), np.zeros((P, DEP_TILE, DEP_TILE, 2), dtype=np.float32))


def ddgi_update(field, settings, frame, irradiance, depth, lights=None, probes=None, nthreads=0, want_radiance=False):
    """One DDGI-mode probe update in place on (irradiance, depth) -> optional radiance [P*n, 4]."""
    P = field.probe_count[0] * field.probe_count[1] * field.probe_count[2]
    n = rays_per_probe(field)
    rad = np.zeros((P * n, 4), dtype=np.float32) if want_radiance else None
    lp, nl = None, 0
    if lights is not None:
        lights = np.ascontiguousarray(lights, dtype=LIGHT_DTYPE)
        lp, nl = lights.ctypes.data_as(C.c_void_p), len(lights)
    first, count = (0, P) if probes is None else probes
    lib().oracle_ddgi_update(C.byref(field), C.byref(settings), lp, nl, C.c_uint32(frame),
                             irradiance.ctypes.data_as(C.c_void_p), depth.ctypes.data_as(C.c_void_p),
                             rad.ctypes.data_as(C.c_void_p) if want_radiance else None, first, count, nthreads)
    return rad


def ddgi_sample(field, irradiance, depth, pos, nrm):
    pos = np.ascontiguousarray(pos, dtype=np.float32)
    nrm = np.ascontiguousarray(nrm, dtype=np.float32)
    n = pos.shape[0]
    rgb = np.zeros((n, 3), dtype=np.float32)
    cage = np.zeros((n, 8), dtype=np.int32)
    lib().oracle_ddgi_sample(C.byref(field), irradiance.ctypes.data_as(C.c_void_p), depth.ctypes.data_as(C.c_void_p),
                             pos.ctypes.data_as(C.c_void_p), nrm.ctypes.data_as(C.c_void_p), C.c_uint64(n),
                             rgb.ctypes.data_as(C.c_void_p), cage.ctypes.data_as(C.c_void_p))
    return rgb, cage


# ---- camera rays + integrators -----------------------------------------

class OCamera(C.Structure):
    _fields_ = [("matrix", C.c_float * 16), ("params", C.c_float * 4)]


def render(field, settings, camera, tex0, tex1, ddgi_mode=False, lights=None, want_float=False):
    """compute_pass.comp:main -> rgba8 [H, W, 4] (+ rgb f32).  tex0/tex1: REF rasters (u8) or DDGI tiles (f32).
    `camera` is any ctypes struct with the 80-byte Camera UBO layout."""
    w, h = settings.screen_width, settings.screen_height
    img = np.zeros((h, w, 4), dtype=np.uint8)
    rgb = np.zeros((h, w, 3), dtype=np.float32) if want_float else None
    lp, nl = None, 0
    if lights is not None:
        lights = np.ascontiguousarray(lights, dtype=LIGHT_DTYPE)
        lp, nl = lights.ctypes.data_as(C.c_void_p), len(lights)
    tex0 = np.ascontiguousarray(tex0)
    tex1 = np.ascontiguousarray(tex1)
    lib().oracle_render(C.byref(field), C.byref(settings), C.byref(camera), lp, nl, 1 if ddgi_mode else 0,
                        tex0.ctypes.data_as(C.c_void_p), tex1.ctypes.data_as(C.c_void_p), w, h,
                        img.ctypes.data_as(C.c_void_p), rgb.ctypes.data_as(C.c_void_p) if want_float else None)
    return (img, rgb) if want_float else img
