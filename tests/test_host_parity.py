"""CPU-side parity of the product's HOST code against the oracle (no GPU needed): pinned
elementary functions, scene bake, probe-ray generation."""
import ctypes as C

import numpy as np
import pytest

from tests.common import CONFIGS


def _bits(a):
    return np.asarray(a, dtype=np.float32).view(np.uint32)


def _sample_args(rng):
    xs = [rng.uniform(-10, 10, 20000), rng.uniform(-2e6, 2e6, 20000), rng.uniform(-3e8, 3e8, 20000),
          np.array([0.0, -0.0, 1e-30, 3.14159265, 6.2831853, 1.5707964, 17000.123, 2.2e8, -2.2e8])]
    return np.concatenate(xs).astype(np.float32)


def test_pinned_sincos_product_equals_oracle_bitwise_and_tracks_libm(ddgi, oracle):
    lib = ddgi.load_library()
    rng = np.random.default_rng(7)
    xs = _sample_args(rng)
    ps = np.array([lib.ddgi_pinned_sinf(float(x)) for x in xs], dtype=np.float32)
    pc = np.array([lib.ddgi_pinned_cosf(float(x)) for x in xs], dtype=np.float32)
    os_ = np.array([oracle.lib().oracle_sinf(float(x)) for x in xs], dtype=np.float32)
    oc = np.array([oracle.lib().oracle_cosf(float(x)) for x in xs], dtype=np.float32)
    assert np.array_equal(_bits(ps), _bits(os_))
    assert np.array_equal(_bits(pc), _bits(oc))
    # closeness to the correctly rounded value (computed in float64): within 1 ulp of float32
    ref_s = np.sin(xs.astype(np.float64))
    ref_c = np.cos(xs.astype(np.float64))
    ulp = np.spacing(np.maximum(np.abs(ref_s), 1e-30).astype(np.float32)).astype(np.float64)
    assert np.max(np.abs(ps - ref_s) / ulp) <= 1.0
    ulp = np.spacing(np.maximum(np.abs(ref_c), 1e-30).astype(np.float32)).astype(np.float64)
    assert np.max(np.abs(pc - ref_c) / ulp) <= 1.0
    # it is the correctly rounded value almost always, and so is glibc's sinf: the two agree
    # bit-for-bit in the overwhelming majority of cases (LITERAL vs PINNED oracle arithmetic)
    assert np.mean(_bits(ps) == _bits(ref_s.astype(np.float32))) > 0.9999
    libm = C.CDLL("libm.so.6")
    libm.sinf.restype = C.c_float
    libm.sinf.argtypes = [C.c_float]
    libm_s = np.array([libm.sinf(float(x)) for x in xs], dtype=np.float32)
    assert np.mean(_bits(ps) == _bits(libm_s)) > 0.98  # glibc sinf itself is not correctly rounded
    assert np.isnan(lib.ddgi_pinned_sinf(float("inf"))) and np.isnan(lib.ddgi_pinned_cosf(float("nan")))


def test_pinned_small_angle_sincos_product_equals_oracle_bitwise(ddgi, oracle):
    """P6b: the binary32 sine/cosine of the hemisphere sample's angle (around = rand * 2 pi): product and oracle copies
    agree bit for bit, and stay within 2e-7 of the true value — the Vulkan spec allows a driver 2^-11."""
    lib = ddgi.load_library()
    olib = oracle.lib()
    olib.oracle_sincos_small.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    rng = np.random.default_rng(9)
    two_pi = np.float32(6.2831853071795864769)
    xs = np.concatenate([(rng.random(60000, dtype=np.float32) * two_pi), rng.uniform(-40, 40, 5000).astype(np.float32),
                         np.float32([0.0, -0.0, 1e-30, 0.78539816, 0.7853982, 1.5707964, 3.1415927, 4.712389, 6.2831850, 6.2831855])]).astype(np.float32)
    got = np.zeros((len(xs), 2), dtype=np.float32)
    want = np.zeros((len(xs), 2), dtype=np.float32)
    s, c = C.c_float(), C.c_float()
    for i, x in enumerate(xs):
        lib.ddgi_pinned_sincos_small(float(x), C.byref(s), C.byref(c))
        got[i] = (s.value, c.value)
        olib.oracle_sincos_small(float(x), C.byref(s), C.byref(c))
        want[i] = (s.value, c.value)
    assert np.array_equal(_bits(got), _bits(want))
    x64 = xs.astype(np.float64)
    assert np.max(np.abs(got[:, 0] - np.sin(x64))) < 2e-7 and np.max(np.abs(got[:, 1] - np.cos(x64))) < 2e-7


def test_pinned_acos_product_equals_oracle_bitwise_and_tracks_libm(ddgi, oracle):
    lib = ddgi.load_library()
    rng = np.random.default_rng(8)
    xs = np.concatenate([rng.uniform(-1, 1, 50000), [-1, 1, 0, 0.5, -0.5, 0.49999997, 0.99999994]]).astype(np.float32)
    pa = np.array([lib.ddgi_pinned_acosf(float(x)) for x in xs], dtype=np.float32)
    oa = np.array([oracle.lib().oracle_acosf(float(x)) for x in xs], dtype=np.float32)
    assert np.array_equal(_bits(pa), _bits(oa))
    ref = np.arccos(xs.astype(np.float64))
    ulp = np.spacing(np.maximum(ref, 1e-30).astype(np.float32)).astype(np.float64)
    assert np.max(np.abs(pa - ref) / ulp) <= 1.0
    assert np.isnan(lib.ddgi_pinned_acosf(1.0000001)) and np.isnan(lib.ddgi_pinned_acosf(float("nan")))


@pytest.mark.parametrize("scene,box", [
    (1, ((-16, 16), (-16, 16), (-2, 32))),
    (2, ((-32, 32), (-10, 10), (-22, 22))),
    (0, ((-56, 46), (-30, 24), (-50, 44))),
])
def test_scene_bake_equals_procedural_getBlockAt(ddgi, oracle, scene, box):
    """The clamped lookup into the bake (what the kernels traverse) against the oracle's
    procedural getBlockAt on a box well beyond the bake."""
    g = oracle.lib().oracle_get_block_at
    (x0, x1), (y0, y1), (z0, z1) = box
    shape = (x1 - x0 + 1, y1 - y0 + 1, z1 - z0 + 1)
    want = np.zeros(shape, dtype=np.int32)
    got = np.zeros(shape, dtype=np.int32)
    for ix, x in enumerate(range(x0, x1 + 1)):
        for iy, y in enumerate(range(y0, y1 + 1)):
            for iz, z in enumerate(range(z0, z1 + 1)):
                want[ix, iy, iz] = g(x, y, z, scene)
                got[ix, iy, iz] = ddgi.scene_block_at(scene, x, y, z)
    assert np.array_equal(want > 0, got > 0), "occupancy differs"
    # block types must agree EVERYWHERE, not only on voxels a ray can arrive at from empty space: a
    # probe placed inside solid rock starts its rays inside a voxel and is shaded with that type
    assert np.array_equal(want, got), f"{(want != got).sum()} voxels have a different block type"


@pytest.mark.parametrize("name", ["c1_cornell", "cave_small", "cave_odd", "c2_cornell"])
def test_host_probe_rays_equal_oracle_bitwise(ddgi, oracle, name):
    counts, side, s, origin, _ = CONFIGS[name]
    mine = ddgi.generate_probe_rays_host(ddgi.make_field(counts, side, s, origin), seed=1)
    ref = oracle.generate_probe_rays(oracle.make_field(counts, side, s, origin), oracle.new_rand_state(1))
    assert mine.tobytes() == ref.tobytes()


def test_repeated_generate_calls_continue_the_rand_sequence(ddgi, oracle):
    # Q1: each later generate_probe_rays() call continues the same rand() sequence
    counts, side, s, origin, _ = CONFIGS["c1_cornell"]
    st = oracle.new_rand_state(1)
    f = oracle.make_field(counts, side, s, origin)
    first = oracle.generate_probe_rays(f, st)
    second = oracle.generate_probe_rays(f, st)
    assert first.tobytes() != second.tobytes()
    mine2 = ddgi.generate_probe_rays_host(ddgi.make_field(counts, side, s, origin), seed=1, skip_calls=1)
    assert mine2.tobytes() == second.tobytes()


@pytest.mark.parametrize("scene", [0, 1, 2])
def test_skip_field_is_a_lower_bound_of_the_free_radius(ddgi, scene):
    """The fast march (tolerance mode) skips `code - 1` voxels beyond grid_march's own step on the strength of the scene's skip
    field: code 0 = occupied, else every voxel within Chebyshev distance code - 1 — in the world the kernels see, i.e. with clamped
    lookups outside the bake box — must be empty, and the code must be the largest (up to 3) for which that holds.  Brute force."""
    from scipy import ndimage

    lo, codes = ddgi.scene_skip_field(scene)
    nz, ny, nx = codes.shape
    occ = np.zeros(codes.shape, dtype=bool)
    for iz in range(nz):   # the bake, voxel by voxel, through the same clamped lookup the kernels' world is defined by
        for iy in range(ny):
            for ix in range(0, nx, max(1, nx // 8)):   # (a spread of columns suffices to tie `codes == 0` to the bake ...)
                assert (ddgi.scene_block_at(scene, lo[0] + ix, lo[1] + iy, lo[2] + iz) > 0) == (codes[iz, iy, ix] == 0)
    occ = codes == 0                                   # (... the rest of the check uses the field's own occupancy)
    pad = np.pad(occ, 3, mode="edge")                  # outside the box the border layer repeats
    dist = ndimage.distance_transform_cdt(~pad, metric="chessboard")[3:-3, 3:-3, 3:-3]   # Chebyshev distance to the nearest occupied voxel
    want = np.where(occ, 0, np.minimum(dist, 3)).astype(np.uint8)   # free radius r = dist - 1; code = 1 + min(r, 2) = min(dist, 3)
    assert np.array_equal(codes, want)
    assert (codes == 3).any() and (codes == 1).any()
