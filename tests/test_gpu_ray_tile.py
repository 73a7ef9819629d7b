"""Non-square ray counts on the GPU (ddgi_set_ray_tile) and BASELINE config C4:
64x32x64 probes x 512 rays per probe (a 32 x 16 ray tile), on one GPU.

REF mode: texels, sampled rgb and cage indices bit-exact against the oracle (PINNED arithmetic) with the
same tile; DDGI mode: n = tile_x * tile_y Fibonacci rays, float tiles bit-exact."""
import numpy as np
import pytest

from tests.common import CONFIGS, shading_points

pytestmark = pytest.mark.gpu

C4 = {"counts": (64, 32, 64), "side": 1, "s": 16, "origin": (1.4, 0.0, 1.0), "scene": 0, "tile": (32, 16)}


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.mark.parametrize("name,tile", [("cave_small", (8, 4)), ("c1_cornell", (32, 16)), ("house_small", (3, 11)), ("cave_odd", (1, 7))])
def test_ref_mode_tile_bit_exact(ddgi, oracle, name, tile):
    counts, side, s, origin, scene = CONFIGS[name]
    pos, nrm = shading_points(np.random.default_rng(23), counts, side, origin, 2048)
    nrm[:4] = [[0, 0, 1], [0, 0, -1], [1, 0, 0], [0, 1, 0]]
    with ddgi.ProbeEngine(ddgi.make_field(counts, side, s, origin), ddgi.make_settings(scene, 8)) as eng:
        eng.set_ray_tile(*tile)
        assert eng.ray_tile == tile and eng.rays_per_probe == tile[0] * tile[1]
        with pytest.raises(ddgi.DDGIError):
            eng.probe_update()                 # the rays of the old tile are gone
        eng.generate_probe_rays(seed=1)
        rays = eng.get_probe_rays()
        eng.probe_update()
        albedo, distance = eng.read_textures()
        rgb, cage = eng.sample(pos, nrm)
        eng.set_ray_tile(0, 0)                 # back to the field's square tile
        assert eng.ray_tile == (s, s)
    oracle.set_ray_tile(*tile)
    f = oracle.make_field(counts, side, s, origin)
    want_rays = oracle.generate_probe_rays(f, oracle.new_rand_state(1))
    assert rays.tobytes() == want_rays.tobytes()
    want_a, want_d = oracle.probe_update(f, oracle.make_settings(scene, 8), want_rays)
    assert albedo.shape == want_a.shape == (counts[1] * tile[1], counts[0] * counts[2] * tile[0], 4)
    assert np.array_equal(albedo, want_a) and not distance.any()
    want_rgb, want_cage = oracle.sample(f, albedo, distance, pos, nrm)
    assert np.array_equal(cage, want_cage)
    assert np.array_equal(_bits(rgb), _bits(want_rgb))
    assert (cage[:, 0] >= 0).any()


@pytest.mark.parametrize("name,tile", [("c1_cornell", (25, 2)), ("cave_small", (32, 16)), ("cave_odd", (7, 1))])
def test_ddgi_mode_any_ray_count_bit_exact(ddgi, oracle, name, tile):
    counts, side, s, origin, scene = CONFIGS[name]
    pos, nrm = shading_points(np.random.default_rng(29), counts, side, origin, 1024)
    f = oracle.make_field(counts, side, s, origin)
    irr, dep = oracle.new_tiles(f)
    with ddgi.ProbeEngine(ddgi.make_field(counts, side, s, origin), ddgi.make_settings(scene, 8)) as eng:
        eng.set_mode(ddgi.MODE_DDGI)
        eng.set_ray_tile(*tile)
        oracle.set_ray_tile(*tile)
        for frame in range(2):
            eng.probe_update(ddgi.make_settings(scene, 8, time=2.0 * (frame + 1)))
            oracle.ddgi_update(f, oracle.make_settings(scene, 8, time=2.0 * (frame + 1)), frame, irr, dep)
            g_irr, g_dep = eng.read_tiles()
            assert np.array_equal(_bits(g_irr), _bits(irr)), f"irradiance tiles differ at frame {frame}"
            assert np.array_equal(_bits(g_dep), _bits(dep)), f"depth tiles differ at frame {frame}"
        rgb, cage = eng.sample(pos, nrm)
    want_rgb, want_cage = oracle.ddgi_sample(f, irr, dep, pos, nrm)
    assert np.array_equal(cage, want_cage) and np.array_equal(_bits(rgb), _bits(want_rgb))


def test_c4_full_size_ref_mode(ddgi, oracle):
    """BASELINE config C4 on one GPU: 131 072 probes x 512 rays = 67 108 864 probe rays (3.2 GB of
    ProbeRay records resident in HBM).  Size-independent properties + 96 oracle-checked probes."""
    c = C4
    tx, ty = c["tile"]
    with ddgi.ProbeEngine(ddgi.make_field(c["counts"], c["side"], c["s"], c["origin"]), ddgi.make_settings(c["scene"], 8)) as eng:
        eng.set_ray_tile(tx, ty)
        assert eng.num_rays == 67108864
        eng.generate_probe_rays(seed=1)
        eng.probe_update()
        a1, d1 = eng.read_textures()
        ms = eng.last_update_ms()
        eng.probe_update()
        a2, _ = eng.read_textures()
        rays = eng.get_probe_rays()
        # the cage sampler on a grid whose albedo texture (268 MB) is far beyond the caches: a frame's worth of points takes sample_probe
        # from the per-texel table (1.07 GB, 2x2x2 bricks of probes, a 32 x 16 tile) — equal to the per-point path bit for bit
        rng = np.random.default_rng(5)
        n_pts = 1_200_000
        half = np.asarray(c["counts"], dtype=np.float64) * c["side"] * 0.49
        pos = (rng.uniform(-1, 1, size=(n_pts, 3)) * half + np.asarray(c["origin"])).astype(np.float32)
        nrm = rng.normal(size=(n_pts, 3)).astype(np.float32)
        rgb_table, cage_table = eng.sample(pos, nrm)
        eng.set_tuning("sample_box", 0)
        rgb_direct, cage_direct = eng.sample(pos, nrm)
        eng.set_tuning("sample_box", 1)
    assert np.array_equal(cage_table, cage_direct) and np.array_equal(rgb_table.view(np.uint32), rgb_direct.view(np.uint32))
    assert (cage_table[:, 0] >= 0).mean() > 0.9
    assert a1.shape == (32 * ty, 64 * 64 * tx, 4)
    assert np.array_equal(a1, a2)                       # Q18: every frame writes the same texels
    assert not d1.any() and (a1[..., 3] == 255).all()
    assert ms["trace_ms"] > 0
    tiles = a1[..., :3].reshape(32, ty, 64 * 64, tx, 3).any(axis=(1, 3, 4))
    assert tiles.mean() > 0.9
    oracle.set_ray_tile(tx, ty)
    f = oracle.make_field(c["counts"], c["side"], c["s"], c["origin"])
    # the oracle's own rays for a sample of probes equal the engine's (a full second copy would be 3.2 GB)
    small = oracle.make_field((4, 2, 4), c["side"], c["s"], c["origin"])
    dirs = oracle.generate_probe_rays(small, oracle.new_rand_state(1))["direction"][: tx * ty]
    assert np.array_equal(rays["direction"][: tx * ty], dirs) and np.array_equal(rays["direction"][-tx * ty:], dirs)
    probes = np.random.default_rng(7).choice(64 * 32 * 64, size=96, replace=False)
    want = oracle.probe_update_probes(f, oracle.make_settings(c["scene"], 8), rays.view(oracle.RAY_DTYPE), probes)
    cxz = 64 * 64
    for p in probes:
        x0, y0 = (int(p) % cxz) * tx, (int(p) // cxz) * ty
        assert np.array_equal(a1[y0:y0 + ty, x0:x0 + tx], want[y0:y0 + ty, x0:x0 + tx]), f"probe {p}"
    # ... and a slice of the sampled batch against the oracle, on the textures read back
    want_rgb, want_cage = oracle.sample(f, a1, d1, pos[:20000], nrm[:20000])
    assert np.array_equal(cage_table[:20000], want_cage) and np.array_equal(rgb_table[:20000].view(np.uint32), want_rgb.view(np.uint32))


def test_c4_full_size_ddgi_mode(ddgi, oracle):
    """C4 in DDGI mode (512 Fibonacci rays per probe, octahedral blend): two frames, the first 24 and the
    last 24 probes (reference order) against the oracle, bit for bit, plus properties of all tiles."""
    c = C4
    f = oracle.make_field(c["counts"], c["side"], c["s"], c["origin"])
    P = 64 * 32 * 64
    with ddgi.ProbeEngine(ddgi.make_field(c["counts"], c["side"], c["s"], c["origin"]), ddgi.make_settings(c["scene"], 8)) as eng:
        eng.set_mode(ddgi.MODE_DDGI)
        eng.set_ray_tile(*c["tile"])
        for frame in range(2):
            eng.probe_update(ddgi.make_settings(c["scene"], 8, time=2.0 * (frame + 1)))
        g_irr, g_dep = eng.read_tiles()
        ms = eng.last_update_ms()
    assert np.isfinite(g_irr).all() and np.isfinite(g_dep).all() and g_irr[..., :3].max() > 0
    assert (g_irr[..., 3] == 1.0).all() and ms["blend_ms"] > 0
    assert np.array_equal(g_irr[:, 0, 0], g_irr[:, 6, 6]) and np.array_equal(g_dep[:, 0, 0], g_dep[:, 14, 14])   # octahedral border
    oracle.set_ray_tile(*c["tile"])
    for first in (0, P - 24):
        irr, dep = oracle.new_tiles(f)
        for frame in range(2):
            oracle.ddgi_update(f, oracle.make_settings(c["scene"], 8, time=2.0 * (frame + 1)), frame, irr, dep, probes=(first, 24))
        assert np.array_equal(_bits(g_irr[first:first + 24]), _bits(irr[first:first + 24]))
        assert np.array_equal(_bits(g_dep[first:first + 24]), _bits(dep[first:first + 24]))
