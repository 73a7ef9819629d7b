"""DDGI mode (the reference's dormant pieces switched on: Fibonacci rays, octahedral irradiance /
depth tiles with hysteresis, Chebyshev-weighted cage sample, animated lights).  There is no live
reference behaviour to compare with, so the HIP path is validated against the
oracle's restatement of the same specification: float tiles and sampled rgb BIT-EXACT (both sides
evaluate the pinned arithmetic in the same order), cage indices bit-exact."""
import numpy as np
import pytest

from tests.common import CONFIGS, shading_points

pytestmark = pytest.mark.gpu

FOUR_LIGHTS_CAVE = [  # the commented 4-light cave table, assets/shaders/structs.glsl:65-68
    (20.0, (1.0, 1.0, 1.0), (4, 17.5, 8.5)),
    (10.0, (1.0, 0.5, 0.1), (0, 2, 0)),
    (10.0, (0.1, 1.1, 1.0), (5, 0, 0)),
    (10.0, (1.1, 0.0, 1.1), (0, 5, 0)),
]


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.mark.parametrize("name,lights,frames", [("c1_cornell", None, 3), ("cave_small", None, 3), ("cave_small", FOUR_LIGHTS_CAVE, 2)])
def test_ddgi_update_and_sample_bit_exact_vs_oracle(ddgi, oracle, name, lights, frames):
    counts, side, s, origin, scene = CONFIGS[name]
    f = oracle.make_field(counts, side, s, origin)
    irr, dep = oracle.new_tiles(f)
    larr = None if lights is None else np.array(lights, dtype=oracle.LIGHT_DTYPE)
    pos, nrm = shading_points(np.random.default_rng(17), counts, side, origin, 2048)
    with ddgi.ProbeEngine(ddgi.make_field(counts, side, s, origin), ddgi.make_settings(scene, 8)) as eng:
        eng.set_mode(ddgi.MODE_DDGI)
        if larr is not None:
            eng.set_lights(scene, larr)
        for frame in range(frames):
            st = ddgi.make_settings(scene, 8, time=2.0 * (frame + 1))   # RVPT::update adds 2 per frame
            eng.probe_update(st)
            oracle.ddgi_update(f, oracle.make_settings(scene, 8, time=2.0 * (frame + 1)), frame, irr, dep, lights=larr)
            g_irr, g_dep = eng.read_tiles()
            assert np.array_equal(_bits(g_irr), _bits(irr)), f"irradiance tiles differ at frame {frame}"
            assert np.array_equal(_bits(g_dep), _bits(dep)), f"depth tiles differ at frame {frame}"
        rgb, cage = eng.sample(pos, nrm)
    want_rgb, want_cage = oracle.ddgi_sample(f, irr, dep, pos, nrm)
    assert np.array_equal(cage, want_cage)
    assert np.array_equal(_bits(rgb), _bits(want_rgb))
    assert np.isfinite(irr).all() and irr[..., :3].max() > 0
    assert (cage[:, 0] >= 0).mean() > 0.05


@pytest.mark.parametrize("name", ["c2_cornell", "cave_odd"])
def test_ddgi_sample_paths_agree_on_edge_cages_and_grouped_batches(ddgi, oracle, name):
    """The Chebyshev sampler on a batch large enough to be grouped by cage (k_sample_count / k_sample_scan_runs / k_sample_place) and
    as it comes: points everywhere (some outside), crowded into one cage, and on the field's last cage layer in x, whose +x corners'
    probe indices wrap into the next z row (Q4: slab_slot_of_corner must decode those like the reference does).  Both paths equal
    the oracle bit for bit, cage indices included."""
    counts, side, s, origin, scene = CONFIGS[name]
    rng = np.random.default_rng(29)
    spread, nrm = shading_points(rng, counts, side, origin, 24000)
    o = np.asarray(origin, dtype=np.float32)
    crowd = (rng.uniform(0.05, 0.95, size=(5000, 3)) * side + o).astype(np.float32)
    edge = (rng.uniform(-0.5, 0.5, size=(5000, 3)) * np.float32(side) * np.asarray(counts, dtype=np.float32) + o).astype(np.float32)
    edge[:, 0] = o[0] + side * (counts[0] // 2 - 0.5)  # the last cage layer in x
    pos = np.concatenate([spread, crowd, edge]).astype(np.float32)
    nrm = np.concatenate([nrm, rng.normal(size=(len(pos) - len(nrm), 3)).astype(np.float32)])
    f = oracle.make_field(counts, side, s, origin)
    irr, dep = oracle.new_tiles(f)
    with ddgi.ProbeEngine(ddgi.make_field(counts, side, s, origin), ddgi.make_settings(scene, 8)) as eng:
        eng.set_mode(ddgi.MODE_DDGI)
        for frame in range(2):
            eng.probe_update(ddgi.make_settings(scene, 8, time=2.0 * (frame + 1)))
            oracle.ddgi_update(f, oracle.make_settings(scene, 8, time=2.0 * (frame + 1)), frame, irr, dep)
        got = {}
        for group in (1, 0):
            eng.set_tuning("sample_group", group)
            got[group] = eng.sample(pos, nrm)
        # the same batch in cage order (what a frame's pixels are): the grouping finds every run in a few bins and writes no permutation
        # (k_sample_place: *perm_off) — same results at the same indices
        cell = np.floor((pos - o) / np.float32(side)).astype(np.int64)
        order = np.lexsort((cell[:, 0], cell[:, 1], cell[:, 2]))
        eng.set_tuning("sample_group", 1)
        rgb_o, cage_o = eng.sample(pos[order], nrm[order])
    want_rgb, want_cage = oracle.ddgi_sample(f, irr, dep, pos, nrm)
    for group, (rgb, cage) in got.items():
        assert np.array_equal(cage, want_cage), f"sample_group {group}"
        assert np.array_equal(_bits(rgb), _bits(want_rgb)), f"sample_group {group}"
    assert np.array_equal(cage_o, want_cage[order]) and np.array_equal(_bits(rgb_o), _bits(want_rgb[order]))
    inside = want_cage[:, 0] >= 0
    assert 0.02 < inside.mean() < 1.0, inside.mean()


def test_ddgi_temporal_behaviour_and_mode_switch(ddgi):
    counts, side, s, origin, scene = CONFIGS["c1_cornell"]
    with ddgi.ProbeEngine(ddgi.make_field(counts, side, s, origin, hysteresis=0.9), ddgi.make_settings(scene, 8)) as eng:
        eng.set_mode(ddgi.MODE_DDGI)
        eng.probe_update()
        a, _ = eng.read_tiles()
        eng.probe_update()
        b, _ = eng.read_tiles()
        assert not np.array_equal(a, b)            # frames differ: rotation + RNG carry the frame index
        eng.set_frame(0)
        eng.set_mode(ddgi.MODE_REF)                 # back to REF: textures re-created, rays needed again
        with pytest.raises(ddgi.DDGIError):
            eng.read_tiles()
        eng.generate_probe_rays(seed=1)
        eng.probe_update()
        albedo, _ = eng.read_textures()
        assert albedo[..., :3].any()
        eng.set_mode(ddgi.MODE_DDGI)
        eng.probe_update()
        c, _ = eng.read_tiles()
        assert np.array_equal(a, c)                # same frame index, zeroed tiles -> same first update
    # border texels equal their octahedral wrap source
    assert np.array_equal(a[:, 0, 0], a[:, 6, 6]) and np.array_equal(a[:, 0, 3], a[:, 1, 4]) and np.array_equal(a[:, 2, 7], a[:, 5, 6])


def test_ddgi_sharded_slabs(ddgi, oracle):
    counts, side, s, origin, scene = CONFIGS["cave_small"]
    f = oracle.make_field(counts, side, s, origin)
    irr, dep = oracle.new_tiles(f)
    oracle.ddgi_update(f, oracle.make_settings(scene, 8, time=0.0), 0, irr, dep)
    cx, cy, cz = counts
    for rank in range(2):
        with ddgi.ProbeEngine(ddgi.make_field(counts, side, s, origin), ddgi.make_settings(scene, 8), rank=rank, world=2) as eng:
            eng.set_mode(ddgi.MODE_DDGI)
            eng.probe_update(ddgi.make_settings(scene, 8, time=0.0))
            g_irr, g_dep = eng.read_tiles()
        mine = np.array([(p % (cx * cz)) // cx // (cz // 2) == rank for p in range(cx * cy * cz)])
        assert np.array_equal(_bits(g_irr[mine]), _bits(irr[mine])) and not g_irr[~mine].any()
        assert np.array_equal(_bits(g_dep[mine]), _bits(dep[mine])) and not g_dep[~mine].any()


@pytest.mark.gpu
@pytest.mark.parametrize("counts,s", [((3, 3, 3), 5), ((2, 1, 3), 3), ((5, 2, 2), 8), ((1, 1, 1), 1)])
def test_blend_kernels_agree_on_ragged_shapes(ddgi, oracle, counts, s):
    """Probe counts that are not a multiple of the 8-probe record group and odd ray counts: the
    MFMA blend (k_blend_weights + k_probe_blend_depth / _irr / _mfma), the one-probe-per-workgroup blend
    (DDGI_BLEND_KERNEL=probe) and the oracle give the same tiles, bit for bit, over three frames."""
    import os
    side, origin, scene = 6, (0.0, 0.0, 15.0), 1
    f = oracle.make_field(counts, side, s, origin, hysteresis=0.7)
    st = oracle.make_settings(scene, 4)
    n_probes = counts[0] * counts[1] * counts[2]
    o_irr = np.zeros((n_probes, 8, 8, 4), dtype=np.float32)
    o_dep = np.zeros((n_probes, 16, 16, 2), dtype=np.float32)
    results = {}
    for kernel in ("scalar", "probe", "division"):
        if kernel != "scalar":
            os.environ["DDGI_BLEND_KERNEL"] = kernel
        try:
            with ddgi.ProbeEngine(ddgi.make_field(counts, side, s, origin, hysteresis=0.7), ddgi.make_settings(scene, 4)) as eng:
                eng.set_mode(ddgi.MODE_DDGI)
                for _ in range(3):
                    eng.probe_update()
                results[kernel] = eng.read_tiles()
        finally:
            os.environ.pop("DDGI_BLEND_KERNEL", None)
    for frame in range(3):
        oracle.ddgi_update(f, st, frame, o_irr, o_dep)
    for kernel, (irr, dep) in results.items():
        assert np.array_equal(_bits(irr), _bits(o_irr)), kernel
        assert np.array_equal(_bits(dep), _bits(o_dep)), kernel


@pytest.mark.gpu
def test_persistent_depth_blend_agrees_with_the_per_probe_kernel_on_a_large_grid(ddgi):
    """More than 8 192 probes x 256 rays: the depth tiles are blended by the persistent kernel that keeps its weight tile in
    registers (k_probe_blend_depth_res), 576 groups over one workgroup per CU (so the workgroups' loops end unevenly).  The
    one-probe-per-workgroup kernel (DDGI_BLEND_KERNEL=probe; checked against the oracle above) must give the same bits,
    with hysteresis over two frames."""
    import os
    counts, side, s, origin, scene = (24, 16, 24), 2, 16, (0.0, 0.0, 0.0), 0
    results = {}
    # "division": the MFMA kernels with the compiler's `/` for every quotient — the path a group takes whose sums lie outside the
    # short division's domain (pm::div_prepared; tests/exact_rcp_sqrt_check.hip), which no scene here reaches by itself
    for kernel in ("mfma", "probe", "division"):
        if kernel != "mfma":
            os.environ["DDGI_BLEND_KERNEL"] = kernel
        try:
            with ddgi.ProbeEngine(ddgi.make_field(counts, side, s, origin, hysteresis=0.7), ddgi.make_settings(scene, 3)) as eng:
                eng.set_mode(ddgi.MODE_DDGI)
                for _ in range(2):
                    eng.probe_update()
                results[kernel] = eng.read_tiles()
        finally:
            os.environ.pop("DDGI_BLEND_KERNEL", None)
    assert results["mfma"][1].any() and results["mfma"][0].any()
    for other in ("probe", "division"):
        assert np.array_equal(_bits(results["mfma"][0]), _bits(results[other][0])), other
        assert np.array_equal(_bits(results["mfma"][1]), _bits(results[other][1])), other
