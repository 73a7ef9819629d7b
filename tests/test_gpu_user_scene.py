"""baked scenes on disk and user scenes (scene id 3)."""
import ctypes as C

import numpy as np
import pytest

from tests.common import CONFIGS


def test_scene_file_roundtrip_matches_host_bake(ddgi, tmp_path):
    """CPU: the DDGIVOX1 file of each built-in scene holds exactly the bake the kernels traverse."""
    for scene in (0, 1, 2):
        path = tmp_path / f"scene{scene}.ddgivox"
        ddgi.scene_save(scene, str(path))
        lo, dim, types = ddgi.read_scene_file(str(path))
        rng = np.random.default_rng(scene)
        for _ in range(300):
            x, y, z = (int(rng.integers(lo[a], lo[a] + dim[a])) for a in range(3))
            assert types[z - lo[2], y - lo[1], x - lo[0]] == ddgi.scene_block_at(scene, x, y, z)
        assert types.max() <= 13 and (types > 0).any()


@pytest.mark.gpu
def test_cornell_loaded_as_user_scene_reproduces_scene_1(ddgi, oracle, tmp_path):
    counts, side, s, origin, _ = CONFIGS["c1_cornell"]
    path = tmp_path / "cornell.ddgivox"
    ddgi.scene_save(1, str(path))
    light = np.array([(15.0, (1, 1, 1), (0, 8, 13))], dtype=ddgi.LIGHT_DTYPE)   # structs.glsl:81
    with ddgi.ProbeEngine(ddgi.make_field(counts, side, s, origin), ddgi.make_settings(1, 8)) as eng:
        eng.generate_probe_rays(seed=1)
        eng.probe_update()
        want, _ = eng.read_textures()
        with pytest.raises(ddgi.DDGIError):
            eng.probe_update(ddgi.make_settings(3, 8))    # no user scene loaded yet
        eng.load_scene(str(path))
        eng.set_lights(3, light)
        eng.probe_update(ddgi.make_settings(3, 8))
        got, _ = eng.read_textures()
    assert np.array_equal(got, want)


@pytest.mark.gpu
def test_random_user_scene_vs_oracle(ddgi, oracle):
    """A caller-made voxel scene: random blocks of every type in an 18^3 box, two lights."""
    rng = np.random.default_rng(12)
    lo = (-9, -9, 6)
    types = np.zeros((18, 18, 18), dtype=np.uint8)                  # [z, y, x]
    mask = rng.random(types.shape) < 0.08
    types[mask] = rng.integers(1, 14, size=int(mask.sum()))
    types[0, :, :] = 5                                              # a floor slab at z = lo.z ... (extrudes outwards)
    types[:, 0, :] = 10
    counts, side, s, origin = (3, 3, 3), 4, 8, (0.0, 0.0, 15.0)
    lights = np.array([(12.0, (1.0, 0.9, 0.8), (0.5, 3.5, 14.5)), (6.0, (0.3, 0.5, 1.0), (-4.5, -2.5, 18.5))], dtype=ddgi.LIGHT_DTYPE)
    with ddgi.ProbeEngine(ddgi.make_field(counts, side, s, origin), ddgi.make_settings(3, 8)) as eng:
        eng.set_scene_grid(lo, types)
        eng.set_lights(3, lights)
        eng.generate_probe_rays(seed=1)
        eng.probe_update()
        got, _ = eng.read_textures()
    lo_a = (C.c_int32 * 3)(*lo)
    dim_a = (C.c_int32 * 3)(18, 18, 18)
    oracle.lib().oracle_set_user_scene(lo_a, dim_a, types.ctypes.data_as(C.c_void_p))
    f = oracle.make_field(counts, side, s, origin)
    rays = oracle.generate_probe_rays(f, oracle.new_rand_state(1))
    want, _ = oracle.probe_update(f, oracle.make_settings(3, 8), rays, lights=np.array(lights, dtype=oracle.LIGHT_DTYPE))
    assert np.array_equal(got, want)
    assert got[..., :3].any()


@pytest.mark.gpu
def test_large_user_scene_vs_oracle(ddgi, oracle):
    """A 96x40x96 voxel scene (46 KB of occupancy bits next to the ray pool and the queues in LDS):
    a rough terrain with pillars, three lights, probes in the open space above it."""
    rng = np.random.default_rng(5)
    nx, ny, nz = 96, 40, 96
    lo = (-48, -20, -48)
    types = np.zeros((nz, ny, nx), dtype=np.uint8)                   # [z, y, x]
    height = (6 + 4 * np.sin(np.arange(nx)[None, :] / 9.0) + 3 * np.cos(np.arange(nz)[:, None] / 7.0)).astype(int)
    for y in range(ny):
        types[:, y, :][y < height] = 1 + (y % 13)
    pillars = rng.integers(4, 92, size=(40, 2))
    for px, pz in pillars:
        types[pz, :30, px] = rng.integers(1, 14)
    counts, side, s, origin = (4, 2, 4), 6, 6, (1.0, 8.0, 1.0)
    lights = np.array([(14.0, (1.0, 0.9, 0.8), (0.5, 15.5, 0.5)), (9.0, (0.3, 0.5, 1.0), (-20.5, 12.5, 18.5)),
                       (9.0, (1.0, 0.4, 0.3), (22.5, 10.5, -15.5))], dtype=ddgi.LIGHT_DTYPE)
    with ddgi.ProbeEngine(ddgi.make_field(counts, side, s, origin), ddgi.make_settings(3, 6)) as eng:
        eng.set_scene_grid(lo, types)
        eng.set_lights(3, lights)
        eng.generate_probe_rays(seed=1)
        eng.probe_update()
        got, _ = eng.read_textures()
    oracle.lib().oracle_set_user_scene((C.c_int32 * 3)(*lo), (C.c_int32 * 3)(nx, ny, nz), types.ctypes.data_as(C.c_void_p))
    f = oracle.make_field(counts, side, s, origin)
    rays = oracle.generate_probe_rays(f, oracle.new_rand_state(1))
    want, _ = oracle.probe_update(f, oracle.make_settings(3, 6), rays, lights=np.array(lights, dtype=oracle.LIGHT_DTYPE))
    assert np.array_equal(got, want)
    assert got[..., :3].any()
