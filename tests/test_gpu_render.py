"""SURVEY.md §8(f) row 1 — the primary-visibility consumer (camera rays + the reference's
integrators over the probe field): k_render_primary against the oracle's restatement of
compute_pass.comp:main / camera.glsl / integrators.glsl.  Bit-exact (rgba8 image and the
unquantised float rgb) in PINNED arithmetic, all six integrators, both camera models, both modes."""
import numpy as np
import pytest

from tests.common import CONFIGS

pytestmark = pytest.mark.gpu

CAMS = {
    # scene config -> (origin, rotation_deg) looking into the scene
    "c2_cornell": ((0.5, 1.0, -12.0), (0.0, 0.0, 0.0)),
    "cave_small": ((2.0, 3.0, -6.0), (25.0, 10.0, 0.0)),
}


def _settings(mod, scene, w, h, render_mode, camera_mode=0, time=0.0):
    st = mod.make_settings(scene, 8, time=time)
    st.screen_width, st.screen_height = w, h
    st.render_mode, st.camera_mode = render_mode, camera_mode
    return st


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.mark.parametrize("name", ["c2_cornell", "cave_small"])
def test_render_ref_mode_all_integrators_bit_exact(ddgi, oracle, name):
    counts, side, s, origin, scene = CONFIGS[name]
    w, h = 160, 90
    cam = ddgi.make_camera(*CAMS[name], fov_deg=75.0, aspect=w / h)
    with ddgi.ProbeEngine(ddgi.make_field(counts, side, s, origin), ddgi.make_settings(scene, 8)) as eng:
        eng.generate_probe_rays(seed=1)
        eng.probe_update()
        albedo, distance = eng.read_textures()
        f = oracle.make_field(counts, side, s, origin)
        for render_mode in range(6):
            for camera_mode in ((0, 1) if render_mode in (0, 4) else (0,)):
                img, rgb = eng.render(cam, _settings(ddgi, scene, w, h, render_mode, camera_mode), want_float=True)
                want_img, want_rgb = oracle.render(f, _settings(oracle, scene, w, h, render_mode, camera_mode), cam, albedo, distance,
                                                   want_float=True)
                same = (_bits(rgb) == _bits(want_rgb)) | (np.isnan(rgb) & np.isnan(want_rgb))
                assert same.all(), f"mode {render_mode} camera {camera_mode}: {(~same).sum()} float channels differ"
                assert np.array_equal(img, want_img)
                if render_mode == 0 and camera_mode == 0:
                    assert img[..., :3].std() > 5        # a real picture, not a constant
                    assert (img[..., 3] == 255).all()


def test_render_ddgi_mode_bit_exact(ddgi, oracle):
    name = "cave_small"
    counts, side, s, origin, scene = CONFIGS[name]
    w, h = 128, 72
    cam = ddgi.make_camera(*CAMS[name], fov_deg=60.0, aspect=w / h)
    f = oracle.make_field(counts, side, s, origin)
    irr, dep = oracle.new_tiles(f)
    with ddgi.ProbeEngine(ddgi.make_field(counts, side, s, origin), ddgi.make_settings(scene, 8)) as eng:
        eng.set_mode(ddgi.MODE_DDGI)
        for frame in range(2):
            eng.probe_update(ddgi.make_settings(scene, 8, time=2.0 * frame))
            oracle.ddgi_update(f, oracle.make_settings(scene, 8, time=2.0 * frame), frame, irr, dep)
        for render_mode in (0, 2):
            img, rgb = eng.render(cam, _settings(ddgi, scene, w, h, render_mode, time=2.0), want_float=True)
            lights = oracle.update_lights(scene, 2.0, oracle.shipped_lights(scene))   # DDGI mode animates the lights
            want_img, want_rgb = oracle.render(f, _settings(oracle, scene, w, h, render_mode, time=2.0), cam, irr, dep,
                                               ddgi_mode=True, lights=lights, want_float=True)
            assert np.array_equal(_bits(rgb), _bits(want_rgb))
            assert np.array_equal(img, want_img)


@pytest.mark.parametrize("name", ["c2_cornell", "cave_small"])
def test_probe_visualisation_and_debug_views_bit_exact(ddgi, oracle, name):
    """SURVEY.md 8(f) row 2 on the GPU: RenderSettings::visualize_probes (probes as cyan spheres, integrators.glsl:45-65,
    180-199), the probe-texture blit (compute_pass.comp:116-124, 185-190; render_mode 6) and the cage-index colouring
    (README.md:89-91; render_mode 7), each against the oracle's restatement, bit for bit."""
    counts, side, s, origin, scene = CONFIGS[name]
    w, h = 160, 90
    cam = ddgi.make_camera(*CAMS[name], fov_deg=75.0, aspect=w / h)
    f = oracle.make_field(counts, side, s, origin)
    with ddgi.ProbeEngine(ddgi.make_field(counts, side, s, origin), ddgi.make_settings(scene, 8)) as eng:
        eng.generate_probe_rays(seed=1)
        eng.probe_update()
        albedo, distance = eng.read_textures()
        plain = eng.render(cam, _settings(ddgi, scene, w, h, 0))
        for render_mode, vis in ((0, 1), (2, 1), (6, 0), (7, 0), (3, 1)):
            st, ost = _settings(ddgi, scene, w, h, render_mode), _settings(oracle, scene, w, h, render_mode)
            st.visualize_probes = ost.visualize_probes = vis
            img, rgb = eng.render(cam, st, want_float=True)
            want_img, want_rgb = oracle.render(f, ost, cam, albedo, distance, want_float=True)
            assert np.array_equal(_bits(rgb), _bits(want_rgb)), f"mode {render_mode}"
            assert np.array_equal(img, want_img), f"mode {render_mode}"
            if render_mode == 0:
                cyan = (img[..., :3] == (0, 255, 255)).all(axis=-1)
                assert 0.002 < cyan.mean() < 0.5          # probes are visible, and are not the whole picture
                assert np.array_equal(img[~cyan], plain[~cyan])
            if render_mode == 6:
                # the blit is the probe texture resampled to the screen (nearest texel, top row first)
                H, W = albedo.shape[:2]
                ys = (np.arange(h, dtype=np.float32) * np.float32(H) / np.float32(h)).astype(np.int64)
                xs = (np.arange(w, dtype=np.float32) * np.float32(W) / np.float32(w)).astype(np.int64)
                assert np.array_equal(img[..., :3], albedo[ys][:, xs][..., :3])
            if render_mode == 7:
                assert len(np.unique(img.reshape(-1, 4), axis=0)) > 3     # several cages are on screen
            if render_mode == 3:
                assert np.array_equal(img, eng.render(cam, _settings(ddgi, scene, w, h, 3)))   # the flag only acts on integrators 0 and 2


def test_debug_views_in_ddgi_mode(ddgi, oracle):
    name = "cave_small"
    counts, side, s, origin, scene = CONFIGS[name]
    w, h = 96, 54
    cam = ddgi.make_camera(*CAMS[name], fov_deg=60.0, aspect=w / h)
    f = oracle.make_field(counts, side, s, origin)
    irr, dep = oracle.new_tiles(f)
    with ddgi.ProbeEngine(ddgi.make_field(counts, side, s, origin), ddgi.make_settings(scene, 8)) as eng:
        eng.set_mode(ddgi.MODE_DDGI)
        eng.probe_update(ddgi.make_settings(scene, 8, time=0.0))
        oracle.ddgi_update(f, oracle.make_settings(scene, 8, time=0.0), 0, irr, dep)
        lights = oracle.update_lights(scene, 0.0, oracle.shipped_lights(scene))
        for render_mode, vis in ((0, 1), (7, 0), (6, 0)):
            st, ost = _settings(ddgi, scene, w, h, render_mode), _settings(oracle, scene, w, h, render_mode)
            st.visualize_probes = ost.visualize_probes = vis
            img = eng.render(cam, st)
            want = oracle.render(f, ost, cam, irr, dep, ddgi_mode=True, lights=lights)
            assert np.array_equal(img, want), f"mode {render_mode}"
