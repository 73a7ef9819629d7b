"""Edge cases of the probe path on the GPU (through the C ABI), each against the oracle: ragged sizes,
several lights, zero bounces, empty batches, reconfiguration, and BASELINE's largest grid."""
import numpy as np
import pytest

from tests.common import CONFIGS, c3_oracle_albedo, shading_points

pytestmark = pytest.mark.gpu

FOUR_LIGHTS_CAVE = [  # the commented 4-light cave table, assets/shaders/structs.glsl:65-68
    (20.0, (1.0, 1.0, 1.0), (4, 17.5, 8.5)),
    (10.0, (1.0, 0.5, 0.1), (0, 2, 0)),
    (10.0, (0.1, 1.1, 1.0), (5, 0, 0)),
    (10.0, (1.1, 0.0, 1.1), (0, 5, 0)),
]
THREE_LIGHTS_CORNELL = [  # structs.glsl:83-86
    (10.0, (1.0, 0.1, 0.1), (0, 8, 13)),
    (10.0, (0.1, 0.1, 1.0), (0, 8, 9)),
    (10.0, (0.1, 1.0, 0.1), (-3, 8, 9)),
]


def _run(ddgi, counts, side, s, origin, scene, bounces=8, lights=None):
    with ddgi.ProbeEngine(ddgi.make_field(counts, side, s, origin), ddgi.make_settings(scene, bounces)) as eng:
        if lights is not None:
            eng.set_lights(scene, np.array(lights, dtype=ddgi.LIGHT_DTYPE))
        eng.generate_probe_rays(seed=1)
        eng.probe_update()
        return eng.read_textures()


def _oracle(oracle, counts, side, s, origin, scene, bounces=8, lights=None):
    f = oracle.make_field(counts, side, s, origin)
    rays = oracle.generate_probe_rays(f, oracle.new_rand_state(1))
    larr = None if lights is None else np.array(lights, dtype=oracle.LIGHT_DTYPE)
    return oracle.probe_update(f, oracle.make_settings(scene, bounces), rays, lights=larr)


@pytest.mark.parametrize("counts,side,s,origin,scene", [
    ((3, 1, 2), 7, 5, (1.4, 0.0, 1.0), 0),    # 25 rays/probe, 150 rays: nothing is a multiple of 64
    ((1, 1, 1), 3, 1, (0.0, 0.0, 15.0), 1),   # a single ray
    ((2, 3, 1), 9, 9, (0.0, 0.0, 0.0), 2),    # house, 81 rays/probe
    ((5, 2, 3), 6, 13, (1.4, 0.0, 1.0), 0),   # 5070 rays: more than one 4096-ray chunk, ragged tail
])
def test_ragged_sizes(ddgi, oracle, counts, side, s, origin, scene):
    a, d = _run(ddgi, counts, side, s, origin, scene)
    wa, _ = _oracle(oracle, counts, side, s, origin, scene)
    assert np.array_equal(a, wa) and not d.any()


@pytest.mark.parametrize("name,lights", [("cave_small", FOUR_LIGHTS_CAVE), ("c1_cornell", THREE_LIGHTS_CORNELL)])
def test_several_lights_ref_mode(ddgi, oracle, name, lights):
    """The dormant multi-light tables: per-light feelers, visible-light averaging,
    early return on the first occluded light (Q10)."""
    counts, side, s, origin, scene = CONFIGS[name]
    a, _ = _run(ddgi, counts, side, s, origin, scene, lights=lights)
    wa, _ = _oracle(oracle, counts, side, s, origin, scene, lights=lights)
    assert np.array_equal(a, wa)
    base, _ = _oracle(oracle, counts, side, s, origin, scene)
    assert not np.array_equal(wa, base)   # the table really changed the result


def test_zero_bounces_and_no_lights(ddgi, oracle):
    counts, side, s, origin, scene = CONFIGS["c1_cornell"]
    a, _ = _run(ddgi, counts, side, s, origin, scene, bounces=0)   # color / 0 = NaN -> rgba8 0 (Q14)
    wa, _ = _oracle(oracle, counts, side, s, origin, scene, bounces=0)
    assert np.array_equal(a, wa) and not a[..., :3].any() and (a[..., 3] == 255).all()
    a, _ = _run(ddgi, counts, side, s, origin, scene, lights=[])
    wa, _ = _oracle(oracle, counts, side, s, origin, scene, lights=np.zeros(0, dtype=oracle.LIGHT_DTYPE))
    assert np.array_equal(a, wa)


def test_empty_and_tiny_sample_batches(ddgi, oracle):
    counts, side, s, origin, scene = CONFIGS["c1_cornell"]
    with ddgi.ProbeEngine(ddgi.make_field(counts, side, s, origin), ddgi.make_settings(scene, 8)) as eng:
        with pytest.raises(ddgi.DDGIError) as ei:
            eng.probe_update()                       # no rays yet
        assert ei.value.code == -5
        eng.generate_probe_rays(seed=1)
        eng.probe_update()
        rgb, cage = eng.sample(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32))
        assert rgb.shape == (0, 3) and cage.shape == (0, 8)
        albedo, distance = eng.read_textures()
        pos, nrm = shading_points(np.random.default_rng(1), counts, side, origin, 1)
        rgb, cage = eng.sample(pos, nrm)
    want_rgb, want_cage = oracle.sample(oracle.make_field(counts, side, s, origin), albedo, distance, pos, nrm)
    assert np.array_equal(cage, want_cage) and np.array_equal(rgb.view(np.uint32), want_rgb.view(np.uint32))


def test_reconfigure_recreates_textures(ddgi, oracle):
    """RVPT::recreate_probe_textures (rvpt.cpp:661-755): new counts / rays / spacing on a live handle."""
    c1 = CONFIGS["c1_cornell"]
    c2 = CONFIGS["cave_odd"]
    with ddgi.ProbeEngine(ddgi.make_field(*c1[:4]), ddgi.make_settings(c1[4], 8)) as eng:
        eng.generate_probe_rays(seed=1)
        eng.probe_update()
        eng.configure(ddgi.make_field(*c2[:4]), ddgi.make_settings(c2[4], 8))
        assert not eng.read_textures()[0].any()          # re-created zeroed
        with pytest.raises(ddgi.DDGIError):
            eng.probe_update()                           # rays must be regenerated
        eng.generate_probe_rays(seed=1)                  # continues the rand() sequence (Q1)
        rays = eng.get_probe_rays()
        eng.probe_update()
        a, _ = eng.read_textures()
    st = oracle.new_rand_state(1)
    oracle.generate_probe_rays(oracle.make_field(*c1[:4]), st)
    f2 = oracle.make_field(*c2[:4])
    want_rays = oracle.generate_probe_rays(f2, st)
    assert rays.tobytes() == want_rays.tobytes()
    wa, _ = oracle.probe_update(f2, oracle.make_settings(c2[4], 8), want_rays)
    assert np.array_equal(a, wa)


def test_c5_scale_ddgi_mode(ddgi, oracle):
    """BASELINE config 5's grid on one GPU: 128x64x128 probes x 256 rays (2.7e8 rays), DDGI mode (no ray
    buffer: rays are generated in the kernel), 4 animated lights, two frames with hysteresis.
    Size-independent properties + an oracle check of sampled probes."""
    counts, side, s, origin, scene = (128, 64, 128), 1, 16, (1.4, 0.0, 1.0), 0
    lights = np.array(FOUR_LIGHTS_CAVE, dtype=ddgi.LIGHT_DTYPE)
    with ddgi.ProbeEngine(ddgi.make_field(counts, side, s, origin), ddgi.make_settings(scene, 8)) as eng:
        eng.set_mode(ddgi.MODE_DDGI)
        eng.set_lights(scene, lights)
        for frame in range(2):
            eng.probe_update(ddgi.make_settings(scene, 8, time=2.0 * frame))
        irr, dep = eng.read_tiles()
        ms = eng.last_update_ms()
    assert irr.shape == (128 * 64 * 128, 8, 8, 4) and np.isfinite(irr).all() and np.isfinite(dep).all()
    assert (irr[..., 3] == 1.0).all()   # (radiance may be slightly negative: type-7 albedo has a negative blue, intersection.glsl:933)
    assert np.array_equal(irr[:, 0, 0], irr[:, 6, 6]) and np.array_equal(dep[:, 15, 3], dep[:, 14, 12])  # border wrap
    assert (dep[..., 1] >= 0).all() and dep[..., 0].max() <= 1.5 * side * 1.0001                       # clamped distances
    assert ms["trace_ms"] > 0 and ms["blend_ms"] > 0
    f = oracle.make_field(counts, side, s, origin)
    rng = np.random.default_rng(9)
    import ctypes as C

    for p in rng.choice(128 * 64 * 128, size=6, replace=False):
        # run the oracle on a 1-probe window: offset the base pointers so that probe p lands on our buffers
        irr_buf = np.zeros((8 * 8 * 4,), np.float32)
        dep_buf = np.zeros((16 * 16 * 2,), np.float32)
        for frame in range(2):
            oracle.lib().oracle_ddgi_update(
                C.byref(f), C.byref(oracle.make_settings(scene, 8, time=2.0 * frame)),
                np.ascontiguousarray(lights).ctypes.data_as(C.c_void_p), len(lights), C.c_uint32(frame),
                C.c_void_p(irr_buf.ctypes.data - int(p) * 8 * 8 * 4 * 4), C.c_void_p(dep_buf.ctypes.data - int(p) * 16 * 16 * 2 * 4),
                None, int(p), 1, 1)
        assert np.array_equal(irr[p].reshape(-1).view(np.uint32), irr_buf.view(np.uint32)), f"probe {p}"
        assert np.array_equal(dep[p].reshape(-1).view(np.uint32), dep_buf.view(np.uint32)), f"probe {p}"


def test_large_ref_grid_beyond_the_bake_box(ddgi, oracle):
    """A REF-mode grid that sticks out of the cave's baked box on every side (probes inside solid
    rock, above the terrain, and in the fbm floor band outside the box): sampled oracle check."""
    counts, side, s, origin, scene = (96, 48, 96), 1, 8, (1.4, 0.0, 1.0), 0
    with ddgi.ProbeEngine(ddgi.make_field(counts, side, s, origin), ddgi.make_settings(scene, 8)) as eng:
        eng.generate_probe_rays(seed=1)
        eng.probe_update()
        a, d = eng.read_textures()
    assert not d.any() and (a[..., 3] == 255).all()
    f = oracle.make_field(counts, side, s, origin)
    rays = oracle.generate_probe_rays(f, oracle.new_rand_state(1))
    cx, cy, cz = counts
    rng = np.random.default_rng(4)
    probes = list(rng.choice(cx * cy * cz, size=40, replace=False))
    # force some probes into the floor band outside the box: y index 0..3 (world y -23..-20), x index 0..4
    probes += [y * cx * cz + z * cx + x for y in (0, 2, 3) for z in (1, 50, 94) for x in (0, 3, 95)]
    want = oracle.probe_update_probes(f, oracle.make_settings(scene, 8), rays, probes)
    fld = ddgi.make_field(counts, side, s, origin)
    for p in probes:
        x0, y0 = ddgi.probe_tile_origin(fld, int(p))
        assert np.array_equal(a[y0:y0 + s, x0:x0 + s], want[y0:y0 + s, x0:x0 + s]), f"probe {p}"


_SAFETY_NET_SCRIPT = r"""
import sys
sys.path.insert(0, {root!r})
import numpy as np
import ddgi_amd as ddgi
counts, side, s, origin, scene = (2, 2, 2), 6, 8, (0.0, 0.0, 15.0), 1
with ddgi.ProbeEngine(ddgi.make_field(counts, side, s, origin), ddgi.make_settings(scene, 4)) as eng:
    eng.set_tuning("trace_kernel", 3)
    eng.generate_probe_rays(seed=1)
    eng.probe_update()
    want = eng.read_textures()[0]
    eng.set_tuning("ablate", 8)          # the queue kernel drops every march it posts: rays never finish
    eng.probe_update()
    try:
        eng.synchronize()
        print("NO-ERROR")
    except ddgi.DDGIError as exc:
        print("ABORT-REPORTED" if "aborted" in str(exc) else "OTHER " + str(exc))
    eng.set_tuning("ablate", 0)          # the flag was reported once and cleared: the handle is usable again
    eng.probe_update()
    eng.synchronize()
    print("RECOVERED" if np.array_equal(eng.read_textures()[0], want) else "MISMATCH")
"""


@pytest.mark.gpu
def test_queue_kernel_safety_net_reports_instead_of_hanging(ddgi):
    """Fault injection (profiling build only, tuning "ablate" = 8: the queue kernel drops every march it posts,
    so rays never finish): the kernel's bounded waits trip after about a second, every wave leaves, and the
    next synchronising call returns an error instead of textures — once; a later clean update succeeds.
    The release library has no such switch."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with ddgi.ProbeEngine(ddgi.make_field((2, 2, 2), 6, 8, (0.0, 0.0, 15.0)), ddgi.make_settings(1, 4)) as eng:
        with pytest.raises(ddgi.DDGIError, match="unknown tuning key"):
            eng.set_tuning("ablate", 8)
    env = dict(os.environ, DDGI_LIB=ddgi.build_profiling_library())
    res = subprocess.run([sys.executable, "-c", _SAFETY_NET_SCRIPT.format(root=root)], env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "ABORT-REPORTED" in res.stdout and "RECOVERED" in res.stdout, res.stdout + res.stderr


@pytest.mark.gpu
def test_update_timing_can_be_switched_off(ddgi):
    """Tuning "timing": an update records its events (two in REF mode, three in DDGI mode) unless told not to; the timing queries
    then say DDGI_ERR_NOT_READY instead of returning stale numbers, and the textures are the same either way."""
    counts, side, s, origin, _ = CONFIGS["c2_cornell"]
    with ddgi.ProbeEngine(ddgi.make_field(counts, side, s, origin), ddgi.make_settings(1, 4)) as eng:
        eng.generate_probe_rays(seed=1)
        eng.probe_update()
        timed = eng.read_textures()
        ms = eng.last_update_ms()
        assert ms["trace_ms"] > 0 and ms["blend_ms"] == 0 and abs(ms["total_ms"] - ms["trace_ms"]) < 1e-6   # REF mode: one kernel
        eng.set_tuning("timing", 0)
        eng.probe_update()
        untimed = eng.read_textures()
        with pytest.raises(ddgi.DDGIError) as err:
            eng.last_update_ms()
        assert "not timed" in str(err.value)
        with pytest.raises(ddgi.DDGIError):
            eng.update_history_ms(4)
        eng.set_tuning("timing", 1)
        eng.probe_update()
        assert eng.last_update_ms()["trace_ms"] > 0
        tr, bl = eng.update_history_ms(1)
        assert len(tr) == 1 and tr[0] > 0 and bl[0] == 0
        assert all(np.array_equal(a, b) for a, b in zip(timed, untimed))
        eng.set_mode(ddgi.MODE_DDGI)
        eng.probe_update()
        ms = eng.last_update_ms()
        assert ms["blend_ms"] > 0 and ms["total_ms"] >= ms["trace_ms"] + ms["blend_ms"] - 1e-3


@pytest.mark.parametrize("time", [0.0, 46.0])
def test_full_size_c3_four_lights_every_texel(ddgi, oracle, time):
    """The headline grid under the reference's 4-light cave table (S-Dyn's lights), REF mode, against the oracle: EVERY one of the
    4 194 304 texels with the table as written, and an eighth of the probes (spread over the grid) with the table as
    update_lights(time = 46) moves it (probe_pass.comp:217-251; REF mode itself does not animate — the moved table is handed to
    both sides).  With several lights the event deals with the feelers that the per-light tables of k_light_visibility decide without
    marching them (wf_event: decided_feelers) — a LIT patch ends on the nearest light sphere on its ray, a SHADOW patch counts only
    when no OTHER light's sphere lies on the ray; a hole in either argument shows as a handful of texels here."""
    counts, side, s, origin, scene = CONFIGS["c3_cave"]
    base = np.array(FOUR_LIGHTS_CAVE, dtype=oracle.LIGHT_DTYPE)
    table = base if time == 0.0 else oracle.update_lights(scene, time, base)
    assert time == 0.0 or not np.array_equal(table["pos"], base["pos"])
    lights = np.array([(float(l["intensity"]), tuple(l["col"]), tuple(l["pos"])) for l in table], dtype=ddgi.LIGHT_DTYPE)
    with ddgi.ProbeEngine(ddgi.make_field(counts, side, s, origin), ddgi.make_settings(scene, 8)) as eng:
        eng.set_lights(scene, lights)
        eng.generate_probe_rays(seed=1)
        eng.probe_update()
        got, _ = eng.read_textures()
        eng.set_tuning("light_vis", 0)          # the same without the tables: every feeler marched
        eng.probe_update()
        marched, _ = eng.read_textures()
    want = c3_oracle_albedo(oracle, "pinned", seed=1, lights=table, fraction=1 if time == 0.0 else 8)
    mask = want[..., 3] == 255
    assert mask.all() if time == 0.0 else mask.sum() >= got.shape[0] * got.shape[1] // 10
    assert np.array_equal(marched[mask], want[mask]), "marched feelers differ from the oracle"
    nbad = int((got[mask] != want[mask]).any(axis=-1).sum())
    assert nbad == 0, f"time {time}: {nbad} of {int(mask.sum())} compared texels differ from the oracle"
