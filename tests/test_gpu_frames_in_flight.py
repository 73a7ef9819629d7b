"""Frames in flight (tuning "frames_in_flight"; the reference's MAX_FRAMES_IN_FLIGHT = 2, src/rvpt/rvpt.h:23, with a fence
per frame, rvpt.cpp:277-278): an update submitted as the same work as its predecessor, while the predecessor's launch still
runs, is continued by that launch's resident workgroups into the NEXT texture pair of the handle's ring
(csrc/ddgi_engine.cpp: ddgi_probe_update; csrc/ddgi_trace_wf.hip: k_probe_trace_aq).

Results must not change by a bit — against the oracle, and with data that changes from frame to frame: a ray that landed in
the wrong pair, or an update continued with its predecessor's inputs, leaves the previous frame's texels behind.
"""
import numpy as np
import pytest

from tests.common import CONFIGS, c3_oracle_albedo, shading_points

pytestmark = pytest.mark.gpu


def _engine(ddgi, name, **kw):
    counts, side, s, origin, scene = CONFIGS[name]
    return ddgi.ProbeEngine(ddgi.make_field(counts, side, s, origin), ddgi.make_settings(scene, 8), **kw)


def _oracle_albedo(oracle, name, seed):
    counts, side, s, origin, scene = CONFIGS[name]
    f = oracle.make_field(counts, side, s, origin)
    rays = oracle.generate_probe_rays(f, oracle.new_rand_state(seed))
    return oracle.probe_update(f, oracle.make_settings(scene, 8), rays)[0]


@pytest.mark.parametrize("fif", [1, 2, 3, 4, 8])
@pytest.mark.parametrize("name", ["cave_small", "c2_cornell"])
def test_back_to_back_updates_with_new_rays_every_other_group(ddgi, oracle, name, fif):
    """Groups of back-to-back updates (nothing between them: the continuation's precondition), new ray jitter before every
    group — every pair of the ring holds the PREVIOUS jitter's texels when a group starts, so a texel written to the wrong
    pair, or not written at all, shows."""
    want = {seed: _oracle_albedo(oracle, name, seed) for seed in (1, 2, 3)}
    assert not np.array_equal(want[1], want[2])
    with _engine(ddgi, name) as eng:
        eng.set_tuning("frames_in_flight", fif)
        assert eng.get_tuning("texture_pairs") == fif
        for seed, n_updates in ((1, 2 * fif), (2, fif), (3, fif + 1), (1, 1), (2, 2 * fif + 1)):
            eng.generate_probe_rays(seed=seed, reseed=True)
            for _ in range(n_updates):
                eng.probe_update()
            albedo, distance = eng.read_textures()
            assert np.array_equal(albedo, want[seed]) and not distance.any()


def test_every_update_of_a_group_is_complete_behind_its_own_launch(ddgi, oracle):
    """Consumers between the updates of a group (a host that renders every frame): whatever is enqueued behind an update
    sees that update — here with the jitter changing between groups and a sample after every update."""
    name = "cave_small"
    counts, side, s, origin, scene = CONFIGS[name]
    pos, nrm = shading_points(np.random.default_rng(5), counts, side, origin, 300)
    with _engine(ddgi, name) as ref:
        ref.set_tuning("frames_in_flight", 1)
        want = {}
        for seed in (1, 2):
            ref.generate_probe_rays(seed=seed, reseed=True)
            ref.probe_update()
            want[seed] = (ref.read_textures()[0], ref.sample(pos, nrm))
    with _engine(ddgi, name) as eng:
        for seed in (1, 2, 1):
            eng.generate_probe_rays(seed=seed, reseed=True)
            for k in range(4):
                eng.probe_update()
                rgb, cage = eng.sample(pos, nrm)
                assert np.array_equal(rgb.view(np.uint32), want[seed][1][0].view(np.uint32)) and np.array_equal(cage, want[seed][1][1])
                if k & 1:
                    assert np.array_equal(eng.read_textures()[0], want[seed][0])


def test_a_change_of_lights_or_settings_is_never_continued(ddgi, oracle):
    """An update with other lights / bounces / scene than its predecessor is other work: it must be traced by its own launch
    with its own arguments, however quickly it follows."""
    name = "cave_small"
    counts, side, s, origin, scene = CONFIGS[name]
    lights_a = np.array([(100.0, (1.0, 1.0, 1.0), (4.0, 17.5, 8.5))], dtype=ddgi.LIGHT_DTYPE)
    lights_b = np.array([(60.0, (1.0, 0.6, 0.2), (-3.0, 12.0, 2.5))], dtype=ddgi.LIGHT_DTYPE)
    with _engine(ddgi, name) as ref:
        ref.set_tuning("frames_in_flight", 1)
        ref.generate_probe_rays(seed=1)
        want = []
        for lights, bounces in ((lights_a, 8), (lights_b, 8), (lights_b, 3), (lights_a, 3)):
            ref.set_lights(scene, lights)
            ref.probe_update(ddgi.make_settings(scene, bounces))
            want.append(ref.read_textures()[0])
    assert not np.array_equal(want[0], want[1]) and not np.array_equal(want[1], want[2])
    with _engine(ddgi, name) as eng:
        eng.generate_probe_rays(seed=1)
        got = []
        for rep in range(2):
            for i, (lights, bounces) in enumerate(((lights_a, 8), (lights_b, 8), (lights_b, 3), (lights_a, 3))):
                eng.set_lights(scene, lights)
                eng.probe_update(ddgi.make_settings(scene, bounces))
                eng.probe_update()                           # ... and its continuation
                if rep:
                    assert np.array_equal(eng.read_textures()[0], want[i])


def test_c3_updates_are_continued_and_bit_exact(ddgi, oracle):
    """The headline configuration: back-to-back updates ARE continued (the mechanism engages: workgroups go on with the next
    update's rays) and every texel of the last update's pair equals the oracle's; the pairs held another jitter's texels before."""
    want = c3_oracle_albedo(oracle, "pinned", seed=1)
    with _engine(ddgi, "c3_cave") as eng:
        eng.generate_probe_rays(seed=7, reseed=True)
        eng.probe_update()
        eng.probe_update()
        eng.synchronize()
        before = eng.get_tuning("continued_workgroups")
        eng.generate_probe_rays(seed=1, reseed=True)
        for _ in range(6):
            eng.probe_update()
        albedo = eng.read_textures()[0]
        continued = eng.get_tuning("continued_workgroups") - before
        assert np.array_equal(albedo, want)
        assert continued >= 3 * 128, f"only {continued} workgroups went on with a later update's rays in 6 back-to-back updates"
        # the same with one update per launch
        eng.set_tuning("frames_in_flight", 1)
        before = eng.get_tuning("continued_workgroups")
        eng.probe_update()
        eng.probe_update()
        assert np.array_equal(eng.read_textures()[0], want)
        assert eng.get_tuning("continued_workgroups") == before


def test_device_textures_pins_the_pair(ddgi):
    """A host that asks for the texture addresses may keep them (in-place all-gather, interop): the handle stays on that pair."""
    with _engine(ddgi, "cave_small") as eng:
        eng.generate_probe_rays(seed=1)
        eng.probe_update()
        eng.probe_update()
        want = eng.read_textures()[0]
        p0 = eng.device_textures()["tex0"]
        for _ in range(3):
            eng.probe_update()
            assert eng.device_textures()["tex0"] == p0
        assert np.array_equal(eng.read_textures()[0], want)


def test_reserved_cus_leave_the_result_alone(ddgi, oracle):
    """Tuning "reserve_cus": the queue kernel launches fewer workgroups than there are CUs (room for an exchange's kernels on a
    sharded grid) — same texels, chains of updates included."""
    name = "cave_small"
    want = _oracle_albedo(oracle, name, 1)
    with _engine(ddgi, name) as eng:
        eng.generate_probe_rays(seed=1, reseed=True)
        for reserve in (3, 0, 250):
            eng.set_tuning("reserve_cus", reserve)
            assert eng.get_tuning("reserve_cus") == reserve
            for _ in range(5):
                eng.probe_update()
            albedo, distance = eng.read_textures()
            assert np.array_equal(albedo, want) and not distance.any(), reserve
