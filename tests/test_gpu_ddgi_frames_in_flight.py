"""Frames in flight for updates whose inputs CHANGE from one to the next (round 5): DDGI mode — per-frame ray rotation and RNG key,
lights animated by RenderSettings::time (the reference's dormant update_lights, assets/shaders/probe_pass.comp:217-251, under a
host that adds 2 to `time` per frame, src/rvpt/rvpt.cpp:281, with MAX_FRAMES_IN_FLIGHT = 2, src/rvpt/rvpt.h:23).

What differs between the updates a launch works on travels in a per-update record (csrc/ddgi_types.h: UpdK) that the queue kernel
reads per update; the ray records of the updates of a group go to a ring of buffers (the blend of update k reads buffer k while a
launch already traces k + 1); the light-feeler classes of the updates a launch may go on with are made when the chain starts, for
the light positions the host's time steps predict (csrc/ddgi_engine.cpp: assign_vis) — an update whose lights are elsewhere
starts a chain of its own.

Results must not change by a bit: against the oracle's frame-by-frame restatement, and against the same frames traced one
update per launch."""
import ctypes as C

import numpy as np
import pytest

from tests.common import CONFIGS
from tests.test_gpu_ddgi_mode import FOUR_LIGHTS_CAVE

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def _run(ddgi, name, times, fif, lights=None, relight=None, counts_side=None):
    """The frames `times` submitted back to back (nothing between the updates: a host that runs ahead); tiles after the last."""
    counts, side, s, origin, scene = CONFIGS[name] if counts_side is None else counts_side
    with ddgi.ProbeEngine(ddgi.make_field(counts, side, s, origin), ddgi.make_settings(scene, 8)) as eng:
        eng.set_mode(ddgi.MODE_DDGI)
        eng.set_tuning("frames_in_flight", fif)
        if lights is not None:
            eng.set_lights(scene, np.array(lights, dtype=ddgi.LIGHT_DTYPE))
        before = eng.get_tuning("continued_workgroups")
        for k, t in enumerate(times):
            if relight and k in relight:
                eng.set_lights(scene, np.array(relight[k], dtype=ddgi.LIGHT_DTYPE))
            eng.probe_update(ddgi.make_settings(scene, 8, time=t))
        irr, dep = eng.read_tiles()
        return irr, dep, eng.get_tuning("continued_workgroups") - before


def _oracle_frames(oracle, name, times, lights=None, relight=None):
    counts, side, s, origin, scene = CONFIGS[name]
    f = oracle.make_field(counts, side, s, origin)
    irr, dep = oracle.new_tiles(f)
    larr = None if lights is None else np.array(lights, dtype=oracle.LIGHT_DTYPE)
    for k, t in enumerate(times):
        if relight and k in relight:
            larr = np.array(relight[k], dtype=oracle.LIGHT_DTYPE)
        oracle.ddgi_update(f, oracle.make_settings(scene, 8, time=t), k, irr, dep, lights=larr)
    return irr, dep


STEADY = [2.0 * (k + 1) for k in range(7)]            # RVPT::update: time += 2 per frame
RAGGED = [2.0, 4.0, 7.0, 7.5, 7.5, 20.0, 22.0, 24.0, 26.0]  # steps the prediction gets wrong, a frame that does not move, steady again


@pytest.mark.parametrize("fif", [1, 2, 8])
@pytest.mark.parametrize("name,lights,times", [("c1_cornell", None, STEADY), ("cave_small", None, STEADY), ("cave_small", FOUR_LIGHTS_CAVE, STEADY),
                                               ("cave_small", None, RAGGED), ("cave_small", FOUR_LIGHTS_CAVE, RAGGED[:6])])
def test_back_to_back_ddgi_frames_bit_exact_vs_oracle(ddgi, oracle, name, lights, times, fif):
    want_irr, want_dep = _oracle_frames(oracle, name, times, lights)
    irr, dep, _ = _run(ddgi, name, times, fif, lights)
    assert np.array_equal(_bits(irr), _bits(want_irr)), "irradiance tiles differ"
    assert np.array_equal(_bits(dep), _bits(want_dep)), "depth tiles differ"


def test_light_table_replaced_inside_a_group(ddgi, oracle):
    """ddgi_set_lights between two updates of a group: the next update's lights are nowhere near a prediction — it must be traced
    with ITS lights (and its feeler classes), the ones after it with theirs."""
    other = [(35.0, (1.0, 0.7, 0.4), (-6.0, 9.0, 3.0))]
    relight = {3: other, 5: FOUR_LIGHTS_CAVE[:1]}
    want_irr, want_dep = _oracle_frames(oracle, "cave_small", STEADY, None, relight)
    for fif in (1, 8):
        irr, dep, _ = _run(ddgi, "cave_small", STEADY, fif, None, relight)
        assert np.array_equal(_bits(irr), _bits(want_irr)) and np.array_equal(_bits(dep), _bits(want_dep)), fif


def test_c3_ddgi_updates_are_continued_and_bit_exact(ddgi, oracle):
    """The headline grid in DDGI mode, lights moving with every frame: back-to-back updates ARE continued (workgroups go on with the
    next update's rays), the tiles equal those of one update per launch bit for bit, and a spread sample of probes equals the
    oracle's frame-by-frame result."""
    name = "c3_cave"
    counts, side, s, origin, scene = CONFIGS[name]
    times = [2.0 * (k + 1) for k in range(10)]
    # (two groups of updates: the first teaches the handle that this host runs ahead, see ddgi_engine::runahead)
    irr8, dep8, continued = _run(ddgi, name, times, 8)
    irr1, dep1, none = _run(ddgi, name, times, 1)
    assert none == 0
    assert continued >= 3 * 128, f"only {continued} workgroups went on with a later update's rays in {len(times)} back-to-back DDGI updates"
    assert np.array_equal(_bits(irr8), _bits(irr1)) and np.array_equal(_bits(dep8), _bits(dep1))
    f = oracle.make_field(counts, side, s, origin)
    n_probes = counts[0] * counts[1] * counts[2]
    for p in np.random.default_rng(3).choice(n_probes, size=5, replace=False):
        # the oracle on a 1-probe window: the base pointers are offset so that probe p lands on these buffers
        irr_buf, dep_buf = np.zeros((8 * 8 * 4,), np.float32), np.zeros((16 * 16 * 2,), np.float32)
        for frame, t in enumerate(times):
            oracle.lib().oracle_ddgi_update(C.byref(f), C.byref(oracle.make_settings(scene, 8, time=t)), None, 0, C.c_uint32(frame),
                                            C.c_void_p(irr_buf.ctypes.data - int(p) * 8 * 8 * 4 * 4), C.c_void_p(dep_buf.ctypes.data - int(p) * 16 * 16 * 2 * 4),
                                            None, int(p), 1, 1)
        assert np.array_equal(irr8[p].reshape(-1).view(np.uint32), irr_buf.view(np.uint32)), f"probe {p}"
        assert np.array_equal(dep8[p].reshape(-1).view(np.uint32), dep_buf.view(np.uint32)), f"probe {p}"


def test_s_dyn_shape_four_lights_chained_equals_unchained(ddgi):
    """BASELINE configs[4]'s ingredients on a grid a test can afford: 4 animated lights + hysteresis, 24x12x24 probes — chained
    and one update per launch agree bit for bit over 9 frames."""
    shape = ((24, 12, 24), 2, 16, (1.4, 0.0, 1.0), 0)
    times = [2.0 * k for k in range(9)]
    a = _run(ddgi, None, times, 8, FOUR_LIGHTS_CAVE, counts_side=shape)
    b = _run(ddgi, None, times, 1, FOUR_LIGHTS_CAVE, counts_side=shape)
    assert np.array_equal(_bits(a[0]), _bits(b[0])) and np.array_equal(_bits(a[1]), _bits(b[1]))
    assert b[2] == 0
