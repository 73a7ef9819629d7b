"""The multi-GPU exchange behind the C ABI (ddgi_exchange_*, ddgi_comm_*; csrc/ddgi_exchange.cpp) on the GPU.

The test box has one GPU, so the communicator has one rank — the whole mechanism still runs (RCCL loaded at run
time, ncclCommInitRank through the library, in-place ncclAllGather on the handle's / the communication stream,
the two alternating texture pairs, the DDGI blend reading its previous tiles from the other pair, consumers
waiting for the latest exchange).  With two or more devices visible the last test drives one handle per
device from one process through a real multi-rank communicator."""
import ctypes as C

import numpy as np
import pytest

from tests.common import CONFIGS, shading_points

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.fixture()
def comm1(ddgi):
    comm = ddgi.comm_create(ddgi.comm_unique_id(), 1, 0, 0)
    yield comm
    ddgi.comm_destroy(comm)


@pytest.mark.parametrize("pipelined", [False, True])
def test_ref_mode_exchange_matches_the_plain_engine(ddgi, comm1, pipelined):
    name = "c1_cornell"
    counts, side, s, origin, scene = CONFIGS[name]
    pos, nrm = shading_points(np.random.default_rng(5), counts, side, origin, 512)
    want = []
    with ddgi.ProbeEngine(ddgi.make_field(counts, side, s, origin), ddgi.make_settings(scene, 8)) as ref:
        for seed in (1, 2, 3):
            ref.generate_probe_rays(seed=seed, reseed=True)
            ref.probe_update()
            want.append((ref.read_textures()[0], ref.sample(pos, nrm)))
    assert not np.array_equal(want[1][0], want[2][0])
    with ddgi.ProbeEngine(ddgi.make_field(counts, side, s, origin), ddgi.make_settings(scene, 8)) as eng:
        with pytest.raises(ddgi.DDGIError):
            eng.exchange()                                   # not initialised yet
        eng.exchange_init(comm1, pipelined=pipelined)
        ptrs = []
        for k, seed in enumerate((1, 2, 3)):
            eng.generate_probe_rays(seed=seed, reseed=True)
            eng.probe_update()
            eng.exchange()
            ptrs.append(eng.device_textures()["tex0"])
            albedo, distance = eng.read_textures()           # a consumer: waits for the exchange by itself
            assert np.array_equal(albedo, want[k][0]) and not distance.any()
            rgb, cage = eng.sample(pos, nrm)
            assert np.array_equal(_bits(rgb), _bits(want[k][1][0])) and np.array_equal(cage, want[k][1][1])
        if pipelined:
            assert len(set(ptrs)) == 3 and eng.get_tuning("texture_pairs") == 2 * eng.get_tuning("frames_in_flight")   # a ring of 2 x frames_in_flight pairs, one after the other
            with pytest.raises(ddgi.DDGIError):
                eng.bind_textures(ptrs[0], ptrs[1])          # the pipelined exchange owns the pairs
        else:
            assert ptrs[0] == ptrs[1] == ptrs[2]
        eng.exchange_finish()
        eng.exchange_init(None)                              # detach: back to the handle's own pair
        eng.probe_update()
        assert np.array_equal(eng.read_textures()[0], want[2][0])
        with pytest.raises(ddgi.DDGIError):
            eng.exchange()


@pytest.mark.parametrize("pipelined", [False, True])
def test_ddgi_mode_exchange_keeps_the_temporal_blend(ddgi, oracle, comm1, pipelined):
    """DDGI mode blends into the previous tiles: with the pipelined exchange those live in the OTHER texture pair.
    Four frames must equal the oracle's (and hence the plain engine's) frame by frame."""
    counts, side, s, origin, scene = CONFIGS["cave_small"]
    f = oracle.make_field(counts, side, s, origin)
    irr, dep = oracle.new_tiles(f)
    with ddgi.ProbeEngine(ddgi.make_field(counts, side, s, origin), ddgi.make_settings(scene, 8)) as eng:
        eng.set_mode(ddgi.MODE_DDGI)
        eng.probe_update(ddgi.make_settings(scene, 8, time=2.0))      # one frame BEFORE the exchange is attached:
        oracle.ddgi_update(f, oracle.make_settings(scene, 8, time=2.0), 0, irr, dep)   # the converged field must carry over
        eng.exchange_init(comm1, pipelined=pipelined)
        for frame in range(1, 5):
            eng.probe_update(ddgi.make_settings(scene, 8, time=2.0 * (frame + 1)))
            eng.exchange()
            oracle.ddgi_update(f, oracle.make_settings(scene, 8, time=2.0 * (frame + 1)), frame, irr, dep)
            g_irr, g_dep = eng.read_tiles()
            assert np.array_equal(_bits(g_irr), _bits(irr)), f"irradiance tiles differ at frame {frame}"
            assert np.array_equal(_bits(g_dep), _bits(dep)), f"depth tiles differ at frame {frame}"


def test_grouped_exchange_records_its_end_at_the_group_end(ddgi, comm1):
    """Inside ddgi_exchange_group_begin/end RCCL only records the all-gather; it reaches the communication stream at the
    outermost group end, and only there can "this exchange is over" be recorded (an event recorded in ddgi_exchange would fire
    before the collective).  One rank is enough to run that path: new rays every frame, a consumer after every bracket."""
    counts, side, s, origin, scene = CONFIGS["c1_cornell"]
    lib = ddgi.load_library()
    with ddgi.ProbeEngine(ddgi.make_field(counts, side, s, origin), ddgi.make_settings(scene, 8)) as ref, \
            ddgi.ProbeEngine(ddgi.make_field(counts, side, s, origin), ddgi.make_settings(scene, 8)) as eng:
        eng.exchange_init(comm1, pipelined=True)
        for frame in range(5):
            for e in (ref, eng):
                e.generate_probe_rays(seed=frame + 1, reseed=True)
                e.probe_update()
            assert lib.ddgi_exchange_group_begin() == 0
            assert lib.ddgi_exchange_group_begin() == 0      # nested brackets: only the outermost end counts
            eng.exchange()
            assert lib.ddgi_exchange_group_end() == 0
            assert lib.ddgi_exchange_group_end() == 0
            if frame != 2:
                assert np.array_equal(eng.read_textures()[0], ref.read_textures()[0]), f"frame {frame}"
        eng.exchange_finish()
        eng.synchronize()


def test_a_failed_update_keeps_the_pair(ddgi, comm1):
    """ddgi_probe_update picks the texture pair it will write before it plans the launch; when the plan fails the handle must
    still point at the pair of the latest finished update (consumers and the next update depend on it)."""
    counts, side, s, origin, scene = CONFIGS["c1_cornell"]
    with ddgi.ProbeEngine(ddgi.make_field(counts, side, s, origin), ddgi.make_settings(scene, 8)) as eng:
        eng.exchange_init(comm1, pipelined=True)
        eng.generate_probe_rays(seed=1)
        eng.probe_update()
        eng.exchange()
        want = eng.read_textures()[0]
        before = eng.device_textures()["tex0"]
        with pytest.raises(ddgi.DDGIError):
            eng.probe_update(ddgi.make_settings(3, 8))       # scene 3 without a user scene loaded: NOT_READY
        assert eng.device_textures()["tex0"] == before
        assert np.array_equal(eng.read_textures()[0], want)
        eng.probe_update(ddgi.make_settings(scene, 8))       # the next update alternates as if nothing had happened
        eng.exchange()
        assert eng.device_textures()["tex0"] != before
        assert np.array_equal(eng.read_textures()[0], want)


def test_communicator_must_match_the_shard(ddgi, comm1):
    counts, side, s, origin, scene = CONFIGS["cave_small"]
    with ddgi.ProbeEngine(ddgi.make_field(counts, side, s, origin), ddgi.make_settings(scene, 8), rank=1, world=2) as eng:
        with pytest.raises(ddgi.DDGIError, match="communicator is rank 0 of 1"):
            eng.exchange_init(comm1)


def test_one_process_drives_every_visible_gpu(ddgi, oracle):
    """Two (or more) devices, one process: one sharded handle per device, a multi-rank communicator from
    ddgi_comm_create_all, grouped exchanges.  Every rank must end up with the unsharded field."""
    import torch

    ndev = torch.cuda.device_count()
    if ndev < 2:
        pytest.skip(f"NOT RUN: needs >= 2 GPUs for a real multi-rank RCCL all-gather, this box has {ndev}")
    world = 2
    counts, side, s, origin, scene = CONFIGS["cave_small"]
    lib = ddgi.load_library()
    comms = (C.c_void_p * world)()
    devs = (C.c_int * world)(*range(world))
    assert lib.ddgi_comm_create_all(world, devs, comms) == 0, lib.ddgi_last_error()
    engines = [ddgi.ProbeEngine(ddgi.make_field(counts, side, s, origin), ddgi.make_settings(scene, 8), device=r, rank=r, world=world) for r in range(world)]
    try:
        f = oracle.make_field(counts, side, s, origin)
        for r, eng in enumerate(engines):
            eng.exchange_init(comms[r], pipelined=True)
        for frame in range(4):
            # new rays every frame: a consumer that ran ahead of the all-gather would read the previous frame's texels
            for eng in engines:
                eng.generate_probe_rays(seed=frame + 1, reseed=True)
                eng.probe_update()
            assert lib.ddgi_exchange_group_begin() == 0
            for eng in engines:
                eng.exchange()
            assert lib.ddgi_exchange_group_end() == 0
            if frame in (1, 3):   # (frames 0 and 2: the next update is issued while the exchange is still in flight)
                want, _ = oracle.probe_update(f, oracle.make_settings(scene, 8), oracle.generate_probe_rays(f, oracle.new_rand_state(frame + 1)))
                for r, eng in enumerate(engines):
                    assert np.array_equal(eng.read_textures()[0], want), f"rank {r}, frame {frame}"
    finally:
        for eng in engines:
            eng.close()
        for c in comms:
            lib.ddgi_comm_destroy(c)
