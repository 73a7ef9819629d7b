"""GPU parity tests proper: the HIP path (through the C ABI) against the CPU oracle.

Tolerances
  * PINNED oracle arithmetic (the arithmetic the kernels implement, DESIGN.md "Arithmetic
    pinning"): rgba8 texels BIT-EXACT, cage indices bit-exact, sampled rgb bit-exact (float32).
  * LITERAL oracle arithmetic (every GLSL operator one IEEE op, libm sin/cos): the reference
    itself is only defined up to its driver's precision, so the stated tolerance is
    |texel difference| <= 1/255 on >= 99.9 % of texel channels and a mean absolute difference
    below 0.05/255 (Cornell).  For the cave the same bound is checked at >= 99 % because the
    noise hashes amplify last-bit differences of sin.
"""
import numpy as np
import pytest

from tests.common import CONFIGS, c3_oracle_albedo, shading_points, texel_tolerance_stats

pytestmark = pytest.mark.gpu


def _engine(ddgi, name, max_bounces=8, **kw):
    counts, side, s, origin, scene = CONFIGS[name]
    return ddgi.ProbeEngine(ddgi.make_field(counts, side, s, origin), ddgi.make_settings(scene, max_bounces), **kw)


def _oracle_textures(oracle, name, max_bounces=8, rays=None):
    counts, side, s, origin, scene = CONFIGS[name]
    f = oracle.make_field(counts, side, s, origin)
    if rays is None:
        rays = oracle.generate_probe_rays(f, oracle.new_rand_state(1))
    return oracle.probe_update(f, oracle.make_settings(scene, max_bounces), rays), f, rays


@pytest.mark.parametrize("name", ["c1_cornell", "cave_small", "cave_odd", "house_small", "c2_cornell"])
def test_probe_update_bit_exact_vs_pinned_oracle(ddgi, oracle, name):
    with _engine(ddgi, name) as eng:
        eng.generate_probe_rays(seed=1)
        eng.probe_update()
        albedo, distance = eng.read_textures()
    (want_a, want_d), _, _ = _oracle_textures(oracle, name)
    assert albedo.shape == want_a.shape
    nbad = int((albedo != want_a).any(axis=-1).sum())
    assert nbad == 0, f"{nbad} of {albedo.shape[0] * albedo.shape[1]} texels differ from the oracle"
    assert not distance.any() and not want_d.any()     # `distances` is never assigned
    assert (albedo[..., 3] == 255).all()
    assert albedo[..., :3].any()                        # something is lit


@pytest.mark.parametrize("bounces", [1, 3])
def test_probe_update_other_bounce_counts(ddgi, oracle, bounces):
    with _engine(ddgi, "cave_small", max_bounces=bounces) as eng:
        eng.generate_probe_rays(seed=1)
        eng.probe_update()
        albedo, _ = eng.read_textures()
    (want_a, _), _, _ = _oracle_textures(oracle, "cave_small", max_bounces=bounces)
    assert np.array_equal(albedo, want_a)


@pytest.mark.parametrize("name,frac", [("c2_cornell", 0.999), ("cave_small", 0.999)])
def test_probe_update_within_tolerance_of_literal_oracle(ddgi, oracle, name, frac):
    with _engine(ddgi, name) as eng:
        eng.generate_probe_rays(seed=1)
        eng.probe_update()
        albedo, _ = eng.read_textures()
    oracle.set_arith(False)  # LITERAL: IEEE ops in GLSL source order, libm sin/cos
    (want_a, _), _, _ = _oracle_textures(oracle, name)
    oracle.set_arith(True)
    diff = np.abs(albedo[..., :3].astype(np.int32) - want_a[..., :3].astype(np.int32))
    assert (diff <= 1).mean() >= frac
    assert diff.mean() < 0.05


def test_uploaded_rays_and_readback_of_generated_rays(ddgi, oracle):
    # the boundary also takes caller-made rays (probe_buffer.copy_to, rvpt.cpp:285)
    counts, side, s, origin, scene = CONFIGS["c1_cornell"]
    rng = np.random.default_rng(3)
    f = oracle.make_field(counts, side, s, origin)
    rays = oracle.generate_probe_rays(f, oracle.new_rand_state(1))
    d = rng.normal(size=(len(rays), 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays["direction"] = d
    rays["origin"] += rng.uniform(-0.4, 0.4, size=(len(rays), 3)).astype(np.float32)
    with _engine(ddgi, "c1_cornell") as eng:
        eng.upload_probe_rays(rays)
        assert eng.get_probe_rays().tobytes() == rays.tobytes()
        eng.probe_update()
        albedo, _ = eng.read_textures()
        bad = rays.copy()
        bad["probe_info"][5, 1] = s  # tile_x out of range: must be rejected, not written
        with pytest.raises(ddgi.DDGIError):
            eng.upload_probe_rays(bad)
    (want_a, _), _, _ = _oracle_textures(oracle, "c1_cornell", rays=rays)
    assert np.array_equal(albedo, want_a)


def test_per_frame_upload_of_a_large_ray_buffer(ddgi):
    """What the reference's host does every frame (probe_buffer.copy_to of the WHOLE buffer, rvpt.cpp:285) on a grid large enough for the upload's host
    threads (C2: 131 072 rays, 6 MB — two ranges): a second and third upload go through the page-locked host copy; a bad ray in EACH range is rejected
    naming the LOWEST index, and leaves the rays uploaded before in place; textures equal those of the generated rays they are."""
    with _engine(ddgi, "c2_cornell") as eng:
        eng.generate_probe_rays(seed=5)
        rays = eng.get_probe_rays()
        eng.probe_update()
        want = eng.read_textures()[0].copy()
        other = rays.copy()
        other["direction"] = -other["direction"]
        for upload in (other, rays, rays):
            eng.upload_probe_rays(upload)
            assert eng.get_probe_rays().tobytes() == upload.tobytes()
        eng.probe_update()
        assert np.array_equal(eng.read_textures()[0], want)
        # a buffer that changed in ONE chunk only (64 Ki rays each): that run alone crosses PCIe; an unchanged buffer touches neither the GPU nor the
        # frames in flight
        part = rays.copy()
        part["direction"][100000:100010] = -part["direction"][100000:100010]
        eng.upload_probe_rays(part)
        assert eng.get_probe_rays().tobytes() == part.tobytes()
        eng.probe_update()
        changed = eng.read_textures()[0].copy()
        assert not np.array_equal(changed, want)
        for _ in range(4):
            eng.upload_probe_rays(part)          # what the reference's host does per frame: the same rays again
            eng.probe_update()
        assert np.array_equal(eng.read_textures()[0], changed)
        eng.upload_probe_rays(rays)
        eng.probe_update()
        assert np.array_equal(eng.read_textures()[0], want)
        bad = other.copy()
        bad["probe_info"][40001, 0] = -1.0           # (both ranges hold one: the lower index is the one named)
        bad["probe_info"][120000, 2] = 1e9
        with pytest.raises(ddgi.DDGIError, match="ray 40001:"):
            eng.upload_probe_rays(bad)
        bad["probe_info"][3, 0] = float("nan")
        with pytest.raises(ddgi.DDGIError, match="ray 3:"):
            eng.upload_probe_rays(bad)
        assert eng.get_probe_rays().tobytes() == rays.tobytes()
        eng.probe_update()
        assert np.array_equal(eng.read_textures()[0], want)
        # a reconfiguration moves the host copy (it is unpinned first); uploads go on
        eng.set_ray_tile(8, 8)
        eng.generate_probe_rays(seed=5)
        small = eng.get_probe_rays()
        eng.upload_probe_rays(small)
        eng.probe_update()
        eng.synchronize()


def test_generated_rays_match_oracle_and_sequence_continues(ddgi, oracle):
    counts, side, s, origin, _ = CONFIGS["c1_cornell"]
    st = oracle.new_rand_state(1)
    f = oracle.make_field(counts, side, s, origin)
    with _engine(ddgi, "c1_cornell") as eng:
        eng.generate_probe_rays(seed=1)
        assert eng.get_probe_rays().tobytes() == oracle.generate_probe_rays(f, st).tobytes()
        eng.generate_probe_rays(seed=1)  # second call continues the sequence (Q1)
        assert eng.get_probe_rays().tobytes() == oracle.generate_probe_rays(f, st).tobytes()
        eng.generate_probe_rays(seed=1, reseed=True)
        assert eng.get_probe_rays().tobytes() == oracle.generate_probe_rays(f, oracle.new_rand_state(1)).tobytes()


@pytest.mark.parametrize("name", ["c2_cornell", "cave_small", "cave_odd"])
def test_sample_bit_exact_vs_pinned_oracle(ddgi, oracle, name):
    counts, side, s, origin, scene = CONFIGS[name]
    pos, nrm = shading_points(np.random.default_rng(11), counts, side, origin, 4096)
    # a few degenerate directions: straight along +-z (acos argument is 0/0 -> NaN -> row 0, Q8)
    nrm[:4] = [[0, 0, 1], [0, 0, -1], [1, 0, 0], [0, 1, 0]]
    with _engine(ddgi, name) as eng:
        eng.generate_probe_rays(seed=1)
        eng.probe_update()
        albedo, distance = eng.read_textures()
        rgb, cage = eng.sample(pos, nrm)
    want_rgb, want_cage = oracle.sample(oracle.make_field(counts, side, s, origin), albedo, distance, pos, nrm)
    assert np.array_equal(cage, want_cage)                       # probe-cage indices bit-exact
    assert np.array_equal(rgb.view(np.uint32), want_rgb.view(np.uint32))
    inside = (cage[:, 0] >= 0)
    assert 0.05 < inside.mean() < 1.0                            # both branches exercised
    assert np.array_equal(rgb[~inside], np.tile(np.float32([1, 0, 1]), ((~inside).sum(), 1)))


@pytest.mark.parametrize("name", ["c2_cornell", "cave_odd"])
def test_sample_paths_agree(ddgi, oracle, name):
    """A large REF batch takes sample_probe from its per-texel table (k_sample_box_filter, rebuilt after every update), a
    medium one is grouped by cage, a small one goes as it comes; "sample_box" / "sample_group" switch the first two off.  Every
    path must equal the oracle bit for bit: points everywhere, crowded into one cage, on the field's last cage layer (whose corner
    indices wrap, Q4) and outside."""
    counts, side, s, origin, scene = CONFIGS[name]
    rng = np.random.default_rng(23)
    spread, nrm = shading_points(rng, counts, side, origin, 60000)          # everywhere, some outside
    o = np.asarray(origin, dtype=np.float32)
    crowd = (rng.uniform(0.05, 0.95, size=(9000, 3)) * side + o).astype(np.float32)
    top = (rng.uniform(-0.5, 0.5, size=(6000, 3)) * np.float32(side) * np.asarray(counts, dtype=np.float32) + o).astype(np.float32)
    top[:, 0] = o[0] + side * (counts[0] // 2 - 0.5)                                             # the last cage layer in x
    pos = np.concatenate([spread, crowd, top]).astype(np.float32)
    nrm = np.concatenate([nrm, rng.normal(size=(len(pos) - len(nrm), 3)).astype(np.float32)])
    nrm[:4] = [[0, 0, 1], [0, 0, -1], [1, 0, 0], [0, 1, 0]]                                      # degenerate directions (Q8)
    f = oracle.make_field(counts, side, s, origin)
    with _engine(ddgi, name) as eng:
        for seed in (1, 2):                                                 # the table must follow the textures
            eng.generate_probe_rays(seed=seed, reseed=True)
            eng.probe_update()
            got = {}
            for box, group in ((1, 1), (0, 1), (0, 0)):
                eng.set_tuning("sample_box", box)
                eng.set_tuning("sample_group", group)
                got[(box, group)] = eng.sample(pos, nrm)
            # (the grouped path on a batch that comes in cage order: no permutation is written, k_sample_place — same results)
            cell = np.floor((pos - o) / np.float32(side)).astype(np.int64)
            order = np.lexsort((cell[:, 0], cell[:, 1], cell[:, 2]))
            eng.set_tuning("sample_box", 0)
            eng.set_tuning("sample_group", 1)
            in_order = eng.sample(pos[order], nrm[order])
            eng.set_tuning("sample_box", 1)
            small = eng.sample(pos[:777], nrm[:777])                         # a small batch reuses the table that is there
            albedo, distance = eng.read_textures()
            want_rgb, want_cage = oracle.sample(f, albedo, distance, pos, nrm)
            for key, (rgb, cage) in got.items():
                assert np.array_equal(cage, want_cage), f"seed {seed}, (sample_box, sample_group) = {key}"
                assert np.array_equal(rgb.view(np.uint32), want_rgb.view(np.uint32)), f"seed {seed}, (sample_box, sample_group) = {key}"
            assert np.array_equal(small[0].view(np.uint32), want_rgb[:777].view(np.uint32)) and np.array_equal(small[1], want_cage[:777])
            assert np.array_equal(in_order[0].view(np.uint32), want_rgb[order].view(np.uint32)) and np.array_equal(in_order[1], want_cage[order])
    inside = want_cage[:, 0] >= 0
    assert 0.02 < inside.mean() < 1.0, inside.mean()


def test_grouped_batch_beyond_2p24_points_is_sampled_in_pieces(ddgi, oracle):
    """The grouping kernels count runs of a batch in 16-bit counters (256 runs of fewer than 65 536 points): a batch of more than
    2^24 points goes through in pieces.  Every point of a 2^24 + 70 001 point batch grouped by cage must equal the same batch
    ungrouped, bit for bit, and the oracle on a spread sample — with the per-texel table off, so that the grouped path is the
    one that runs (REF), and in DDGI mode, which has no table."""
    import torch

    name = "cave_small"
    counts, side, s, origin, scene = CONFIGS[name]
    n = (1 << 24) + 70001
    g = torch.Generator(device="cuda").manual_seed(5)
    half = torch.tensor([c * side * 0.55 for c in counts], dtype=torch.float32, device="cuda")
    pos = (torch.rand((n, 3), generator=g, device="cuda") * 2 - 1) * half + torch.tensor(origin, dtype=torch.float32, device="cuda")
    nrm = torch.randn((n, 3), generator=g, device="cuda")
    out = {}
    with _engine(ddgi, name) as eng:
        eng.generate_probe_rays(seed=1)
        eng.probe_update()
        eng.set_tuning("sample_box", 0)
        for group in (1, 0):
            eng.set_tuning("sample_group", group)
            rgb = torch.full((n, 3), -1.0, dtype=torch.float32, device="cuda")
            cage = torch.full((n, 8), -7, dtype=torch.int32, device="cuda")
            torch.cuda.synchronize()        # (the engine runs on its own stream)
            eng.sample_device(pos.data_ptr(), nrm.data_ptr(), n, rgb.data_ptr(), cage.data_ptr())
            eng.synchronize()
            torch.cuda.synchronize()
            out[group] = (rgb, cage)
        assert torch.equal(out[1][1], out[0][1]) and torch.equal(out[1][0].view(torch.int32), out[0][0].view(torch.int32))
        pick = torch.cat([torch.arange(0, n, 4099, device="cuda"), torch.arange(n - 3000, n, device="cuda")])   # (incl. the last piece)
        albedo, distance = eng.read_textures()
        want_rgb, want_cage = oracle.sample(oracle.make_field(counts, side, s, origin), albedo, distance, pos[pick].cpu().numpy(), nrm[pick].cpu().numpy())
        assert np.array_equal(out[1][1][pick].cpu().numpy(), want_cage)
        assert np.array_equal(out[1][0][pick].cpu().numpy().view(np.uint32), want_rgb.view(np.uint32))
        assert 0.1 < (want_cage[:, 0] >= 0).mean() < 1.0
        # DDGI mode: no table, always grouped
        del out
        eng.set_mode(ddgi.MODE_DDGI)
        for _ in range(2):
            eng.probe_update()
        res = {}
        for group in (1, 0):
            eng.set_tuning("sample_group", group)
            rgb = torch.full((n, 3), -1.0, dtype=torch.float32, device="cuda")
            torch.cuda.synchronize()
            eng.sample_device(pos.data_ptr(), nrm.data_ptr(), n, rgb.data_ptr(), None)
            eng.synchronize()
            torch.cuda.synchronize()
            res[group] = rgb
        assert torch.equal(res[1].view(torch.int32), res[0].view(torch.int32))


def test_sharded_slabs_equal_the_full_grid(ddgi, oracle):
    """z-slab sharding: rank r of 2 fills exactly its slab of the slab-major
    texture and nothing else; the union equals the unsharded result."""
    name = "cave_small"
    counts, side, s, origin, scene = CONFIGS[name]
    (want_a, _), _, _ = _oracle_textures(oracle, name)
    parts = []
    for rank in range(2):
        with _engine(ddgi, name, rank=rank, world=2) as eng:
            eng.generate_probe_rays(seed=1)
            eng.probe_update()
            a, _ = eng.read_textures()
            info = eng.device_textures()
            assert info["slab_bytes0"] * 2 == info["tex0_bytes"] and info["slab_offset0"] == rank * info["slab_bytes0"]
            parts.append(a)
    # rank's columns in the raster: tile column = z*cx + x  (probe_pass.comp:139-145)
    cx, cz = counts[0], counts[2]
    colmask = np.zeros(want_a.shape[1], dtype=bool)
    colmask[: (cz // 2) * cx * s] = True
    assert np.array_equal(parts[0][:, colmask], want_a[:, colmask]) and not parts[0][:, ~colmask].any()
    assert np.array_equal(parts[1][:, ~colmask], want_a[:, ~colmask]) and not parts[1][:, colmask].any()


def test_full_size_c3_full_grid_vs_oracle(ddgi, oracle):
    """BASELINE config C3 (32x16x32 probes x 256 rays, cave): size-independent properties and a byte
    comparison of EVERY texel of the grid with the oracle (4 194 304 rays; seconds on the GPU box's host
    cores, see BENCH_r01.json's cpu_baseline)."""
    counts, side, s, origin, scene = CONFIGS["c3_cave"]
    with _engine(ddgi, "c3_cave") as eng:
        eng.generate_probe_rays(seed=1)
        eng.probe_update()
        a1, d1 = eng.read_textures()
        eng.probe_update()
        a2, _ = eng.read_textures()
        ms = eng.last_update_ms()
    assert np.array_equal(a1, a2)                 # Q18: no frame term in the RNG -> identical frames
    assert not d1.any() and (a1[..., 3] == 255).all()
    assert a1.shape == (16 * 16, 32 * 32 * 16, 4)
    assert ms["trace_ms"] > 0
    # every probe's tile is written: rays that see light exist all over the cave; rock-bound probes
    # get the 0.2*base*lambert term, so fully black tiles are rare
    tiles = a1[..., :3].reshape(16, 16, 1024, 16, 3).any(axis=(1, 3, 4))
    assert tiles.mean() > 0.9
    want = c3_oracle_albedo(oracle, "pinned")
    nbad = int((a1 != want).any(axis=-1).sum())
    assert nbad == 0, f"{nbad} of {a1.shape[0] * a1.shape[1]} texels of the full C3 grid differ from the oracle"
    # ... and the stated tolerance against the LITERAL arithmetic (IEEE operations in GLSL source order, libm sin / cos / acos),
    # the nearest thing to the reference's own semantics, on the headline configuration: |d| <= 1/255 on >= 99.9 % of the
    # texel channels, mean |d| < 0.05/255
    within, mean, differing = texel_tolerance_stats(a1, c3_oracle_albedo(oracle, "literal"))
    assert within >= 0.999 and mean < 0.05, f"HIP vs LITERAL oracle on C3: {within * 100:.4f} % within 1/255, mean {mean:.5f}/255, {differing} texels differ"


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [2, 3])
def test_full_size_c3_full_grid_other_ray_sets(ddgi, oracle, seed):
    """The full C3 grid again with differently seeded probe rays: every texel against the oracle.  The work the kernels skip
    (feelers decided by k_light_visibility's classes and lists, dead feelers) rests on arguments about the reference's float
    march; a hole in one shows as a handful of texels in 4 million (a SHADOW rule that overlooked grid_march stepping over a
    voxel whose entry plane it lands on exactly differed in ONE texel), so more than one ray set is compared."""
    with _engine(ddgi, "c3_cave") as eng:
        eng.generate_probe_rays(seed=seed)
        eng.probe_update()
        a1, _ = eng.read_textures()
    # (every texel of the grid is compared for seed 1 — test_full_size_c3_full_grid_vs_oracle; of these ray sets a quarter of the
    # probes, spread over the grid: the oracle's raster is what the GPU suite's time goes into)
    want = c3_oracle_albedo(oracle, "pinned", seed=seed, fraction=4)
    mask = want[..., 3] == 255
    assert mask.sum() >= a1.shape[0] * a1.shape[1] // 5
    nbad = int((a1[mask] != want[mask]).any(axis=-1).sum())
    assert nbad == 0, f"seed {seed}: {nbad} of {int(mask.sum())} compared texels of the C3 grid differ from the oracle"


def test_torch_owned_textures_and_stream(ddgi, oracle):
    """The multi-GPU plumbing at world size 1: textures allocated by torch and bound into the
    engine (ddgi_bind_textures), kernels on torch's current stream (ddgi_set_stream), the in-place
    all-gather a no-op.  Same bytes as the engine-owned path."""
    import torch

    from ddgi_amd import distributed as dd

    name = "cave_small"
    counts, side, s, origin, scene = CONFIGS[name]
    (want_a, _), _, _ = _oracle_textures(oracle, name)
    with _engine(ddgi, name) as eng:
        stream = torch.cuda.Stream()
        with torch.cuda.stream(stream):
            eng.set_stream(stream.cuda_stream)
            tex = dd.ShardedTextures(eng, torch.device("cuda", 0))
            eng.generate_probe_rays(seed=1)
            eng.probe_update()
            tex.all_gather()
            stream.synchronize()
            slab = tex.tex0.cpu().numpy().view(np.uint8)
            raster = dd.slab_major_to_raster(slab, counts, s, (4,))
            assert np.array_equal(raster, want_a)
            albedo, _ = eng.read_textures()          # the C ABI reads the bound buffers too
            assert np.array_equal(albedo, want_a)
            pos, nrm = shading_points(np.random.default_rng(2), counts, side, origin, 512)
            d_pos = torch.from_numpy(pos).cuda()
            d_nrm = torch.from_numpy(nrm).cuda()
            d_rgb = torch.empty((512, 3), dtype=torch.float32, device="cuda")
            d_cage = torch.empty((512, 8), dtype=torch.int32, device="cuda")
            eng.sample_device(d_pos.data_ptr(), d_nrm.data_ptr(), 512, d_rgb.data_ptr(), d_cage.data_ptr())
            stream.synchronize()
            want_rgb, want_cage = oracle.sample(oracle.make_field(counts, side, s, origin), want_a, np.zeros_like(want_a), pos, nrm)
            assert np.array_equal(d_cage.cpu().numpy(), want_cage)
            assert np.array_equal(d_rgb.cpu().numpy().view(np.uint32), want_rgb.view(np.uint32))
            tex.close()


@pytest.mark.gpu
def test_pipelined_exchange_on_a_one_rank_rccl_group(ddgi):
    """The pipelined (double-buffered, side-stream) exchange with a real RCCL communicator of one
    rank: bind alternation, stream/event ordering and the collective on the communication stream.
    Update k must end up in pair k&1, identical to what a plain engine produces for the same rays."""
    import socket

    import torch
    import torch.distributed as dist

    from ddgi_amd import distributed as dd

    name = "c1_cornell"
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        want = []
        with _engine(ddgi, name) as ref:
            for seed in (1, 2, 3):
                ref.generate_probe_rays(seed=seed)
                ref.probe_update()
                want.append(ref.read_textures()[0])
        counts, side, s, origin, scene = CONFIGS[name]
        with _engine(ddgi, name) as eng:
            eng.set_stream(torch.cuda.current_stream().cuda_stream)
            tex = dd.ShardedTextures(eng, torch.device("cuda", 0), pipelined=True)
            for seed in (1, 2, 3):
                eng.generate_probe_rays(seed=seed)
                tex.begin_step()
                eng.probe_update()
                tex.all_gather()
            tex.finish()
            torch.cuda.synchronize()
            last = dd.slab_major_to_raster(tex.latest()[0].cpu().numpy(), counts, s, (4,))
            prev = dd.slab_major_to_raster(tex.bufs[1 - tex.cur][0].cpu().numpy(), counts, s, (4,))
            assert np.array_equal(last, want[2]) and np.array_equal(prev, want[1])
            assert not np.array_equal(want[1], want[2])
            tex.close()
    finally:
        dist.destroy_process_group()


def test_sample_device_cage_buffer_need_not_be_16_byte_aligned(ddgi, oracle):
    """The 8 cage indices of a point are written with two 16-byte stores when the caller's buffer allows it, with eight 4-byte
    stores when it does not (a sub-allocated int32 buffer, include/ddgi_probe.h: ddgi_sample_device) — same values, REF and DDGI."""
    import torch

    name = "cave_small"
    counts, side, s, origin, scene = CONFIGS[name]
    pos_h, nrm_h = shading_points(np.random.default_rng(3), counts, side, origin, 5000)
    pos, nrm = torch.from_numpy(pos_h).cuda(), torch.from_numpy(nrm_h).cuda()
    n = pos.shape[0]
    with _engine(ddgi, name) as eng:
        eng.generate_probe_rays(seed=1)
        eng.probe_update()
        for mode in ("ref", "ddgi"):
            if mode == "ddgi":
                eng.set_mode(ddgi.MODE_DDGI)
                eng.probe_update()
            got = []
            for off in (0, 1, 3):   # int32 elements: 0, 4 and 12 bytes off a 16-byte boundary
                rgb = torch.empty((n, 3), dtype=torch.float32, device="cuda")
                flat = torch.full((n * 8 + 8,), -7, dtype=torch.int32, device="cuda")
                torch.cuda.synchronize()
                eng.sample_device(pos.data_ptr(), nrm.data_ptr(), n, rgb.data_ptr(), flat.data_ptr() + 4 * off)
                eng.synchronize()
                torch.cuda.synchronize()
                assert (flat[:off] == -7).all() and (flat[off + n * 8:] == -7).all()
                got.append((flat[off:off + n * 8].clone(), rgb))
            for cage, rgb in got[1:]:
                assert torch.equal(cage, got[0][0]) and torch.equal(rgb.view(torch.int32), got[0][1].view(torch.int32)), mode
            assert (got[0][0] != -7).all()
