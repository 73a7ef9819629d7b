"""world_size-2 gloo test of the N > 1 path's host logic (no GPU): z-slab partition, slab-major
layout, the in-place all-gather and the slab-major -> reference-raster conversion.  Each rank's
slab is produced here by the ORACLE standing in for the kernels (test infrastructure only)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.common import CONFIGS

NAME = "cave_small"


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ddgi_amd import distributed as dd
        from oracle import oracle_py as O

        counts, side, s, origin, scene = CONFIGS[NAME]
        cx, cy, cz = counts
        f = O.make_field(counts, side, s, origin)
        rays = O.generate_probe_rays(f, O.new_rand_state(1))
        z0, z1 = dd.slab_range(cz, rank, world)
        mine = [y * cx * cz + z * cx + x for y in range(cy) for z in range(z0, z1) for x in range(cx)]
        raster = O.probe_update_probes(f, O.make_settings(scene, 8), rays, mine, nthreads=2)
        slab = dd.raster_to_slab_major(raster, counts, s, (4,))
        # only this rank's z-layers may be non-zero before the exchange
        other = np.ones(cz, dtype=bool)
        other[z0:z1] = False
        assert not slab[other].any() and slab[z0:z1].any()
        full = torch.from_numpy(slab.reshape(-1).copy())
        dd.all_gather_slabs(full, rank, world)
        got = dd.slab_major_to_raster(full.numpy(), counts, s, (4,))
        np.save(os.path.join(out_dir, f"rank{rank}.npy"), got)
    finally:
        dist.destroy_process_group()


def test_zslab_allgather_world2_gloo(tmp_path, oracle):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    counts, side, s, origin, scene = CONFIGS[NAME]
    f = oracle.make_field(counts, side, s, origin)
    rays = oracle.generate_probe_rays(f, oracle.new_rand_state(1))
    want, _ = oracle.probe_update(f, oracle.make_settings(scene, 8), rays)
    for r in range(world):
        got = np.load(tmp_path / f"rank{r}.npy")
        assert np.array_equal(got, want), f"rank {r}: gathered field differs from the unsharded result"


def test_layout_conversions_roundtrip():
    from ddgi_amd import distributed as dd

    counts, s = (3, 2, 4), 5
    cx, cy, cz = counts
    rng = np.random.default_rng(0)
    slab = rng.integers(0, 255, size=(cz, cy, cx, s, s, 4), dtype=np.uint8)
    raster = dd.slab_major_to_raster(slab, counts, s, (4,))
    assert raster.shape == (cy * s, cx * cz * s, 4)
    for z in range(cz):
        for y in range(cy):
            for x in range(cx):
                p = y * cx * cz + z * cx + x
                tx0, ty0 = (p % (cx * cz)) * s, (p // (cx * cz)) * s  # probe_pass.comp:139-145
                assert np.array_equal(raster[ty0:ty0 + s, tx0:tx0 + s], slab[z, y, x])
    assert np.array_equal(dd.raster_to_slab_major(raster, counts, s, (4,)), slab)
    assert dd.slab_range(8, 3, 4) == (6, 8)
    with pytest.raises(ValueError):
        dd.slab_range(9, 0, 2)


class _FakeEngine:
    """Stands in for a sharded ProbeEngine: records which buffers are bound."""

    def __init__(self, rank, world, nbytes):
        self.rank, self.world, self.nbytes = rank, world, nbytes
        self.bound = None

    def device_textures(self):
        return {"tex0_bytes": self.nbytes, "tex1_bytes": self.nbytes}

    def bind_textures(self, p0, p1):
        self.bound = (p0, p1)


def _pipelined_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ddgi_amd import distributed as dd

        nbytes = 4096
        per = nbytes // world
        eng = _FakeEngine(rank, world, nbytes)
        tex = dd.ShardedTextures(eng, torch.device("cpu"), pipelined=True)
        seen = []
        for k in range(5):
            tex.begin_step()
            t0, t1 = tex.latest()
            assert eng.bound == (t0.data_ptr(), t1.data_ptr())          # the update writes the pair that was just bound
            t0[rank * per:(rank + 1) * per] = (k + 1) * 10 + rank          # "probe_update k": this rank's slab only
            tex.all_gather()
            seen.append(t0.data_ptr())
        tex.finish()
        assert seen[0] == seen[2] == seen[4] and seen[1] == seen[3] and seen[0] != seen[1]   # two pairs, alternating
        t0, t1 = tex.latest()
        np.save(os.path.join(out_dir, f"p{rank}_last.npy"), t0.numpy())
        np.save(os.path.join(out_dir, f"p{rank}_prev.npy"), tex.bufs[1 - tex.cur][0].numpy())
        assert not t1.any()                                              # the constant image is not exchanged (and stays zero)
        tex.close()
    finally:
        dist.destroy_process_group()


def test_pipelined_exchange_world2_gloo(tmp_path):
    """Double-buffered exchange (REF mode): update k is gathered from pair k&1 while k+1 writes the other."""
    world = 2
    mp.spawn(_pipelined_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    per = 4096 // world
    for r in range(world):
        last = np.load(tmp_path / f"p{r}_last.npy")
        prev = np.load(tmp_path / f"p{r}_prev.npy")
        for q in range(world):
            assert (last[q * per:(q + 1) * per] == 50 + q).all()     # update 4 (k+1 = 5) of every rank
            assert (prev[q * per:(q + 1) * per] == 40 + q).all()     # update 3 in the other pair
