"""z-slab sharding of the probe grid across the GPUs of one node.

New design — the reference is single-GPU and has no collective.  Every probe ray is independent;
the only cross-probe dependency is the cage sample, whose 8 probes can straddle a slab boundary.
So: rank r of G owns the probes with z in [r*cz/G, (r+1)*cz/G), traces (and in DDGI mode blends)
only those, and ONE all-gather per texture then gives every rank the whole field before sampling.

The device textures are slab-major ([z][y][x][texel...]) precisely so that a rank's contribution
is one contiguous, equal-sized chunk: the all-gather runs in place on the full buffer
(all_gather_into_tensor(full, full[rank*chunk:(rank+1)*chunk])), no packing kernels.  The
reference's raster layout (tile column = z*cx + x) would scatter a slab over cy strided bands.

PyTorch is plumbing here: it owns the buffers and the RCCL communicator (backend "nccl" on ROCm);
the kernels write straight into those buffers through ddgi_bind_textures.
"""
import numpy as np


def slab_range(cz, rank, world):
    """Probe z-layers owned by `rank`."""
    if cz % world:
        raise ValueError(f"probe_count.z = {cz} is not divisible by world = {world}")
    per = cz // world
    return rank * per, (rank + 1) * per


def slab_bytes(full_bytes, world):
    if full_bytes % world:
        raise ValueError("texture size not divisible by world")
    return full_bytes // world


def all_gather_slabs(full, rank, world, group=None):
    """In-place all-gather of a slab-major 1-D tensor `full`: every rank contributes the chunk
    [rank*n/world, (rank+1)*n/world).  Works with RCCL (GPU) and gloo (CPU tests)."""
    import torch.distributed as dist

    n = full.numel()
    per = slab_bytes(n, world)
    mine = full[rank * per:(rank + 1) * per]
    try:
        dist.all_gather_into_tensor(full, mine, group=group)
    except (RuntimeError, NotImplementedError):
        # backends without the flat primitive: gather into views of the same buffer
        chunks = [full[r * per:(r + 1) * per] for r in range(world)]
        dist.all_gather(chunks, mine.clone(), group=group)
    return full


def slab_major_to_raster(slab, counts, s, texel_shape=()):
    """[cz][cy][cx][s][s]+texel -> the reference raster [cy*s][cx*cz*s]+texel
    (tile of probe p at ((p mod cx*cz)*s, (p div cx*cz)*s), probe_pass.comp:139-145)."""
    cx, cy, cz = counts
    a = np.asarray(slab).reshape((cz, cy, cx, s, s) + tuple(texel_shape))
    # raster[y*s + ty][(z*cx + x)*s + tx]
    a = np.moveaxis(a, (0, 1, 2, 3, 4), (2, 0, 3, 1, 4))  # -> [cy][ty][cz][cx][tx]
    return a.reshape((cy * s, cz * cx * s) + tuple(texel_shape))


def raster_to_slab_major(raster, counts, s, texel_shape=()):
    cx, cy, cz = counts
    a = np.asarray(raster).reshape((cy, s, cz, cx, s) + tuple(texel_shape))
    a = np.moveaxis(a, (0, 1, 2, 3, 4), (1, 3, 0, 2, 4))  # -> [cz][cy][cx][ty][tx]
    return np.ascontiguousarray(a)


class ShardedTextures:
    """Torch-owned full-grid texture buffers bound into a sharded ProbeEngine + their all-gather."""

    def __init__(self, engine, device, group=None):
        import torch

        self.engine = engine
        self.group = group
        info = engine.device_textures()
        self.tex0 = torch.zeros(info["tex0_bytes"], dtype=torch.uint8, device=device)
        self.tex1 = torch.zeros(info["tex1_bytes"], dtype=torch.uint8, device=device)
        engine.bind_textures(self.tex0.data_ptr(), self.tex1.data_ptr())

    def all_gather(self):
        """One collective per texture, on the current torch stream (= the engine's stream)."""
        if self.engine.world == 1:
            return
        all_gather_slabs(self.tex0, self.engine.rank, self.engine.world, self.group)
        all_gather_slabs(self.tex1, self.engine.rank, self.engine.world, self.group)

    def close(self):
        self.engine.bind_textures(None, None)
