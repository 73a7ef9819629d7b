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
    sx, sy = (s, s) if np.isscalar(s) else s  # ray tile (tile_x, tile_y), ddgi_set_ray_tile
    a = np.asarray(slab).reshape((cz, cy, cx, sy, sx) + tuple(texel_shape))
    # raster[y*sy + ty][(z*cx + x)*sx + tx]
    a = np.moveaxis(a, (0, 1, 2, 3, 4), (2, 0, 3, 1, 4))  # -> [cy][ty][cz][cx][tx]
    return a.reshape((cy * sy, cz * cx * sx) + tuple(texel_shape))


def raster_to_slab_major(raster, counts, s, texel_shape=()):
    cx, cy, cz = counts
    sx, sy = (s, s) if np.isscalar(s) else s
    a = np.asarray(raster).reshape((cy, sy, cz, cx, sx) + tuple(texel_shape))
    a = np.moveaxis(a, (0, 1, 2, 3, 4), (1, 3, 0, 2, 4))  # -> [cz][cy][cx][ty][tx]
    return np.ascontiguousarray(a)


class ShardedTextures:
    """Torch-owned full-grid texture buffers bound into a sharded ProbeEngine + their all-gather.

    Usage per update:   tex.begin_step(); engine.probe_update(); tex.all_gather()
    and once before the textures are consumed / timed:   tex.finish()
    Construct it (and call its methods) under the torch stream the updates should run on: the
    constructor points the engine at torch's current stream (ddgi_set_stream).

    pipelined=False: one buffer pair; the collectives run in order with the kernels on the current
    stream (update k+1 starts after update k's exchange).
    pipelined=True (REF mode: an update rewrites the rank's whole slab and reads nothing back): two
    buffer pairs used alternately and a communication stream — update k+1 writes pair (k+1)&1 while
    pair k&1 is still being exchanged, so the exchange hides behind the next update's kernel (the
    probe rays of different updates are independent; only the consumer needs the gathered field).
    `latest()` is the pair holding the most recent update; it is complete after `finish()` (or, on the
    consumer's stream, after waiting for `ready_event()`).

    `skip_constant` (default on): in REF mode the reference never assigns its `distances` image
    (probe_pass.comp:276,302 — every rank writes zeros into its slab of zero-initialised buffers), so
    exchanging it moves no information; it is left out.
    """

    def __init__(self, engine, device, group=None, pipelined=False, ddgi_mode=False, skip_constant=True):
        import torch

        if pipelined and ddgi_mode:
            raise ValueError("pipelined exchange needs an update that does not read its own previous output: REF mode only")
        self.engine = engine
        self.group = group
        self.device = torch.device(device)
        self.cuda = self.device.type == "cuda"
        self.pipelined = bool(pipelined)
        self.gather_tex1 = bool(ddgi_mode) or not skip_constant
        info = engine.device_textures()
        nbuf = 2 if self.pipelined else 1
        self.bufs = [(torch.zeros(info["tex0_bytes"], dtype=torch.uint8, device=self.device),
                      torch.zeros(info["tex1_bytes"], dtype=torch.uint8, device=self.device)) for _ in range(nbuf)]
        self.comm = torch.cuda.Stream(self.device) if (self.pipelined and self.cuda) else None
        self.sent = [None] * nbuf   # event on the comm stream: buffer i's last exchange is over
        self.k = 0                  # updates exchanged so far
        self.cur = 0
        self.tex0, self.tex1 = self.bufs[0]
        if self.cuda:
            # the collectives, the `written` event and the waits below all live on torch's current stream:
            # the engine must launch on that stream too (its own stream is a private non-blocking one),
            # or an exchange could start before the kernels that fill the slab have finished
            engine.set_stream(torch.cuda.current_stream(self.device).cuda_stream)
        engine.bind_textures(self.tex0.data_ptr(), self.tex1.data_ptr())

    def begin_step(self):
        """Selects (and binds) the buffer pair the next probe_update writes."""
        if not self.pipelined:
            return
        import torch

        self.cur = self.k % 2
        if self.cuda and self.sent[self.cur] is not None:
            torch.cuda.current_stream(self.device).wait_event(self.sent[self.cur])  # its previous exchange has left the buffer
        self.tex0, self.tex1 = self.bufs[self.cur]
        self.engine.bind_textures(self.tex0.data_ptr(), self.tex1.data_ptr())

    def _exchange(self, pair):
        all_gather_slabs(pair[0], self.engine.rank, self.engine.world, self.group)
        if self.gather_tex1:
            all_gather_slabs(pair[1], self.engine.rank, self.engine.world, self.group)

    def all_gather(self):
        """One collective per texture that carries information; on the current stream, or (pipelined)
        on the communication stream once the update just issued on the current stream is done."""
        import torch.distributed as dist

        if self.engine.world == 1 and not (dist.is_available() and dist.is_initialized()):
            self.k += 1  # single process without a process group: nothing to exchange
            return
        pair = self.bufs[self.cur]
        if self.comm is None:
            self._exchange(pair)
        else:
            import torch

            written = torch.cuda.Event()
            written.record(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(self.comm):
                self.comm.wait_event(written)
                self._exchange(pair)
                ev = torch.cuda.Event()
                ev.record(self.comm)
                self.sent[self.cur] = ev
        self.k += 1

    def latest(self):
        """(tex0, tex1) of the most recent update (complete after finish())."""
        return self.bufs[self.cur]

    def ready_event(self):
        """Pipelined + CUDA: the event after which latest() is complete; else None."""
        return self.sent[self.cur] if self.comm is not None else None

    def finish(self):
        """Makes the current stream wait for every exchange issued so far."""
        if self.comm is not None:
            import torch

            torch.cuda.current_stream(self.device).wait_stream(self.comm)

    def close(self):
        self.finish()
        self.engine.bind_textures(None, None)
