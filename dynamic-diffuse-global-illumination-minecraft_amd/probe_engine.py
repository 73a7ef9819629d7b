"""ctypes binding of libddgi_probe.so (include/ddgi_probe.h).

Mirrors, for the probe path only, the public surface of the reference's `class RVPT`
(src/rvpt/rvpt.h:33-92): the `ir` (IrradianceField) and `render_settings` records,
`generate_probe_rays()`, the per-frame `update()`/`draw()` pair (here: `probe_update()`), and
`recreate_probe_textures()` (here: `configure()`).  All compute happens in the HIP library; if it
cannot be loaded, or no gfx950 GPU is present, calls raise DDGIError — there is no fallback.
"""
import ctypes as C
import os

import numpy as np

from .build import build_library, library_path

MODE_REF = 0
MODE_DDGI = 1
# ddgi_status (include/ddgi_probe.h)
ERR_INVALID_ARGUMENT, ERR_NO_DEVICE, ERR_HIP, ERR_OUT_OF_MEMORY, ERR_NOT_READY, ERR_UNSUPPORTED, ERR_TIMEOUT = -1, -2, -3, -4, -5, -6, -7
P2P_ADDRESS_BYTES = 512  # DDGI_P2P_ADDRESS_BYTES


class DDGIError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"ddgi error {code}: {message}")
        self.code = code


class IrradianceField(C.Structure):
    """RVPT::IrradianceField (src/rvpt/rvpt.h:82-90), 48 bytes, std140."""
    _fields_ = [
        ("probe_count", C.c_int32 * 3),
        ("side_length", C.c_int32),
        ("hysteresis", C.c_float),
        ("sqrt_rays_per_probe", C.c_int32),
        ("_pad0", C.c_int32 * 2),
        ("field_origin", C.c_float * 3),
        ("visualize", C.c_uint8),
        ("_pad1", C.c_uint8 * 3),
    ]


class RenderSettings(C.Structure):
    """RVPT::RenderSettings (src/rvpt/rvpt.h:70-80), 32 bytes."""
    _fields_ = [
        ("screen_width", C.c_int32),
        ("screen_height", C.c_int32),
        ("max_bounces", C.c_int32),
        ("camera_mode", C.c_int32),
        ("render_mode", C.c_int32),
        ("scene", C.c_int32),
        ("time", C.c_float),
        ("visualize_probes", C.c_int32),
    ]


class Camera(C.Structure):
    """Camera UBO of the render pass (compute_pass.comp:30-35): column-major mat4 + params."""
    _fields_ = [("matrix", C.c_float * 16), ("params", C.c_float * 4)]


def make_camera(origin, rotation_deg=(0.0, 0.0, 0.0), fov_deg=75.0, aspect=16.0 / 9.0, scale=4.0):
    """Camera::recalculate_values + get_data (src/rvpt/camera.cpp:18-26, 76-110): translate, then
    rotate about UP (y) by rotation.x, RIGHT (x) by rotation.y, FORWARD (z) by rotation.z."""
    def rot(axis, deg):
        a = np.deg2rad(deg)
        c, s = np.cos(a), np.sin(a)
        x, y, z = axis
        return np.array([[c + x * x * (1 - c), x * y * (1 - c) - z * s, x * z * (1 - c) + y * s, 0],
                         [y * x * (1 - c) + z * s, c + y * y * (1 - c), y * z * (1 - c) - x * s, 0],
                         [z * x * (1 - c) - y * s, z * y * (1 - c) + x * s, c + z * z * (1 - c), 0],
                         [0, 0, 0, 1]], dtype=np.float64)
    m = np.eye(4)
    m[:3, 3] = origin
    m = m @ rot((0, 1, 0), rotation_deg[0]) @ rot((1, 0, 0), rotation_deg[1]) @ rot((0, 0, 1), rotation_deg[2])
    cam = Camera()
    cam.matrix[:] = [float(v) for v in m.T.reshape(-1).astype(np.float32)]  # column-major
    cam.params[:] = [float(np.float32(aspect)), float(np.float32(np.deg2rad(fov_deg))), float(scale), 0.0]
    return cam


class Light(C.Structure):
    """struct Light (assets/shaders/structs.glsl:54-59)."""
    _fields_ = [("intensity", C.c_float), ("col", C.c_float * 3), ("pos", C.c_float * 3)]


PROBE_RAY_DTYPE = np.dtype(
    [("origin", "<f4", 3), ("_p0", "<f4"), ("direction", "<f4", 3), ("_p1", "<f4"),
     ("probe_info", "<f4", 3), ("_p2", "<f4")])
LIGHT_DTYPE = np.dtype([("intensity", "<f4"), ("col", "<f4", 3), ("pos", "<f4", 3)])
assert C.sizeof(IrradianceField) == 48 and C.sizeof(RenderSettings) == 32
assert PROBE_RAY_DTYPE.itemsize == 48 and LIGHT_DTYPE.itemsize == 28 == C.sizeof(Light)


def make_field(counts=(9, 7, 9), side=11, s=20, origin=(1.4, 0.0, 1.0), hysteresis=0.9):
    """Defaults are the reference's (rvpt.h:84-88)."""
    f = IrradianceField()
    f.probe_count[:] = [int(c) for c in counts]
    f.side_length = int(side)
    f.hysteresis = float(hysteresis)
    f.sqrt_rays_per_probe = int(s)
    f.field_origin[:] = [float(o) for o in origin]
    f.visualize = 1
    return f


def make_settings(scene=0, max_bounces=8, time=0.0):
    st = RenderSettings()
    st.screen_width, st.screen_height = 1600, 900
    st.max_bounces = int(max_bounces)
    st.scene = int(scene)
    st.time = float(time)
    return st


_lib = None

_VP = C.c_void_p
_SIGNATURES = {
    "ddgi_create": (C.c_int, [_VP, _VP, C.c_int, C.POINTER(_VP)]),
    "ddgi_create_sharded": (C.c_int, [_VP, _VP, C.c_int, C.c_int, C.c_int, C.POINTER(_VP)]),
    "ddgi_destroy": (C.c_int, [_VP]),
    "ddgi_configure": (C.c_int, [_VP, _VP, _VP]),
    "ddgi_reconfigure": (C.c_int, [_VP, _VP, _VP, C.c_int]),
    "ddgi_set_mode": (C.c_int, [_VP, C.c_int]),
    "ddgi_set_lights": (C.c_int, [_VP, C.c_int, _VP, C.c_int]),
    "ddgi_set_ray_tile": (C.c_int, [_VP, C.c_int, C.c_int]),
    "ddgi_get_ray_tile": (C.c_int, [_VP, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "ddgi_get_texture_size": (C.c_int, [_VP, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "ddgi_generate_probe_rays": (C.c_int, [_VP, C.c_uint32, C.c_int]),
    "ddgi_upload_probe_rays": (C.c_int, [_VP, _VP, C.c_size_t]),
    "ddgi_get_probe_rays": (C.c_int, [_VP, _VP, C.c_size_t]),
    "ddgi_probe_update": (C.c_int, [_VP, _VP]),
    "ddgi_synchronize": (C.c_int, [_VP]),
    "ddgi_tune": (C.c_int, [_VP]),
    "ddgi_set_tuning": (C.c_int, [_VP, C.c_char_p, C.c_int]),
    "ddgi_get_tuning": (C.c_int, [_VP, C.c_char_p, C.POINTER(C.c_int)]),
    "ddgi_exchange_init": (C.c_int, [_VP, _VP, C.c_int]),
    "ddgi_exchange": (C.c_int, [_VP]),
    "ddgi_exchange_finish": (C.c_int, [_VP]),
    "ddgi_exchange_p2p_export": (C.c_int, [_VP, C.c_int, _VP]),
    "ddgi_exchange_p2p_init": (C.c_int, [_VP, _VP, C.c_int]),
    "ddgi_exchange_transport": (C.c_int, [_VP, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "ddgi_exchange_ranks": (C.c_int, [_VP, C.POINTER(C.c_int)]),
    "ddgi_exchange_group_begin": (C.c_int, []),
    "ddgi_exchange_group_end": (C.c_int, []),
    "ddgi_comm_unique_id": (C.c_int, [_VP]),
    "ddgi_comm_create": (C.c_int, [_VP, C.c_int, C.c_int, C.c_int, C.POINTER(_VP)]),
    "ddgi_comm_create_all": (C.c_int, [C.c_int, _VP, _VP]),
    "ddgi_comm_destroy": (C.c_int, [_VP]),
    "ddgi_last_update_ms": (C.c_int, [_VP, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "ddgi_update_history_ms": (C.c_int, [_VP, _VP, _VP, C.c_int, C.POINTER(C.c_int)]),
    "ddgi_trace_stats": (C.c_int, [_VP, C.c_int, _VP]),
    "ddgi_read_textures": (C.c_int, [_VP, _VP, _VP]),
    "ddgi_read_tiles": (C.c_int, [_VP, _VP, _VP]),
    "ddgi_set_frame": (C.c_int, [_VP, C.c_uint32]),
    "ddgi_sample": (C.c_int, [_VP, _VP, _VP, C.c_size_t, _VP, _VP]),
    "ddgi_render": (C.c_int, [_VP, _VP, _VP, _VP, _VP]),
    "ddgi_render_device": (C.c_int, [_VP, _VP, _VP, _VP, _VP]),
    "ddgi_set_stream": (C.c_int, [_VP, _VP]),
    "ddgi_device_textures": (C.c_int, [_VP] + [C.POINTER(_VP), C.POINTER(C.c_size_t)] * 2 + [C.POINTER(C.c_size_t)] * 4),
    "ddgi_bind_textures": (C.c_int, [_VP, _VP, _VP]),
    "ddgi_sample_device": (C.c_int, [_VP, _VP, _VP, C.c_size_t, _VP, _VP]),
    "ddgi_abi_version": (C.c_int, []),
    "ddgi_last_error": (C.c_char_p, []),
    "ddgi_texture_size": (C.c_int, [_VP, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "ddgi_probe_tile_origin": (C.c_int, [_VP, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "ddgi_generate_probe_rays_host": (C.c_int, [_VP, C.c_uint32, C.c_int, _VP, C.c_size_t]),
    "ddgi_generate_probe_rays_host_tile": (C.c_int, [_VP, C.c_int, C.c_int, C.c_uint32, C.c_int, _VP, C.c_size_t]),
    "ddgi_scene_save": (C.c_int, [C.c_int, C.c_char_p]),
    "ddgi_scene_load": (C.c_int, [_VP, C.c_char_p]),
    "ddgi_scene_set_grid": (C.c_int, [_VP, _VP, _VP, _VP]),
    "ddgi_scene_block_at": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "ddgi_scene_skip_field": (C.c_int, [C.c_int, _VP, _VP, _VP, C.c_size_t]),
    "ddgi_pinned_sinf": (C.c_float, [C.c_float]),
    "ddgi_pinned_cosf": (C.c_float, [C.c_float]),
    "ddgi_pinned_acosf": (C.c_float, [C.c_float]),
    "ddgi_pinned_sincos_small": (C.c_int, [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)


def load_library(build=True):
    """Loads (building first if stale and `build`) libddgi_probe.so; raises if it is missing."""
    global _lib
    if _lib is None:
        alt = os.environ.get("DDGI_LIB")  # profiling aid: an alternative build of the same sources (make alt) for A/B runs
        if alt:
            import torch  # noqa: F401
            lib = C.CDLL(alt)
            for name, (res, args) in _SIGNATURES.items():
                fn = getattr(lib, name)
                fn.restype = res
                fn.argtypes = args
            _lib = lib
            return _lib
        if build and os.environ.get("DDGI_NO_BUILD", "0") != "1":
            try:
                build_library()
            except Exception:
                if not os.path.exists(library_path()):
                    raise
        if not os.path.exists(library_path()):
            raise DDGIError(-2, f"{library_path()} is missing: build it with __graft_entry__.build()")
        # PyTorch wheels bundle their own libamdhip64; a process must hold ONE HIP runtime, so when
        # torch is installed let it load its runtime first and our library binds to that one.
        try:
            import torch  # noqa: F401
        except Exception:
            pass
        lib = C.CDLL(library_path())
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def _check(rc):
    if rc != 0:
        raise DDGIError(rc, load_library().ddgi_last_error().decode("utf-8", "replace"))


def texture_size(field):
    w, h = C.c_int(), C.c_int()
    _check(load_library().ddgi_texture_size(C.byref(field), C.byref(w), C.byref(h)))
    return w.value, h.value


def probe_tile_origin(field, probe_index):
    x, y = C.c_int(), C.c_int()
    _check(load_library().ddgi_probe_tile_origin(C.byref(field), probe_index, C.byref(x), C.byref(y)))
    return x.value, y.value


def generate_probe_rays_host(field, seed=1, skip_calls=0, tile=None):
    """Host-only RVPT::generate_probe_rays (rvpt.cpp:1147-1224); no GPU needed.  tile = (tile_x, tile_y)
    for a non-square ray tile (ddgi_set_ray_tile)."""
    c = field.probe_count
    tx, ty = tile if tile else (field.sqrt_rays_per_probe, field.sqrt_rays_per_probe)
    n = max(c[0] * c[1] * c[2] * tx * ty, 0)
    rays = np.zeros(min(n, 1 << 33), dtype=PROBE_RAY_DTYPE)
    _check(load_library().ddgi_generate_probe_rays_host_tile(C.byref(field), tx, ty, seed, skip_calls,
                                                             rays.ctypes.data_as(C.c_void_p), n))
    return rays


def comm_unique_id():
    """128-byte RCCL unique id (rank 0 makes it and hands it to the other ranks)."""
    buf = (C.c_uint8 * 128)()
    _check(load_library().ddgi_comm_unique_id(buf))
    return bytes(buf)


def comm_create(unique_id, world, rank, device):
    """ncclCommInitRank through the library's RCCL instance -> communicator address (int)."""
    buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
    comm = C.c_void_p()
    _check(load_library().ddgi_comm_create(buf, world, rank, device, C.byref(comm)))
    return comm.value


def comm_destroy(comm):
    _check(load_library().ddgi_comm_destroy(C.c_void_p(comm)))


def scene_save(scene, path):
    """Bake of a built-in scene -> DDGIVOX1 file (host only)."""
    _check(load_library().ddgi_scene_save(scene, os.fsencode(path)))


def read_scene_file(path):
    """(lo, dim, types[z, y, x]) of a DDGIVOX1 file."""
    raw = open(path, "rb").read()
    if raw[:8] != b"DDGIVOX1":
        raise ValueError("not a DDGIVOX1 file")
    hdr = np.frombuffer(raw, dtype="<i4", count=8, offset=8)
    lo, dim = tuple(int(v) for v in hdr[2:5]), tuple(int(v) for v in hdr[5:8])
    types = np.frombuffer(raw, dtype=np.uint8, count=dim[0] * dim[1] * dim[2], offset=8 + 32 + 4)
    return lo, dim, types.reshape(dim[2], dim[1], dim[0]).copy()


def scene_skip_field(scene):
    """(lo, codes[z, y, x]) — the fast march's skip field of a built-in scene, one byte per voxel of the bake box (host only)."""
    lo, dim = (C.c_int32 * 3)(), (C.c_int32 * 3)()
    _check(load_library().ddgi_scene_skip_field(scene, lo, dim, None, 0))
    codes = np.empty((dim[2], dim[1], dim[0]), dtype=np.uint8)
    _check(load_library().ddgi_scene_skip_field(scene, lo, dim, _ptr(codes), codes.size))
    return tuple(lo), codes


def scene_block_at(scene, x, y, z):
    return load_library().ddgi_scene_block_at(scene, x, y, z)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class ProbeEngine:
    """One GPU's probe path.  Attributes `ir` and `render_settings` mirror RVPT's public members."""

    def __init__(self, field, settings, device=0, rank=0, world=1):
        self._lib = load_library()
        self.ir = field
        self.render_settings = settings
        self.rank, self.world = rank, world
        self._h = C.c_void_p()
        _check(self._lib.ddgi_create_sharded(C.byref(field), C.byref(settings), device, rank, world,
                                             C.byref(self._h)))

    # -- lifetime ------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.ddgi_destroy(self._h)
            self._h = C.c_void_p()

    shutdown = close  # RVPT::shutdown

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- configuration -------------------------------------------------------------------------
    def configure(self, field=None, settings=None):
        """RVPT::recreate_probe_textures (rvpt.cpp:661-755)."""
        if field is not None:
            self.ir = field
        if settings is not None:
            self.render_settings = settings
        _check(self._lib.ddgi_configure(self._h, C.byref(self.ir), C.byref(self.render_settings)))

    recreate_probe_textures = configure

    def reconfigure(self, field=None, settings=None, carry_over=True):
        """configure() that keeps the tiles of every probe whose world position survives the change
        (include/ddgi_probe.h: ddgi_reconfigure)."""
        if field is not None:
            self.ir = field
        if settings is not None:
            self.render_settings = settings
        _check(self._lib.ddgi_reconfigure(self._h, C.byref(self.ir), C.byref(self.render_settings), 1 if carry_over else 0))

    def set_mode(self, mode):
        _check(self._lib.ddgi_set_mode(self._h, mode))

    def set_ray_tile(self, tile_x=0, tile_y=0):
        """Non-square ray tile: rays per probe = tile_x * tile_y (include/ddgi_probe.h: ddgi_set_ray_tile)."""
        _check(self._lib.ddgi_set_ray_tile(self._h, int(tile_x), int(tile_y)))

    @property
    def ray_tile(self):
        tx, ty = C.c_int(), C.c_int()
        _check(self._lib.ddgi_get_ray_tile(self._h, C.byref(tx), C.byref(ty)))
        return tx.value, ty.value

    @property
    def rays_per_probe(self):
        tx, ty = self.ray_tile
        return tx * ty

    def texture_size(self):
        w, h = C.c_int(), C.c_int()
        _check(self._lib.ddgi_get_texture_size(self._h, C.byref(w), C.byref(h)))
        return w.value, h.value

    def load_scene(self, path):
        """User scene (scene id 3) from a DDGIVOX1 file."""
        _check(self._lib.ddgi_scene_load(self._h, os.fsencode(path)))

    def set_scene_grid(self, lo, types_zyx):
        """User scene (scene id 3) from a uint8 array [z, y, x] of block types 0..13 at voxel-id origin lo."""
        t = np.ascontiguousarray(types_zyx, dtype=np.uint8)
        lo_a = (C.c_int32 * 3)(*[int(v) for v in lo])
        dim_a = (C.c_int32 * 3)(t.shape[2], t.shape[1], t.shape[0])
        _check(self._lib.ddgi_scene_set_grid(self._h, lo_a, dim_a, _ptr(t)))

    def set_lights(self, scene, lights):
        arr = np.ascontiguousarray(lights, dtype=LIGHT_DTYPE)
        _check(self._lib.ddgi_set_lights(self._h, scene, _ptr(arr), len(arr)))

    # -- rays ----------------------------------------------------------------------------------
    @property
    def num_probes(self):
        c = self.ir.probe_count
        return c[0] * c[1] * c[2]

    @property
    def num_rays(self):
        return self.num_probes * self.rays_per_probe

    def generate_probe_rays(self, seed=1, reseed=False):
        """RVPT::generate_probe_rays (rvpt.cpp:1177-1224) + upload (rvpt.cpp:285)."""
        _check(self._lib.ddgi_generate_probe_rays(self._h, seed, 1 if reseed else 0))

    def upload_probe_rays(self, rays):
        rays = np.ascontiguousarray(rays, dtype=PROBE_RAY_DTYPE)
        _check(self._lib.ddgi_upload_probe_rays(self._h, _ptr(rays), len(rays)))

    def get_probe_rays(self):
        rays = np.zeros(self.num_rays, dtype=PROBE_RAY_DTYPE)
        _check(self._lib.ddgi_get_probe_rays(self._h, _ptr(rays), len(rays)))
        return rays

    # -- hot path ------------------------------------------------------------------------------
    def probe_update(self, settings=None):
        """The probe half of RVPT::update + draw (rvpt.cpp:265-290, 1096-1129). Asynchronous."""
        if settings is not None:
            self.render_settings = settings
        _check(self._lib.ddgi_probe_update(self._h, C.byref(self.render_settings)))

    def synchronize(self):
        _check(self._lib.ddgi_synchronize(self._h))

    def tune(self):
        """Measure the trace kernel's wave split for the current configuration now (blocking)."""
        _check(self._lib.ddgi_tune(self._h))

    def set_tuning(self, name, value):
        _check(self._lib.ddgi_set_tuning(self._h, name.encode(), int(value)))

    def get_tuning(self, name):
        v = C.c_int()
        _check(self._lib.ddgi_get_tuning(self._h, name.encode(), C.byref(v)))
        return v.value

    # -- multi-GPU exchange (include/ddgi_probe.h: ddgi_exchange_*) ------------------------------
    def exchange_init(self, nccl_comm, pipelined=False):
        """nccl_comm: an ncclComm_t as an integer address (comm_create below), or None to detach."""
        _check(self._lib.ddgi_exchange_init(self._h, C.c_void_p(nccl_comm), 1 if pipelined else 0))

    def exchange_p2p_export(self, pipelined=False):
        """Publishes this handle's buffers for the peer-to-peer exchange: 512 opaque bytes to hand to every rank."""
        blob = (C.c_uint8 * P2P_ADDRESS_BYTES)()
        _check(self._lib.ddgi_exchange_p2p_export(self._h, 1 if pipelined else 0, blob))
        return bytes(blob)

    def exchange_p2p_init(self, addresses):
        """addresses: every rank's exchange_p2p_export() bytes, in rank order."""
        raw = b"".join(addresses)
        assert len(raw) == P2P_ADDRESS_BYTES * len(addresses)
        buf = (C.c_uint8 * len(raw)).from_buffer_copy(raw)
        _check(self._lib.ddgi_exchange_p2p_init(self._h, buf, len(addresses)))

    def exchange_transport(self):
        t, pl = C.c_int(), C.c_int()
        _check(self._lib.ddgi_exchange_transport(self._h, C.byref(t), C.byref(pl)))
        return {0: "none", 1: "rccl", 2: "p2p"}[t.value], bool(pl.value)

    def exchange_ranks(self):
        """Ranks the attached transport spans (ncclCommCount / mapped peers + 1; 0: no exchange)."""
        n = C.c_int()
        _check(self._lib.ddgi_exchange_ranks(self._h, C.byref(n)))
        return n.value

    def exchange(self):
        _check(self._lib.ddgi_exchange(self._h))

    def exchange_finish(self):
        _check(self._lib.ddgi_exchange_finish(self._h))

    def last_update_ms(self):
        t, b, tot = C.c_float(), C.c_float(), C.c_float()
        _check(self._lib.ddgi_last_update_ms(self._h, C.byref(t), C.byref(b), C.byref(tot)))
        return {"trace_ms": t.value, "blend_ms": b.value, "total_ms": tot.value}

    def update_history_ms(self, capacity=64):
        """Kernel times of the most recent updates (oldest first), from HIP events on the stream."""
        tr = np.zeros(capacity, dtype=np.float32)
        bl = np.zeros(capacity, dtype=np.float32)
        n = C.c_int()
        _check(self._lib.ddgi_update_history_ms(self._h, _ptr(tr), _ptr(bl), capacity, C.byref(n)))
        return tr[: n.value].copy(), bl[: n.value].copy()

    def trace_stats(self, enable=True):
        """Profiling aid: utilisation counters of the trace kernel since the last call."""
        out = np.zeros(64, dtype=np.uint64)
        _check(self._lib.ddgi_trace_stats(self._h, 1 if enable else 0, _ptr(out)))
        keys = ("trips", "lane_steps", "event_rounds", "lane_events", "waves", "rounds", "fetches", "_7",
                "cyc_scan", "cyc_march", "cyc_march_wait", "cyc_list", "cyc_events", "cyc_events_wait", "_14", "_15")
        st = dict(zip(keys, (int(v) for v in out[:16])))
        st["bucket_cycles"] = [int(v) for v in out[16:24]]   # event groups by bucket (ddgi_trace_wf.hip: shade_bucket)
        st["bucket_groups"] = [int(v) for v in out[24:32]]
        # light feelers of block hits: decided by the visibility table (unknown = marched / lit / shadow), dead (Lambert 0), and
        # what the marched ones found (reached the light / hit a block / neither)
        st["feeler_classes"] = dict(zip(("unknown", "table_lit", "table_shadow", "dead", "marched_lit", "marched_shadow", "marched_none", "listed"), (int(v) for v in out[56:64])))
        # queue kernel, counters build: (visits, active lanes) per section of the event code (ddgi_trace_wf.hip: LaneProbe)
        names = ("event", "albedo", "feeler set-up", "feeler sphere test", "light contribution", "bounce: accumulate + hemisphere", "primary set-up", "after albedo",
                 "inline step 1", "inline step 2", "inline step 3", "inline step 4", "write-back", "-", "-", "outside events")
        st["sections"] = {nm: (int(out[32 + 2 * s]), int(out[33 + 2 * s])) for s, nm in enumerate(names) if nm != "-" and s < 12}  # (slots 56.. hold feeler_classes; sections 12.. exist in the lap-timer build only)
        return st

    # -- outputs -------------------------------------------------------------------------------
    def read_textures(self):
        """(albedo, distance) as uint8 [H, W, 4] in the reference's raster layout."""
        w, h = self.texture_size()
        albedo = np.empty((h, w, 4), dtype=np.uint8)
        distance = np.empty((h, w, 4), dtype=np.uint8)
        _check(self._lib.ddgi_read_textures(self._h, _ptr(albedo), _ptr(distance)))
        return albedo, distance

    def read_tiles(self):
        """DDGI mode: (irradiance [P,8,8,4], depth moments [P,16,16,2]) float32, reference probe order."""
        irr = np.empty((self.num_probes, 8, 8, 4), dtype=np.float32)
        dep = np.empty((self.num_probes, 16, 16, 2), dtype=np.float32)
        _check(self._lib.ddgi_read_tiles(self._h, _ptr(irr), _ptr(dep)))
        return irr, dep

    def set_frame(self, frame):
        _check(self._lib.ddgi_set_frame(self._h, frame))

    def sample(self, pos, nrm, want_cage=True):
        """get_diffuse_gi (intersection.glsl:1306-1409) for a batch: -> (rgb [n,3], cage [n,8])."""
        pos = np.ascontiguousarray(pos, dtype=np.float32).reshape(-1, 3)
        nrm = np.ascontiguousarray(nrm, dtype=np.float32).reshape(-1, 3)
        n = pos.shape[0]
        rgb = np.empty((n, 3), dtype=np.float32)
        cage = np.empty((n, 8), dtype=np.int32) if want_cage else None
        _check(self._lib.ddgi_sample(self._h, _ptr(pos), _ptr(nrm), n, _ptr(rgb), _ptr(cage)))
        return rgb, cage

    def render(self, camera, settings, want_float=False):
        """compute_pass.comp:main for the probe-consuming integrators -> rgba8 [H, W, 4] (+ rgb f32)."""
        w, h = settings.screen_width, settings.screen_height
        img = np.empty((h, w, 4), dtype=np.uint8)
        rgb = np.empty((h, w, 3), dtype=np.float32) if want_float else None
        _check(self._lib.ddgi_render(self._h, C.byref(camera), C.byref(settings), _ptr(img), _ptr(rgb)))
        return (img, rgb) if want_float else img

    def render_device(self, camera, settings, rgba8_ptr, rgb_f32_ptr=None):
        """render() on device pointers, asynchronous on the handle's stream."""
        _check(self._lib.ddgi_render_device(self._h, C.byref(camera), C.byref(settings), C.c_void_p(rgba8_ptr), C.c_void_p(rgb_f32_ptr)))

    # -- device-pointer level ------------------------------------------------------------------
    def set_stream(self, hip_stream_ptr):
        _check(self._lib.ddgi_set_stream(self._h, C.c_void_p(hip_stream_ptr)))

    def device_textures(self):
        t0, t1 = C.c_void_p(), C.c_void_p()
        b0, b1, so0, sb0, so1, sb1 = (C.c_size_t() for _ in range(6))
        _check(self._lib.ddgi_device_textures(self._h, C.byref(t0), C.byref(b0), C.byref(t1), C.byref(b1),
                                              C.byref(so0), C.byref(sb0), C.byref(so1), C.byref(sb1)))
        return {"tex0": t0.value, "tex0_bytes": b0.value, "tex1": t1.value, "tex1_bytes": b1.value,
                "slab_offset0": so0.value, "slab_bytes0": sb0.value, "slab_offset1": so1.value,
                "slab_bytes1": sb1.value}

    def bind_textures(self, tex0_ptr, tex1_ptr):
        _check(self._lib.ddgi_bind_textures(self._h, C.c_void_p(tex0_ptr), C.c_void_p(tex1_ptr)))

    def sample_device(self, pos_ptr, nrm_ptr, n, rgb_ptr, cage_ptr=None):
        _check(self._lib.ddgi_sample_device(self._h, C.c_void_p(pos_ptr), C.c_void_p(nrm_ptr), n,
                                            C.c_void_p(rgb_ptr), C.c_void_p(cage_ptr) if cage_ptr else None))
