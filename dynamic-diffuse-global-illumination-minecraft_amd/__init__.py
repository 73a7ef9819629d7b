"""MI355X-native DDGI probe-update engine — host-side Python mirror of the C ABI.

The directory name (mandated by the project layout) contains hyphens, so import it either through
the `ddgi_amd` shim at the repo root or with
`importlib.import_module("dynamic-diffuse-global-illumination-minecraft_amd")`.

The product is libddgi_probe.so (hand-written HIP kernels behind include/ddgi_probe.h); this
package only binds it (ctypes) and adds the torch.distributed plumbing for the z-slab sharded
multi-GPU path.  It never imports anything from oracle/.
"""
from .build import build_library, build_profiling_library, library_path, profiling_library_path  # noqa: F401
from .probe_engine import (  # noqa: F401
    Camera,
    DDGIError,
    IrradianceField,
    Light,
    ProbeEngine,
    RenderSettings,
    PROBE_RAY_DTYPE,
    LIGHT_DTYPE,
    MODE_REF,
    MODE_DDGI,
    ERR_TIMEOUT,
    P2P_ADDRESS_BYTES,
    comm_create,
    comm_destroy,
    comm_unique_id,
    generate_probe_rays_host,
    load_library,
    make_camera,
    make_field,
    make_settings,
    probe_tile_origin,
    read_scene_file,
    scene_block_at,
    scene_skip_field,
    scene_save,
    texture_size,
)
