// ddgi_engine.cpp — the C ABI of include/ddgi_probe.h: handle, device memory, launches.
// There is no CPU compute path in this library: every compute entry point needs a gfx950 device.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "ddgi_host.h"
#include "ddgi_pinned_math.h"
#include "ddgi_scene.h"
#include "ddgi_types.h"

namespace ddgi {
hipError_t launch_probe_trace_ref(const TraceArgs& args, int grid_blocks, hipStream_t stream);
hipError_t launch_probe_sample_ref(const SampleArgs& args, hipStream_t stream);
hipError_t trace_kernel_occupancy(int* blocks_per_cu, size_t lds_bytes);
int wf_pool_size(int nwords, bool multi_light, size_t lds_limit, int threads);
hipError_t launch_probe_trace_wf(const TraceArgs& args, int threads, int pool, int grid_blocks, uint32_t* work_counter, hipStream_t stream);
hipError_t launch_probe_blend(const BlendArgs& args, int num_cus, hipStream_t stream);
int aq_pool_size(int nwords, size_t lds_limit);
hipError_t launch_probe_trace_aq(const TraceArgs& args, int pool, int grid_blocks, int march_waves, uint32_t* work_counter, uint32_t* status, hipStream_t stream);
size_t blend_weights_floats(int n);
size_t blend_record_groups(uint32_t n_local_probes);
hipError_t launch_carry_tiles(void* dst, const void* src, const int32_t* map, uint32_t n_probes, uint32_t words_per_tile, hipStream_t stream);
hipError_t launch_probe_sample_ddgi(const SampleArgs& args, hipStream_t stream);
hipError_t launch_render_primary(const RenderArgs& args, hipStream_t stream);
}  // namespace ddgi

using namespace ddgi;

static_assert(sizeof(ddgi_irradiance_field) == 48, "IrradianceField must be 48 bytes (rvpt.h:82-90)");
static_assert(sizeof(ddgi_render_settings) == 32, "RenderSettings must be 32 bytes (rvpt.h:70-80)");
static_assert(sizeof(ddgi_probe_ray) == 48, "ProbeRay must be 48 bytes (probe.h:5-19)");
static_assert(offsetof(ddgi_irradiance_field, field_origin) == 32, "std140 layout");

// ---- error reporting ---------------------------------------------------------------------------------

static thread_local std::string g_last_error;

static int fail(int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

#define HIP_TRY(expr)                                                                                   \
    do                                                                                                  \
    {                                                                                                   \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess)                                                                           \
            return fail(e_ == hipErrorOutOfMemory ? DDGI_ERR_OUT_OF_MEMORY                              \
                        : (e_ == hipErrorNoDevice || e_ == hipErrorInvalidDevice) ? DDGI_ERR_NO_DEVICE  \
                                                                                   : DDGI_ERR_HIP,      \
                        "%s failed: %s", #expr, hipGetErrorString(e_));                                 \
    } while (0)

// ---- the handle ----------------------------------------------------------------------------------------

struct ddgi_engine
{
    int device = 0;
    int rank = 0, world = 1;
    int mode = DDGI_MODE_REF;
    ddgi_irradiance_field field{};
    ddgi_render_settings settings{};
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int num_cus = 256;

    // lights per scene
    LightK lights[4][kMaxLights];  // [3] = the user scene (DDGI_SCENE_USER)
    int n_lights[4] = {0, 0, 0, 0};
    SceneBake user_scene;          // host copy of the loaded user scene (scene id 3); empty until loaded

    // baked scene on device (per scene id, uploaded lazily)
    struct DevScene
    {
        uint32_t* bits = nullptr;
        uint8_t* types = nullptr;
        SceneK k{};
        bool ready = false;
    } dev_scene[4];

    // memoised lattice hashes on device
    float* d_noise[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    NoiseLut noise{};

    // rays
    GlibcRand rand;
    bool rand_seeded = false;
    std::vector<ddgi_probe_ray> host_rays;  // full grid (what RVPT::probe_rays holds)
    float4* d_rays = nullptr;               // local slab
    size_t d_rays_capacity = 0;             // in rays
    uint32_t n_local_rays = 0;

    // textures (REF: rgba8 texels, slab-major)
    void* own_tex[2] = {nullptr, nullptr};
    void* tex[2] = {nullptr, nullptr};
    size_t tex_bytes[2] = {0, 0};

    static constexpr int kRing = 64;  // timing history: one event triple per recent update
    hipEvent_t ev[kRing][3] = {};
    unsigned long long updates = 0;
    int wait_threshold = 64;
    uint32_t* d_work = nullptr;             // chunk counter of the wavefront trace kernel
    void* d_wf_cold = nullptr;              // wavefront kernel scratch: per-slot shading state
    float4* d_wf_dir = nullptr;
    size_t wf_cold_slots = 0, wf_dir_slots = 0;
    float* d_radiance = nullptr;            // DDGI mode ray records: rgb part then (d, d*d) part (ddgi_types.h: kRecGroup)
    size_t d_radiance_capacity = 0;         // in (record group, ray) pairs
    uint32_t frame = 0;                     // DDGI mode: updates done so far (seeds the ray rotation)
    unsigned long long aq_key = 0;  // configuration the march/event wave split was measured for
    int aq_march = 0;               // that split (0: not measured yet)
    float* d_blend_w = nullptr;  // k_blend_weights output: [256 sums][n][256]
    size_t d_blend_w_floats = 0;
    unsigned long long* d_stats = nullptr;  // profiling aid, allocated on first ddgi_trace_stats(enable)
};

static GridK make_grid(const ddgi_engine* e)
{
    GridK g;
    g.cx = e->field.probe_count[0];
    g.cy = e->field.probe_count[1];
    g.cz = e->field.probe_count[2];
    g.s = e->field.sqrt_rays_per_probe;
    g.side = e->field.side_length;
    for (int a = 0; a < 3; ++a) g.origin[a] = e->field.field_origin[a];
    g.hysteresis = e->field.hysteresis;
    g.czl = g.cz / e->world;
    g.z0 = e->rank * g.czl;
    return g;
}

static int validate_config(const ddgi_irradiance_field* f, const ddgi_render_settings* s, int world)
{
    if (!f || !s) return fail(DDGI_ERR_INVALID_ARGUMENT, "null field/settings");
    for (int a = 0; a < 3; ++a)
        if (f->probe_count[a] < 1) return fail(DDGI_ERR_INVALID_ARGUMENT, "probe_count[%d] = %d < 1", a, f->probe_count[a]);
    if (f->sqrt_rays_per_probe < 1) return fail(DDGI_ERR_INVALID_ARGUMENT, "sqrt_rays_per_probe < 1");
    if (f->side_length < 1) return fail(DDGI_ERR_INVALID_ARGUMENT, "side_length < 1");
    const unsigned long long probes = 1ull * f->probe_count[0] * f->probe_count[1] * f->probe_count[2];
    // probe_info.x carries the probe index as a float (probe.h:18): exact only below 2^24
    if (probes >= (1ull << 24)) return fail(DDGI_ERR_UNSUPPORTED, "more than 2^24 probes: probe_info.x (float) cannot index them");
    const unsigned long long rays = probes * f->sqrt_rays_per_probe * f->sqrt_rays_per_probe;
    if (rays >= (1ull << 32)) return fail(DDGI_ERR_UNSUPPORTED, "more than 2^32 probe rays: the reference's uint RNG seed wraps");
    if (s->scene < 0 || s->scene > 3) return fail(DDGI_ERR_INVALID_ARGUMENT, "scene %d not in {0,1,2,3}", s->scene);
    if (world < 1 || f->probe_count[2] % world != 0)
        return fail(DDGI_ERR_INVALID_ARGUMENT, "probe_count.z = %d is not divisible by world = %d", f->probe_count[2], world);
    return DDGI_OK;
}

static int check_kernel_status(ddgi_engine* e);

static int alloc_textures(ddgi_engine* e)
{
    for (int i = 0; i < 2; ++i)
    {
        if (e->own_tex[i]) (void)hipFree(e->own_tex[i]);
        e->own_tex[i] = nullptr;
    }
    const size_t probes = static_cast<size_t>(e->field.probe_count[0]) * e->field.probe_count[1] * e->field.probe_count[2];
    const size_t s2 = static_cast<size_t>(e->field.sqrt_rays_per_probe) * e->field.sqrt_rays_per_probe;
    if (e->mode == DDGI_MODE_DDGI)
    {
        e->tex_bytes[0] = probes * 8 * 8 * 4 * sizeof(float);    // irradiance tiles
        e->tex_bytes[1] = probes * 16 * 16 * 2 * sizeof(float);  // depth-moment tiles
    }
    else
        e->tex_bytes[0] = e->tex_bytes[1] = probes * s2 * 4;
    e->frame = 0;
    for (int i = 0; i < 2; ++i)
    {
        HIP_TRY(hipMalloc(&e->own_tex[i], e->tex_bytes[i]));
        // the reference leaves the images undefined until the first probe pass (rvpt.cpp:873-890);
        // here they start zeroed
        HIP_TRY(hipMemsetAsync(e->own_tex[i], 0, e->tex_bytes[i], e->stream));
        e->tex[i] = e->own_tex[i];
    }
    return DDGI_OK;
}

static int ensure_scene(ddgi_engine* e, int scene)
{
    ddgi_engine::DevScene& d = e->dev_scene[scene];
    if (d.ready) return DDGI_OK;
    if (scene == 3 && e->user_scene.types.empty()) return fail(DDGI_ERR_NOT_READY, "scene 3 selected but no user scene loaded (ddgi_scene_load / ddgi_scene_set_grid)");
    const SceneBake& b = scene == 3 ? e->user_scene : baked_scene(scene);
    // the kernels address the bitmap with the raw index z*nxy + y*nx + x: store it shifted so that the
    // word boundary falls on a multiple of 32 of the raw index (SceneK::bias32)
    const int bias = (b.lo[2] * b.dim[1] + b.lo[1]) * b.dim[0] + b.lo[0];
    const int bias32 = (bias >= 0 ? bias / 32 : -((-bias + 31) / 32)) * 32;
    const int shift = bias - bias32;  // 0..31
    std::vector<uint32_t> shifted((b.types.size() + shift + 31) / 32, 0u);
    for (size_t i = 0; i < b.types.size(); ++i)
        if (b.types[i]) shifted[(i + shift) >> 5] |= 1u << ((i + shift) & 31);
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d.bits), shifted.size() * sizeof(uint32_t)));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d.types), b.types.size()));
    HIP_TRY(hipMemcpy(d.bits, shifted.data(), shifted.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d.types, b.types.data(), b.types.size(), hipMemcpyHostToDevice));
    for (int a = 0; a < 3; ++a)
    {
        d.k.lo[a] = b.lo[a];
        d.k.hi[a] = b.hi[a];
    }
    d.k.nx = b.dim[0];
    d.k.nxy = b.dim[0] * b.dim[1];
    for (int a = 0; a < 3; ++a)
    {
        d.k.lo_f[a] = static_cast<float>(b.lo[a]);
        d.k.hi_f[a] = static_cast<float>(b.hi[a]);
    }
    d.k.nx_f = static_cast<float>(d.k.nx);
    d.k.nxy_f = static_cast<float>(d.k.nxy);
    d.k.bias = bias;
    d.k.bias32 = bias32;
    d.k.nwords = static_cast<int>(shifted.size());
    d.k.face_empty = b.face_empty;
    d.k.bits = d.bits;
    d.k.types = d.types;
    d.ready = true;
    return DDGI_OK;
}

static int ensure_noise(ddgi_engine* e)
{
    if (e->noise.n2 || std::getenv("DDGI_NO_NOISE_LUT")) return DDGI_OK;
    const NoiseLutHost& h = noise_lut_host();
    const std::vector<float>* src[5] = {&h.n2, &h.n1, &h.wp, &h.wall, &h.r1};
    for (int i = 0; i < 5; ++i)
    {
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&e->d_noise[i]), src[i]->size() * sizeof(float)));
        HIP_TRY(hipMemcpy(e->d_noise[i], src[i]->data(), src[i]->size() * sizeof(float), hipMemcpyHostToDevice));
    }
    e->noise.n2 = e->d_noise[0];
    e->noise.n1 = e->d_noise[1];
    e->noise.wp = e->d_noise[2];
    e->noise.wall = e->d_noise[3];
    e->noise.r1 = e->d_noise[4];
    if (const char* v = std::getenv("DDGI_LUT_OFF"))  // profiling: 1 = no wall table, 2 = no random1 table
    {
        if (std::atoi(v) & 1) e->noise.wall = nullptr;
        if (std::atoi(v) & 2) e->noise.r1 = nullptr;
    }
    return DDGI_OK;
}

static int upload_local_rays(ddgi_engine* e)
{
    const GridK g = make_grid(e);
    const size_t n = static_cast<size_t>(g.s) * g.s;
    const size_t local_probes = static_cast<size_t>(g.cx) * g.cy * g.czl;
    const size_t local_rays = local_probes * n;
    if (local_rays > e->d_rays_capacity)
    {
        if (e->d_rays) (void)hipFree(e->d_rays);
        e->d_rays = nullptr;
        e->d_rays_capacity = 0;
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&e->d_rays), local_rays * sizeof(ddgi_probe_ray)));
        e->d_rays_capacity = local_rays;
    }
    // this rank's probes in reference order: for each y, the z-range [z0, z0+czl) is one contiguous run
    const size_t run = static_cast<size_t>(g.czl) * g.cx * n;  // rays per (y) run
    for (int y = 0; y < g.cy; ++y)
    {
        const size_t src = (static_cast<size_t>(y) * g.cx * g.cz + static_cast<size_t>(g.z0) * g.cx) * n;
        const size_t dst = static_cast<size_t>(y) * run;
        HIP_TRY(hipMemcpyAsync(reinterpret_cast<ddgi_probe_ray*>(e->d_rays) + dst, e->host_rays.data() + src,
                               run * sizeof(ddgi_probe_ray), hipMemcpyHostToDevice, e->stream));
    }
    HIP_TRY(hipStreamSynchronize(e->stream));
    e->n_local_rays = static_cast<uint32_t>(local_rays);
    return DDGI_OK;
}

// ---- C ABI -----------------------------------------------------------------------------------------------

extern "C" {

int ddgi_abi_version(void) { return DDGI_ABI_VERSION; }

const char* ddgi_last_error(void) { return g_last_error.c_str(); }

int ddgi_create_sharded(const ddgi_irradiance_field* field, const ddgi_render_settings* settings, int device,
                        int rank, int world, ddgi_handle* out)
{
    if (!out) return fail(DDGI_ERR_INVALID_ARGUMENT, "null out handle");
    *out = nullptr;
    if (int rc = validate_config(field, settings, world)) return rc;
    if (rank < 0 || rank >= world) return fail(DDGI_ERR_INVALID_ARGUMENT, "rank %d not in [0,%d)", rank, world);
    int count = 0;
    hipError_t he = hipGetDeviceCount(&count);
    if (he != hipSuccess || count <= 0)
        return fail(DDGI_ERR_NO_DEVICE, "no HIP device available (%s); this library has no CPU path",
                    he == hipSuccess ? "device count is 0" : hipGetErrorString(he));
    if (device < 0 || device >= count) return fail(DDGI_ERR_NO_DEVICE, "device %d not in [0,%d)", device, count);
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(DDGI_ERR_NO_DEVICE, "device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);

    ddgi_engine* e = new (std::nothrow) ddgi_engine();
    if (!e) return fail(DDGI_ERR_OUT_OF_MEMORY, "host allocation failed");
    e->device = device;
    e->rank = rank;
    e->world = world;
    e->field = *field;
    e->settings = *settings;
    e->num_cus = prop.multiProcessorCount;
    for (int s = 0; s < 3; ++s) shipped_lights(s, e->lights[s], &e->n_lights[s]);
    int rc = DDGI_OK;
    do
    {
        if (hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking) != hipSuccess)
        {
            rc = fail(DDGI_ERR_HIP, "hipStreamCreate failed");
            break;
        }
        e->own_stream = true;
        for (auto& triple : e->ev)
            for (auto& ev : triple)
                if (hipEventCreate(&ev) != hipSuccess) rc = fail(DDGI_ERR_HIP, "hipEventCreate failed");
        if (rc) break;
        rc = alloc_textures(e);
    } while (0);
    if (rc)
    {
        ddgi_destroy(e);
        return rc;
    }
    *out = e;
    return DDGI_OK;
}

int ddgi_create(const ddgi_irradiance_field* field, const ddgi_render_settings* settings, int device, ddgi_handle* out)
{
    return ddgi_create_sharded(field, settings, device, 0, 1, out);
}

int ddgi_destroy(ddgi_handle e)
{
    if (!e) return DDGI_OK;
    (void)hipSetDevice(e->device);
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    for (int i = 0; i < 2; ++i)
        if (e->own_tex[i]) (void)hipFree(e->own_tex[i]);
    if (e->d_rays) (void)hipFree(e->d_rays);
    if (e->d_stats) (void)hipFree(e->d_stats);
    if (e->d_blend_w) (void)hipFree(e->d_blend_w);
    if (e->d_work) (void)hipFree(e->d_work);
    if (e->d_wf_cold) (void)hipFree(e->d_wf_cold);
    if (e->d_wf_dir) (void)hipFree(e->d_wf_dir);
    if (e->d_radiance) (void)hipFree(e->d_radiance);
    for (auto& p : e->d_noise)
        if (p) (void)hipFree(p);
    for (auto& d : e->dev_scene)
    {
        if (d.bits) (void)hipFree(d.bits);
        if (d.types) (void)hipFree(d.types);
    }
    for (auto& triple : e->ev)
        for (auto& ev : triple)
            if (ev) (void)hipEventDestroy(ev);
    if (e->own_stream && e->stream) (void)hipStreamDestroy(e->stream);
    delete e;
    return DDGI_OK;
}

int ddgi_configure(ddgi_handle e, const ddgi_irradiance_field* field, const ddgi_render_settings* settings)
{
    if (!e) return fail(DDGI_ERR_INVALID_ARGUMENT, "null handle");
    if (int rc = validate_config(field, settings, e->world)) return rc;
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipStreamSynchronize(e->stream));
    e->field = *field;
    e->settings = *settings;
    e->host_rays.clear();
    e->n_local_rays = 0;
    e->updates = 0;
    return alloc_textures(e);
}

// Probe coordinate along one axis: RVPT::generate_probe_rays, src/rvpt/rvpt.cpp:1199-1205 (ddgi_oct.h: probe_position)
static float probe_axis_position(int idx, int count, int side, float origin)
{
    return static_cast<float>(idx - (count - 1) / 2) * static_cast<float>(side) + origin;
}

int ddgi_reconfigure(ddgi_handle e, const ddgi_irradiance_field* field, const ddgi_render_settings* settings, int carry_over)
{
    if (!e) return fail(DDGI_ERR_INVALID_ARGUMENT, "null handle");
    if (!carry_over) return ddgi_configure(e, field, settings);
    if (int rc = validate_config(field, settings, e->world)) return rc;
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipStreamSynchronize(e->stream));
    const ddgi_irradiance_field old = e->field;
    // a tile can be carried over only if it has the same size: always in DDGI mode (8x8 / 16x16
    // octahedral tiles), in REF mode when the rays per probe did not change
    const bool same_tiles = e->mode == DDGI_MODE_DDGI || old.sqrt_rays_per_probe == field->sqrt_rays_per_probe;
    // per axis: which old probe index has exactly the new probe's coordinate (-1: none)
    std::vector<int> axis_map[3];
    for (int a = 0; a < 3; ++a)
    {
        axis_map[a].assign(field->probe_count[a], -1);
        for (int i = 0; i < field->probe_count[a]; ++i)
        {
            const float x = probe_axis_position(i, field->probe_count[a], field->side_length, field->field_origin[a]);
            for (int j = 0; j < old.probe_count[a]; ++j)
                if (probe_axis_position(j, old.probe_count[a], old.side_length, old.field_origin[a]) == x) axis_map[a][i] = j;
        }
    }
    const int cx = field->probe_count[0], cy = field->probe_count[1], cz = field->probe_count[2];
    std::vector<int32_t> map(static_cast<size_t>(cx) * cy * cz, -1);  // slab-major slots (z*cy + y)*cx + x
    size_t carried = 0;
    if (same_tiles)
        for (int z = 0; z < cz; ++z)
            for (int y = 0; y < cy; ++y)
                for (int x = 0; x < cx; ++x)
                {
                    const int ox = axis_map[0][x], oy = axis_map[1][y], oz = axis_map[2][z];
                    if (ox < 0 || oy < 0 || oz < 0) continue;
                    map[(static_cast<size_t>(z) * cy + y) * cx + x] = (oz * old.probe_count[1] + oy) * old.probe_count[0] + ox;
                    carried += 1;
                }
    // keep the old textures alive across the re-allocation
    void* old_own[2] = {e->own_tex[0], e->own_tex[1]};
    const void* old_tex[2] = {e->tex[0], e->tex[1]};
    const size_t old_words[2] = {e->tex_bytes[0] / 4 / (static_cast<size_t>(old.probe_count[0]) * old.probe_count[1] * old.probe_count[2]),
                                 e->tex_bytes[1] / 4 / (static_cast<size_t>(old.probe_count[0]) * old.probe_count[1] * old.probe_count[2])};
    e->own_tex[0] = e->own_tex[1] = nullptr;
    const uint32_t frame = e->frame;
    e->field = *field;
    e->settings = *settings;
    e->host_rays.clear();
    e->n_local_rays = 0;
    e->updates = 0;
    int rc = alloc_textures(e);
    e->frame = frame;  // the temporal sequence (ray rotation, RNG keys) goes on
    if (rc == DDGI_OK && carried > 0)
    {
        int32_t* d_map = nullptr;
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_map), map.size() * sizeof(int32_t)));
        HIP_TRY(hipMemcpyAsync(d_map, map.data(), map.size() * sizeof(int32_t), hipMemcpyHostToDevice, e->stream));
        for (int i = 0; i < 2; ++i)
            HIP_TRY(launch_carry_tiles(e->tex[i], old_tex[i], d_map, static_cast<uint32_t>(map.size()), static_cast<uint32_t>(old_words[i]), e->stream));
        HIP_TRY(hipStreamSynchronize(e->stream));
        (void)hipFree(d_map);
    }
    for (void* p : old_own)
        if (p) (void)hipFree(p);
    return rc;
}

int ddgi_set_mode(ddgi_handle e, int mode)
{
    if (!e) return fail(DDGI_ERR_INVALID_ARGUMENT, "null handle");
    if (mode != DDGI_MODE_REF && mode != DDGI_MODE_DDGI) return fail(DDGI_ERR_INVALID_ARGUMENT, "unknown mode %d", mode);
    if (mode == e->mode) return DDGI_OK;
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipStreamSynchronize(e->stream));
    if (e->tex[0] != e->own_tex[0]) return fail(DDGI_ERR_INVALID_ARGUMENT, "unbind caller textures before changing the mode");
    e->mode = mode;
    e->updates = 0;
    return alloc_textures(e);  // the two modes keep differently shaped textures; both start zeroed
}

int ddgi_set_lights(ddgi_handle e, int scene, const ddgi_light* lights, int n)
{
    if (!e) return fail(DDGI_ERR_INVALID_ARGUMENT, "null handle");
    if (scene < 0 || scene > 3 || n < 0 || n > kMaxLights || (n > 0 && !lights))
        return fail(DDGI_ERR_INVALID_ARGUMENT, "bad light table (scene %d, n %d)", scene, n);
    for (int i = 0; i < n; ++i)
    {
        e->lights[scene][i].intensity = lights[i].intensity;
        for (int a = 0; a < 3; ++a)
        {
            e->lights[scene][i].col[a] = lights[i].col[a];
            e->lights[scene][i].pos[a] = lights[i].pos[a];
        }
    }
    e->n_lights[scene] = n;
    return DDGI_OK;
}

int ddgi_generate_probe_rays(ddgi_handle e, uint32_t seed, int reseed)
{
    if (!e) return fail(DDGI_ERR_INVALID_ARGUMENT, "null handle");
    HIP_TRY(hipSetDevice(e->device));
    if (!e->rand_seeded || reseed)
    {
        e->rand.seed(seed);
        e->rand_seeded = true;
    }
    generate_probe_rays(e->field, e->rand, e->host_rays);
    return upload_local_rays(e);
}

int ddgi_upload_probe_rays(ddgi_handle e, const ddgi_probe_ray* rays, size_t n)
{
    if (!e || !rays) return fail(DDGI_ERR_INVALID_ARGUMENT, "null handle/rays");
    HIP_TRY(hipSetDevice(e->device));
    const GridK g = make_grid(e);
    const size_t probes = static_cast<size_t>(g.cx) * g.cy * g.cz;
    const size_t expect = probes * g.s * g.s;
    if (n != expect) return fail(DDGI_ERR_INVALID_ARGUMENT, "expected %zu rays (full grid), got %zu", expect, n);
    // the reference's shader trusts probe_info blindly (Q13); an out-of-range tile would write
    // outside the texture, so reject it here
    for (size_t i = 0; i < n; ++i)
    {
        const float p = rays[i].probe_info[0], tx = rays[i].probe_info[1], ty = rays[i].probe_info[2];
        if (!(p >= 0.0f && p < static_cast<float>(probes) && tx >= 0.0f && tx < static_cast<float>(g.s) && ty >= 0.0f &&
              ty < static_cast<float>(g.s)))
            return fail(DDGI_ERR_INVALID_ARGUMENT, "ray %zu: probe_info (%g,%g,%g) outside the grid", i, p, tx, ty);
    }
    e->host_rays.assign(rays, rays + n);
    return upload_local_rays(e);
}

int ddgi_get_probe_rays(ddgi_handle e, ddgi_probe_ray* rays, size_t n)
{
    if (!e || !rays) return fail(DDGI_ERR_INVALID_ARGUMENT, "null handle/rays");
    if (e->host_rays.empty()) return fail(DDGI_ERR_NOT_READY, "no probe rays generated or uploaded yet");
    if (n < e->host_rays.size()) return fail(DDGI_ERR_INVALID_ARGUMENT, "capacity %zu < %zu rays", n, e->host_rays.size());
    std::memcpy(rays, e->host_rays.data(), e->host_rays.size() * sizeof(ddgi_probe_ray));
    return DDGI_OK;
}

// What the march/event balance of the queue kernel depends on: grid, rays, scene, bounces, lights, mode, shard.
static unsigned long long aq_config_key(const ddgi_engine* e, const TraceArgs& a, bool ddgi_mode)
{
    unsigned long long h = 1469598103934665603ull;
    auto mix = [&](unsigned long long v) { h = (h ^ v) * 1099511628211ull; };
    for (int i = 0; i < 3; ++i) mix(static_cast<unsigned>(e->field.probe_count[i]));
    mix(static_cast<unsigned>(e->field.side_length));
    mix(static_cast<unsigned>(e->field.sqrt_rays_per_probe));
    for (int i = 0; i < 3; ++i)
    {
        unsigned u;
        std::memcpy(&u, &e->field.field_origin[i], 4);
        mix(u);
    }
    mix(static_cast<unsigned>(a.scene_id)), mix(static_cast<unsigned>(a.max_bounces)), mix(static_cast<unsigned>(a.nl));
    mix(ddgi_mode ? 1u : 0u), mix(static_cast<unsigned>(e->rank)), mix(static_cast<unsigned>(e->world)), mix(a.n_rays);
    return h | 1ull;
}

int ddgi_probe_update(ddgi_handle e, const ddgi_render_settings* settings)
{
    if (!e) return fail(DDGI_ERR_INVALID_ARGUMENT, "null handle");
    if (settings)
    {
        if (settings->scene < 0 || settings->scene > 3) return fail(DDGI_ERR_INVALID_ARGUMENT, "scene %d not in {0,1,2,3}", settings->scene);
        e->settings = *settings;
    }
    const bool ddgi_mode = e->mode == DDGI_MODE_DDGI;
    if (!ddgi_mode && e->n_local_rays == 0) return fail(DDGI_ERR_NOT_READY, "ddgi_probe_update before any probe rays were generated/uploaded");
    HIP_TRY(hipSetDevice(e->device));
    const int scene = e->settings.scene;
    if (int rc = ensure_scene(e, scene)) return rc;
    if (int rc = ensure_noise(e)) return rc;

    TraceArgs a{};
    a.grid = make_grid(e);
    a.scene = e->dev_scene[scene].k;
    a.scene_id = scene;
    a.max_bounces = e->settings.max_bounces;
    a.nl = e->n_lights[scene];
    for (int i = 0; i < a.nl; ++i) a.lights[i] = e->lights[scene][i];
    a.rays = e->d_rays;
    a.n_rays = e->n_local_rays;
    if (ddgi_mode)
    {
        // rays are generated in the kernel; lights follow update_lights(time) (probe_pass.comp:217-251)
        const size_t local_rays = static_cast<size_t>(a.grid.cx) * a.grid.cy * a.grid.czl * a.grid.s * a.grid.s;
        const size_t rec_pairs = blend_record_groups(static_cast<uint32_t>(a.grid.cx) * a.grid.cy * a.grid.czl) * a.grid.s * a.grid.s;
        if (rec_pairs > e->d_radiance_capacity)
        {
            if (e->d_radiance) (void)hipFree(e->d_radiance);
            e->d_radiance = nullptr;
            e->d_radiance_capacity = 0;
            HIP_TRY(hipMalloc(reinterpret_cast<void**>(&e->d_radiance), rec_pairs * 40 * sizeof(float)));
            HIP_TRY(hipMemsetAsync(e->d_radiance, 0, rec_pairs * 40 * sizeof(float), e->stream));
            e->d_radiance_capacity = rec_pairs;
        }
        animate_lights(scene, e->settings.time, e->lights[scene], a.nl, a.lights);
        a.ddgi = 1;
        a.frame_key = frame_key(e->frame);
        frame_rotation(e->frame, a.rot);
        a.rad_rgb = e->d_radiance;
        a.rad_dd = e->d_radiance + rec_pairs * 24;
        a.rays = nullptr;
        a.n_rays = static_cast<uint32_t>(local_rays);
    }
    a.albedo = static_cast<uint32_t*>(e->tex[0]);
    a.distance = static_cast<uint32_t*>(e->tex[1]);
    a.wait_threshold = e->wait_threshold;
    a.stats = e->d_stats;
    a.noise = e->noise;
    if (const char* v = std::getenv("DDGI_WAIT_THRESHOLD")) a.wait_threshold = std::atoi(v);
    if (const char* v = std::getenv("DDGI_ABLATE")) a.ablate = std::atoi(v);
    if (const char* v = std::getenv("DDGI_WF_TAIL")) a.wf_tail = std::atoi(v);
    if (const char* v = std::getenv("DDGI_WF_FETCH")) a.wf_fetch = std::atoi(v);
    if (const char* v = std::getenv("DDGI_WF_CHUNK")) a.wf_chunk = std::atoi(v);
    if (const char* v = std::getenv("DDGI_WF_DRAIN")) a.wf_drain = std::atoi(v);

    // Kernel choice: the wavefront kernel (one persistent 1024-lane workgroup per CU, ray pool in
    // LDS) whenever its pool fits next to the scene bitmap; the ray-per-lane kernel otherwise
    // (or when DDGI_TRACE_KERNEL=lane asks for it, e.g. to cross-check the two).
    const char* kernel_env = std::getenv("DDGI_TRACE_KERNEL");
    const bool force_lane = kernel_env && std::strcmp(kernel_env, "lane") == 0;
    if (ddgi_mode && (force_lane || a.max_bounces < 1)) return fail(DDGI_ERR_UNSUPPORTED, "DDGI mode needs the wavefront trace kernel and max_bounces >= 1");
    int wf_threads = 1024;
    if (const char* v = std::getenv("DDGI_WF_THREADS")) wf_threads = std::atoi(v) == 512 ? 512 : 1024;
    const int wf_blocks_per_cu = 1024 / wf_threads;
    int pool = (force_lane || a.max_bounces < 1) ? 0 : wf_pool_size(a.scene.nwords, a.nl > 1, 160 * 1024 / wf_blocks_per_cu, wf_threads);
    if (const char* v = std::getenv("DDGI_WF_POOL")) pool = pool ? std::min(pool, std::max(wf_threads, std::atoi(v) / 64 * 64)) : 0;
    // the barrier-free queue kernel (k_probe_trace_aq) whenever its pool fits; DDGI_TRACE_KERNEL=rounds asks
    // for the round-based k_probe_trace_wf (cross-check), which also serves the utilisation counters
    const bool ask_queues = kernel_env && std::strcmp(kernel_env, "queues") == 0;
    const bool force_rounds = (kernel_env && std::strcmp(kernel_env, "rounds") == 0) || (a.stats != nullptr && !ask_queues) || wf_threads != 1024;
    const bool use_async = pool > 0 && !force_rounds && aq_pool_size(a.scene.nwords, 160 * 1024) > 0;
    if (use_async)
    {
        pool = std::min(1536, aq_pool_size(a.scene.nwords, 160 * 1024));  // more slots than ~1300 buy nothing (C3: 1024: 3.64 ms, 1280: 3.36, 1536: 3.35, 2048: 3.39)
        if (const char* v = std::getenv("DDGI_AQ_POOL")) pool = std::min(aq_pool_size(a.scene.nwords, 160 * 1024), std::max(1024, std::atoi(v) / 64 * 64));
    }

    hipEvent_t* ev = e->ev[e->updates % ddgi_engine::kRing];
    if (pool > 0)
    {
        if (!e->d_work)
        {
            HIP_TRY(hipMalloc(reinterpret_cast<void**>(&e->d_work), 2 * sizeof(uint32_t)));  // [0] ray counter, [1] kernel status
            HIP_TRY(hipMemsetAsync(e->d_work, 0, 2 * sizeof(uint32_t), e->stream));
        }
        // one persistent workgroup per CU unless the launch is tiny (the launcher sizes the ray claims
        // so that every workgroup gets several: launch_probe_trace_wf)
        const uint32_t chunks = (a.n_rays + 255u) / 256u;
        uint32_t grid = static_cast<uint32_t>(e->num_cus * wf_blocks_per_cu);
        if (grid > chunks) grid = chunks;
        const size_t slots = static_cast<size_t>(grid) * pool;
        if (slots > e->wf_cold_slots)
        {
            if (e->d_wf_cold) (void)hipFree(e->d_wf_cold);
            e->d_wf_cold = nullptr;
            e->wf_cold_slots = 0;
            HIP_TRY(hipMalloc(&e->d_wf_cold, slots * 48));
            e->wf_cold_slots = slots;
        }
        if (a.nl > 1 && slots > e->wf_dir_slots)
        {
            if (e->d_wf_dir) (void)hipFree(e->d_wf_dir);
            e->d_wf_dir = nullptr;
            e->wf_dir_slots = 0;
            HIP_TRY(hipMalloc(reinterpret_cast<void**>(&e->d_wf_dir), slots * sizeof(float4)));
            e->wf_dir_slots = slots;
        }
        a.wf_cold = e->d_wf_cold;
        a.wf_dir = e->d_wf_dir;
        int march_waves = 5;
        if (use_async)
        {
            // How many of the 16 waves march (the rest run events) is the one knob the balance of a scene
            // moves: C3 is fastest at 5 (4: 3.44 ms, 5: 2.96, 6: 3.14, 8: 3.7), Cornell and the house at 6,
            // a sparser cave grid at 3.  The first update of a configuration measures it: a few extra
            // launches of the same (idempotent) trace, hill-climbing from the last value.
            const unsigned long long key = aq_config_key(e, a, ddgi_mode);
            if (const char* v = std::getenv("DDGI_AQ_MARCH")) march_waves = std::min(15, std::max(1, std::atoi(v)));
            else if (e->aq_key == key && e->aq_march > 0) march_waves = e->aq_march;
            else
            {
                const char* tune = std::getenv("DDGI_AUTOTUNE");
                march_waves = e->aq_march > 0 ? e->aq_march : 5;
                if (!(tune && std::atoi(tune) == 0) && a.ablate == 0)  // (not under the profiling / fault-injection switches)
                {
                    auto timed = [&](int mw, float* ms) -> int {  // the faster of two launches
                        *ms = 0.0f;
                        for (int rep = 0; rep < 2; ++rep)
                        {
                            float t = 0.0f;
                            HIP_TRY(hipEventRecord(ev[0], e->stream));
                            HIP_TRY(launch_probe_trace_aq(a, pool, static_cast<int>(grid), mw, e->d_work, e->d_work + 1, e->stream));
                            HIP_TRY(hipEventRecord(ev[1], e->stream));
                            HIP_TRY(hipEventSynchronize(ev[1]));
                            HIP_TRY(hipEventElapsedTime(&t, ev[0], ev[1]));
                            if (rep == 0 || t < *ms) *ms = t;
                        }
                        return DDGI_OK;
                    };
                    float best_ms = 0.0f, ms = 0.0f;
                    // first touches, cold caches, clocks still low: repeat the starting point until it settles
                    if (int rc = timed(march_waves, &ms)) return rc;
                    for (int settle = 0; settle < 6; ++settle)
                    {
                        if (int rc = timed(march_waves, &best_ms)) return rc;
                        const bool steady = std::fabs(best_ms - ms) < 0.015f * best_ms;
                        ms = best_ms;
                        if (steady) break;
                    }
                    for (int dir = -1; dir <= 1; dir += 2)
                    {
                        bool moved = false;
                        for (int mw = march_waves + dir; mw >= 2 && mw <= 12; mw += dir)
                        {
                            if (int rc = timed(mw, &ms)) return rc;
                            if (ms >= best_ms) break;
                            best_ms = ms, march_waves = mw, moved = true;
                        }
                        if (moved) break;  // downhill in this direction: the other one was uphill
                    }
                }
                e->aq_key = key;
                e->aq_march = march_waves;
                if (std::getenv("DDGI_VERBOSE")) std::fprintf(stderr, "[ddgi] queue kernel: %d march waves / %d event waves for this configuration\n", march_waves, 16 - march_waves);
            }
        }
        HIP_TRY(hipEventRecord(ev[0], e->stream));
        if (use_async)
            HIP_TRY(launch_probe_trace_aq(a, pool, static_cast<int>(grid), march_waves, e->d_work, e->d_work + 1, e->stream));
        else
            HIP_TRY(launch_probe_trace_wf(a, wf_threads, pool, static_cast<int>(grid), e->d_work, e->stream));
    }
    else
    {
        const size_t lds = static_cast<size_t>(a.scene.nwords) * sizeof(uint32_t);
        int per_cu = 0;
        HIP_TRY(trace_kernel_occupancy(&per_cu, lds));
        if (per_cu < 1) return fail(DDGI_ERR_UNSUPPORTED, "scene bitmap (%zu B) does not fit in LDS", lds);
        const uint32_t chunks = (a.n_rays + kTraceBlock - 1) / kTraceBlock;
        uint32_t grid = static_cast<uint32_t>(e->num_cus) * static_cast<uint32_t>(per_cu);
        if (grid > chunks) grid = chunks;
        HIP_TRY(hipEventRecord(ev[0], e->stream));
        HIP_TRY(launch_probe_trace_ref(a, static_cast<int>(grid), e->stream));
    }
    HIP_TRY(hipEventRecord(ev[1], e->stream));
    if (ddgi_mode)
    {
        if (pool <= 0) return fail(DDGI_ERR_UNSUPPORTED, "DDGI mode: the ray pool does not fit in LDS next to the scene bitmap");
        BlendArgs b{};
        b.grid = a.grid;
        for (int i = 0; i < 9; ++i) b.rot[i] = a.rot[i];
        b.rad_rgb = a.rad_rgb;
        b.rad_dd = a.rad_dd;
        b.irradiance = static_cast<float*>(e->tex[0]);
        b.depth = static_cast<float*>(e->tex[1]);
        b.n_local_probes = static_cast<uint32_t>(a.grid.cx) * a.grid.cy * a.grid.czl;
        {
            const size_t need = blend_weights_floats(a.grid.s * a.grid.s) + 256;
            if (need > e->d_blend_w_floats)
            {
                if (e->d_blend_w) (void)hipFree(e->d_blend_w);
                e->d_blend_w = nullptr, e->d_blend_w_floats = 0;
                HIP_TRY(hipMalloc(reinterpret_cast<void**>(&e->d_blend_w), need * sizeof(float)));
                e->d_blend_w_floats = need;
            }
            b.w_sum = e->d_blend_w;
            b.w = e->d_blend_w + 256;
            if (std::getenv("DDGI_BLEND_KERNEL")) b.w = b.w_sum = nullptr;  // "probe": one probe per workgroup, weights in place
        }
        HIP_TRY(launch_probe_blend(b, e->num_cus, e->stream));
        e->frame += 1;
    }
    HIP_TRY(hipEventRecord(ev[2], e->stream));
    e->updates += 1;
    return DDGI_OK;
}

// After a stream synchronisation: did a trace kernel trip its safety net (k_probe_trace_aq: a queue wait
// that never ended)?  Then the textures are not valid — say so instead of returning them.
static int check_kernel_status(ddgi_engine* e)
{
    if (!e->d_work) return DDGI_OK;
    uint32_t status = 0;
    HIP_TRY(hipMemcpy(&status, e->d_work + 1, sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (status != 0) return fail(DDGI_ERR_HIP, "the trace kernel aborted (status %u): the probe textures are not valid", status);
    return DDGI_OK;
}

int ddgi_synchronize(ddgi_handle e)
{
    if (!e) return fail(DDGI_ERR_INVALID_ARGUMENT, "null handle");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipStreamSynchronize(e->stream));
    return check_kernel_status(e);
}

int ddgi_last_update_ms(ddgi_handle e, float* trace_ms, float* blend_ms, float* total_ms)
{
    if (!e) return fail(DDGI_ERR_INVALID_ARGUMENT, "null handle");
    if (e->updates == 0) return fail(DDGI_ERR_NOT_READY, "no probe update has been issued yet");
    HIP_TRY(hipSetDevice(e->device));
    hipEvent_t* ev = e->ev[(e->updates - 1) % ddgi_engine::kRing];
    HIP_TRY(hipEventSynchronize(ev[2]));
    float t01 = 0.f, t12 = 0.f, t02 = 0.f;
    HIP_TRY(hipEventElapsedTime(&t01, ev[0], ev[1]));
    HIP_TRY(hipEventElapsedTime(&t12, ev[1], ev[2]));
    HIP_TRY(hipEventElapsedTime(&t02, ev[0], ev[2]));
    if (trace_ms) *trace_ms = t01;
    if (blend_ms) *blend_ms = e->mode == DDGI_MODE_REF ? 0.f : t12;
    if (total_ms) *total_ms = t02;
    return DDGI_OK;
}

int ddgi_update_history_ms(ddgi_handle e, float* trace_ms, float* blend_ms, int capacity, int* n_out)
{
    if (!e || !n_out || capacity < 0) return fail(DDGI_ERR_INVALID_ARGUMENT, "null handle/n_out");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipStreamSynchronize(e->stream));
    unsigned long long have = e->updates < static_cast<unsigned long long>(ddgi_engine::kRing) ? e->updates : ddgi_engine::kRing;
    if (have > static_cast<unsigned long long>(capacity)) have = capacity;
    for (unsigned long long i = 0; i < have; ++i)
    {
        hipEvent_t* ev = e->ev[(e->updates - have + i) % ddgi_engine::kRing];
        float t01 = 0.f, t12 = 0.f;
        HIP_TRY(hipEventElapsedTime(&t01, ev[0], ev[1]));
        HIP_TRY(hipEventElapsedTime(&t12, ev[1], ev[2]));
        if (trace_ms) trace_ms[i] = t01;
        if (blend_ms) blend_ms[i] = e->mode == DDGI_MODE_REF ? 0.f : t12;
    }
    *n_out = static_cast<int>(have);
    return DDGI_OK;
}

int ddgi_trace_stats(ddgi_handle e, int enable, unsigned long long* out64)
{
    if (!e) return fail(DDGI_ERR_INVALID_ARGUMENT, "null handle");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipStreamSynchronize(e->stream));
    if (out64)
    {
        std::memset(out64, 0, 64 * sizeof(unsigned long long));
        if (e->d_stats) HIP_TRY(hipMemcpy(out64, e->d_stats, 64 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    }
    if (enable && !e->d_stats) HIP_TRY(hipMalloc(reinterpret_cast<void**>(&e->d_stats), 64 * sizeof(unsigned long long)));
    if (e->d_stats) HIP_TRY(hipMemset(e->d_stats, 0, 64 * sizeof(unsigned long long)));
    if (!enable && e->d_stats)
    {
        (void)hipFree(e->d_stats);
        e->d_stats = nullptr;
    }
    return DDGI_OK;
}

int ddgi_read_textures(ddgi_handle e, uint8_t* albedo, uint8_t* distance)
{
    if (!e) return fail(DDGI_ERR_INVALID_ARGUMENT, "null handle");
    if (e->mode != DDGI_MODE_REF) return fail(DDGI_ERR_UNSUPPORTED, "ddgi_read_textures is REF-mode only; use ddgi_read_tiles");
    HIP_TRY(hipSetDevice(e->device));
    const GridK g = make_grid(e);
    const size_t s = g.s, s2 = s * s;
    const size_t W = static_cast<size_t>(g.cx) * g.cz * s;
    std::vector<uint32_t> slab(e->tex_bytes[0] / 4);
    uint8_t* outs[2] = {albedo, distance};
    for (int t = 0; t < 2; ++t)
    {
        if (!outs[t]) continue;
        HIP_TRY(hipMemcpyAsync(slab.data(), e->tex[t], e->tex_bytes[t], hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(hipStreamSynchronize(e->stream));
        if (int rc = check_kernel_status(e)) return rc;
        uint32_t* raster = reinterpret_cast<uint32_t*>(outs[t]);
        // slab-major [z][y][x][ty][tx] -> reference raster: tile of probe p at ((p mod cx*cz)*s, (p div cx*cz)*s)
        for (int z = 0; z < g.cz; ++z)
            for (int y = 0; y < g.cy; ++y)
                for (int x = 0; x < g.cx; ++x)
                {
                    const uint32_t* tile = slab.data() + ((static_cast<size_t>(z) * g.cy + y) * g.cx + x) * s2;
                    const size_t col0 = (static_cast<size_t>(z) * g.cx + x) * s;
                    const size_t row0 = static_cast<size_t>(y) * s;
                    for (size_t ty = 0; ty < s; ++ty)
                        std::memcpy(raster + (row0 + ty) * W + col0, tile + ty * s, s * 4);
                }
    }
    return DDGI_OK;
}

int ddgi_read_tiles(ddgi_handle e, float* irradiance, float* depth)
{
    if (!e) return fail(DDGI_ERR_INVALID_ARGUMENT, "null handle");
    if (e->mode != DDGI_MODE_DDGI) return fail(DDGI_ERR_UNSUPPORTED, "ddgi_read_tiles is DDGI-mode only; use ddgi_read_textures");
    HIP_TRY(hipSetDevice(e->device));
    const GridK g = make_grid(e);
    float* outs[2] = {irradiance, depth};
    const size_t per_probe[2] = {8 * 8 * 4, 16 * 16 * 2};
    for (int t = 0; t < 2; ++t)
    {
        if (!outs[t]) continue;
        std::vector<float> slab(e->tex_bytes[t] / sizeof(float));
        HIP_TRY(hipMemcpyAsync(slab.data(), e->tex[t], e->tex_bytes[t], hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(hipStreamSynchronize(e->stream));
        if (int rc = check_kernel_status(e)) return rc;
        // slab-major [z][y][x] -> reference probe order p = y*cx*cz + z*cx + x
        for (int z = 0; z < g.cz; ++z)
            for (int y = 0; y < g.cy; ++y)
                for (int x = 0; x < g.cx; ++x)
                {
                    const size_t q = (static_cast<size_t>(z) * g.cy + y) * g.cx + x;
                    const size_t p = static_cast<size_t>(y) * g.cx * g.cz + static_cast<size_t>(z) * g.cx + x;
                    std::memcpy(outs[t] + p * per_probe[t], slab.data() + q * per_probe[t], per_probe[t] * sizeof(float));
                }
    }
    return DDGI_OK;
}

int ddgi_set_frame(ddgi_handle e, uint32_t frame)
{
    if (!e) return fail(DDGI_ERR_INVALID_ARGUMENT, "null handle");
    e->frame = frame;
    return DDGI_OK;
}

int ddgi_sample_device(ddgi_handle e, const float* d_pos, const float* d_nrm, size_t n, float* d_rgb, int32_t* d_cage)
{
    if (!e) return fail(DDGI_ERR_INVALID_ARGUMENT, "null handle");
    if (n == 0) return DDGI_OK;
    if (!d_pos || !d_nrm || !d_rgb) return fail(DDGI_ERR_INVALID_ARGUMENT, "null device pointer");
    if (n > 0xffffffffull) return fail(DDGI_ERR_INVALID_ARGUMENT, "too many points");
    HIP_TRY(hipSetDevice(e->device));
    SampleArgs a{};
    a.grid = make_grid(e);
    a.albedo = static_cast<const uint32_t*>(e->tex[0]);
    a.distance = static_cast<const uint32_t*>(e->tex[1]);
    a.pos = d_pos;
    a.nrm = d_nrm;
    a.rgb = d_rgb;
    a.cage = d_cage;
    a.n = static_cast<uint32_t>(n);
    if (e->mode == DDGI_MODE_DDGI)
    {
        a.irradiance = static_cast<const float*>(e->tex[0]);
        a.depth = static_cast<const float*>(e->tex[1]);
        HIP_TRY(launch_probe_sample_ddgi(a, e->stream));
        return DDGI_OK;
    }
    HIP_TRY(launch_probe_sample_ref(a, e->stream));
    return DDGI_OK;
}

int ddgi_sample(ddgi_handle e, const float* pos, const float* nrm, size_t n, float* rgb, int32_t* cage)
{
    if (!e) return fail(DDGI_ERR_INVALID_ARGUMENT, "null handle");
    if (n == 0) return DDGI_OK;
    if (!pos || !nrm || !rgb) return fail(DDGI_ERR_INVALID_ARGUMENT, "null pointer");
    HIP_TRY(hipSetDevice(e->device));
    float *d_pos = nullptr, *d_nrm = nullptr, *d_rgb = nullptr;
    int32_t* d_cage = nullptr;
    int rc = DDGI_OK;
    auto cleanup = [&] {
        if (d_pos) (void)hipFree(d_pos);
        if (d_nrm) (void)hipFree(d_nrm);
        if (d_rgb) (void)hipFree(d_rgb);
        if (d_cage) (void)hipFree(d_cage);
    };
#define TRY_OR_CLEAN(expr)                                                                     \
    do                                                                                         \
    {                                                                                          \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
        {                                                                                      \
            cleanup();                                                                         \
            return fail(e_ == hipErrorOutOfMemory ? DDGI_ERR_OUT_OF_MEMORY : DDGI_ERR_HIP,     \
                        "%s failed: %s", #expr, hipGetErrorString(e_));                        \
        }                                                                                      \
    } while (0)
    TRY_OR_CLEAN(hipMalloc(reinterpret_cast<void**>(&d_pos), n * 12));
    TRY_OR_CLEAN(hipMalloc(reinterpret_cast<void**>(&d_nrm), n * 12));
    TRY_OR_CLEAN(hipMalloc(reinterpret_cast<void**>(&d_rgb), n * 12));
    if (cage) TRY_OR_CLEAN(hipMalloc(reinterpret_cast<void**>(&d_cage), n * 32));
    TRY_OR_CLEAN(hipMemcpyAsync(d_pos, pos, n * 12, hipMemcpyHostToDevice, e->stream));
    TRY_OR_CLEAN(hipMemcpyAsync(d_nrm, nrm, n * 12, hipMemcpyHostToDevice, e->stream));
    rc = ddgi_sample_device(e, d_pos, d_nrm, n, d_rgb, d_cage);
    if (rc)
    {
        cleanup();
        return rc;
    }
    TRY_OR_CLEAN(hipMemcpyAsync(rgb, d_rgb, n * 12, hipMemcpyDeviceToHost, e->stream));
    if (cage) TRY_OR_CLEAN(hipMemcpyAsync(cage, d_cage, n * 32, hipMemcpyDeviceToHost, e->stream));
    TRY_OR_CLEAN(hipStreamSynchronize(e->stream));
#undef TRY_OR_CLEAN
    cleanup();
    return DDGI_OK;
}

int ddgi_render_device(ddgi_handle e, const ddgi_camera* cam, const ddgi_render_settings* st, uint32_t* d_rgba8, float* d_rgb_f32)
{
    if (!e || !cam || !st || !d_rgba8) return fail(DDGI_ERR_INVALID_ARGUMENT, "null handle/camera/settings/output");
    if (st->scene < 0 || st->scene > 3) return fail(DDGI_ERR_INVALID_ARGUMENT, "scene %d not in {0,1,2,3}", st->scene);
    if (st->screen_width < 1 || st->screen_height < 1 || 1ll * st->screen_width * st->screen_height > 0x7fffffffll)
        return fail(DDGI_ERR_INVALID_ARGUMENT, "bad image size %d x %d", st->screen_width, st->screen_height);
    if (st->camera_mode != 0 && st->camera_mode != 1) return fail(DDGI_ERR_UNSUPPORTED, "camera_mode %d (only 0 pinhole, 1 ortho)", st->camera_mode);
    HIP_TRY(hipSetDevice(e->device));
    const int scene = st->scene;
    if (int rc = ensure_scene(e, scene)) return rc;
    if (int rc = ensure_noise(e)) return rc;
    RenderArgs r{};
    r.trace.grid = make_grid(e);
    r.trace.scene = e->dev_scene[scene].k;
    r.trace.scene_id = scene;
    r.trace.max_bounces = st->max_bounces;
    r.trace.nl = e->n_lights[scene];
    if (e->mode == DDGI_MODE_DDGI) animate_lights(scene, st->time, e->lights[scene], r.trace.nl, r.trace.lights);
    else
        for (int i = 0; i < r.trace.nl; ++i) r.trace.lights[i] = e->lights[scene][i];
    r.trace.noise = e->noise;
    for (int i = 0; i < 16; ++i) r.cam_matrix[i] = cam->matrix[i];
    for (int i = 0; i < 4; ++i) r.cam_params[i] = cam->params[i];
    const float half = 0.5f * cam->params[1];
    r.pinhole_w = 1.0f / (pm::sinf_pinned(half) / pm::cosf_pinned(half));  // P6: tan := sin / cos
    r.camera_mode = st->camera_mode;
    r.render_mode = st->render_mode;
    r.width = st->screen_width;
    r.height = st->screen_height;
    if (e->mode == DDGI_MODE_DDGI)
    {
        r.irradiance = static_cast<const float*>(e->tex[0]);
        r.depth = static_cast<const float*>(e->tex[1]);
    }
    else
        r.albedo = static_cast<const uint32_t*>(e->tex[0]);
    r.rgba8 = d_rgba8;
    r.rgb_f32 = d_rgb_f32;
    HIP_TRY(launch_render_primary(r, e->stream));
    return DDGI_OK;
}

int ddgi_render(ddgi_handle e, const ddgi_camera* cam, const ddgi_render_settings* st, uint8_t* rgba8, float* rgb_f32)
{
    if (!e || !cam || !st || !rgba8) return fail(DDGI_ERR_INVALID_ARGUMENT, "null handle/camera/settings/output");
    HIP_TRY(hipSetDevice(e->device));
    const size_t n = static_cast<size_t>(st->screen_width > 0 ? st->screen_width : 0) * (st->screen_height > 0 ? st->screen_height : 0);
    uint32_t* d_img = nullptr;
    float* d_f = nullptr;
    if (n == 0) return fail(DDGI_ERR_INVALID_ARGUMENT, "bad image size");
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_img), n * 4));
    if (rgb_f32 && hipMalloc(reinterpret_cast<void**>(&d_f), n * 12) != hipSuccess)
    {
        (void)hipFree(d_img);
        return fail(DDGI_ERR_OUT_OF_MEMORY, "hipMalloc failed");
    }
    int rc = ddgi_render_device(e, cam, st, d_img, d_f);
    hipError_t he = hipSuccess;
    if (rc == DDGI_OK) he = hipMemcpyAsync(rgba8, d_img, n * 4, hipMemcpyDeviceToHost, e->stream);
    if (rc == DDGI_OK && he == hipSuccess && rgb_f32) he = hipMemcpyAsync(rgb_f32, d_f, n * 12, hipMemcpyDeviceToHost, e->stream);
    if (rc == DDGI_OK && he == hipSuccess) he = hipStreamSynchronize(e->stream);
    (void)hipFree(d_img);
    if (d_f) (void)hipFree(d_f);
    if (rc != DDGI_OK) return rc;
    if (he != hipSuccess) return fail(DDGI_ERR_HIP, "render readback failed: %s", hipGetErrorString(he));
    return DDGI_OK;
}

int ddgi_set_stream(ddgi_handle e, void* hip_stream)
{
    if (!e) return fail(DDGI_ERR_INVALID_ARGUMENT, "null handle");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipStreamSynchronize(e->stream));
    if (e->own_stream) (void)hipStreamDestroy(e->stream);
    e->own_stream = false;
    e->stream = static_cast<hipStream_t>(hip_stream);
    return DDGI_OK;
}

int ddgi_device_textures(ddgi_handle e, void** tex0, size_t* tex0_bytes, void** tex1, size_t* tex1_bytes,
                         size_t* slab_offset0, size_t* slab_bytes0, size_t* slab_offset1, size_t* slab_bytes1)
{
    if (!e) return fail(DDGI_ERR_INVALID_ARGUMENT, "null handle");
    if (tex0) *tex0 = e->tex[0];
    if (tex1) *tex1 = e->tex[1];
    if (tex0_bytes) *tex0_bytes = e->tex_bytes[0];
    if (tex1_bytes) *tex1_bytes = e->tex_bytes[1];
    const size_t sb0 = e->tex_bytes[0] / e->world, sb1 = e->tex_bytes[1] / e->world;
    if (slab_offset0) *slab_offset0 = sb0 * e->rank;
    if (slab_bytes0) *slab_bytes0 = sb0;
    if (slab_offset1) *slab_offset1 = sb1 * e->rank;
    if (slab_bytes1) *slab_bytes1 = sb1;
    return DDGI_OK;
}

int ddgi_bind_textures(ddgi_handle e, void* tex0, void* tex1)
{
    if (!e) return fail(DDGI_ERR_INVALID_ARGUMENT, "null handle");
    if ((tex0 == nullptr) != (tex1 == nullptr)) return fail(DDGI_ERR_INVALID_ARGUMENT, "bind both textures or neither");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipStreamSynchronize(e->stream));
    e->tex[0] = tex0 ? tex0 : e->own_tex[0];
    e->tex[1] = tex1 ? tex1 : e->own_tex[1];
    return DDGI_OK;
}

int ddgi_texture_size(const ddgi_irradiance_field* f, int* width, int* height)
{
    if (!f) return fail(DDGI_ERR_INVALID_ARGUMENT, "null field");
    if (width) *width = f->probe_count[0] * f->probe_count[2] * f->sqrt_rays_per_probe;  // rvpt.cpp:873
    if (height) *height = f->probe_count[1] * f->sqrt_rays_per_probe;                    // rvpt.cpp:874
    return DDGI_OK;
}

int ddgi_probe_tile_origin(const ddgi_irradiance_field* f, int probe_index, int* x, int* y)
{
    if (!f) return fail(DDGI_ERR_INVALID_ARGUMENT, "null field");
    const int w = f->probe_count[0] * f->probe_count[2];
    if (w <= 0 || probe_index < 0 || probe_index >= w * f->probe_count[1])
        return fail(DDGI_ERR_INVALID_ARGUMENT, "probe index %d out of range", probe_index);
    const int yp = probe_index / w;  // probe_pass.comp:139-145
    if (x) *x = (probe_index - yp * w) * f->sqrt_rays_per_probe;
    if (y) *y = yp * f->sqrt_rays_per_probe;
    return DDGI_OK;
}

int ddgi_generate_probe_rays_host(const ddgi_irradiance_field* f, uint32_t seed, int skip_calls, ddgi_probe_ray* rays, size_t n)
{
    if (!f || !rays) return fail(DDGI_ERR_INVALID_ARGUMENT, "null field/rays");
    ddgi_render_settings st{};
    if (int rc = validate_config(f, &st, 1)) return rc;
    const size_t expect = static_cast<size_t>(f->probe_count[0]) * f->probe_count[1] * f->probe_count[2] *
                          f->sqrt_rays_per_probe * f->sqrt_rays_per_probe;
    if (n != expect) return fail(DDGI_ERR_INVALID_ARGUMENT, "expected room for %zu rays, got %zu", expect, n);
    GlibcRand rng;
    rng.seed(seed);
    // each generate_probe_rays() call draws 2*s*s values (rvpt.cpp:1161-1162)
    for (long long i = 0; i < 2ll * skip_calls * f->sqrt_rays_per_probe * f->sqrt_rays_per_probe; ++i) (void)rng.next();
    std::vector<ddgi_probe_ray> tmp;
    generate_probe_rays(*f, rng, tmp);
    std::memcpy(rays, tmp.data(), tmp.size() * sizeof(ddgi_probe_ray));
    return DDGI_OK;
}

// ---- baked scenes on disk, user scenes -------------------------------------------
// File: "DDGIVOX1" | int32 version (1) | int32 source scene (-1 = user made) | int32 lo[3] | int32 dim[3] |
//       uint32 noise id | dim.x*dim.y*dim.z bytes of block types, x fastest, then y, then z.
// noise id = bits of the pinned noise2D(1,1): a bake of the cave depends on the pinned sine
// (its floor is an fbm), so a file made with another arithmetic is refused.

static uint32_t noise_id()
{
    const float v = noise2D(1.0f, 1.0f);
    uint32_t u;
    std::memcpy(&u, &v, 4);
    return u;
}

static int fill_user_scene(ddgi_engine* e, const int lo[3], const int dim[3], const uint8_t* types)
{
    e->aq_march = 0;  // the trace kernel's wave split is measured again for the new scene
    for (int a = 0; a < 3; ++a)
        if (dim[a] < 1 || dim[a] > 4096 || lo[a] < -(1 << 20) || lo[a] > (1 << 20)) return fail(DDGI_ERR_INVALID_ARGUMENT, "bad scene box");
    const size_t n = static_cast<size_t>(dim[0]) * dim[1] * dim[2];
    // the kernels linearise voxel ids as z*nxy + y*nx + x in float / 24-bit integer arithmetic
    {
        long long bound = 0;
        const long long pitch[3] = {1, dim[0], 1ll * dim[0] * dim[1]};
        for (int a = 0; a < 3; ++a)
        {
            const long long m = std::max(std::llabs(static_cast<long long>(lo[a])), std::llabs(static_cast<long long>(lo[a]) + dim[a] - 1));
            bound += m * pitch[a];
        }
        if (n > (size_t(1) << 23) || bound >= (1ll << 23))
            return fail(DDGI_ERR_UNSUPPORTED, "scene box too large or too far from the origin: |z*nx*ny + y*nx + x| must stay below 2^23");
    }
    for (size_t i = 0; i < n; ++i)
        if (types[i] > 13) return fail(DDGI_ERR_INVALID_ARGUMENT, "voxel %zu: block type %d not in 0..13", i, types[i]);
    SceneBake b;
    b.scene = 3;
    for (int a = 0; a < 3; ++a) b.lo[a] = lo[a], b.dim[a] = dim[a], b.hi[a] = lo[a] + dim[a] - 1;
    b.types.assign(types, types + n);
    b.bits.clear();
    unsigned fe = 0;  // which border layers are entirely empty (march_escaped)
    for (int axis = 0; axis < 3; ++axis)
        for (int side = 0; side < 2; ++side)
        {
            bool empty = true;
            int c[3];
            const int a1 = (axis + 1) % 3, a2 = (axis + 2) % 3;
            c[axis] = side ? b.hi[axis] : b.lo[axis];
            for (c[a1] = b.lo[a1]; c[a1] <= b.hi[a1] && empty; ++c[a1])
                for (c[a2] = b.lo[a2]; c[a2] <= b.hi[a2]; ++c[a2])
                    if (b.block_at(c[0], c[1], c[2]) > 0)
                    {
                        empty = false;
                        break;
                    }
            if (empty) fe |= 1u << (2 * axis + side);
        }
    b.face_empty = fe;
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipStreamSynchronize(e->stream));
    ddgi_engine::DevScene& d = e->dev_scene[3];
    if (d.bits) (void)hipFree(d.bits);
    if (d.types) (void)hipFree(d.types);
    d = ddgi_engine::DevScene{};
    e->user_scene = std::move(b);
    return DDGI_OK;
}

int ddgi_scene_set_grid(ddgi_handle e, const int32_t lo[3], const int32_t dim[3], const uint8_t* types)
{
    if (!e || !lo || !dim || !types) return fail(DDGI_ERR_INVALID_ARGUMENT, "null argument");
    const int l[3] = {lo[0], lo[1], lo[2]}, d[3] = {dim[0], dim[1], dim[2]};
    return fill_user_scene(e, l, d, types);
}

int ddgi_scene_save(int scene, const char* path)
{
    if (scene < 0 || scene > 2 || !path) return fail(DDGI_ERR_INVALID_ARGUMENT, "scene %d not in {0,1,2} or null path", scene);
    const SceneBake& b = baked_scene(scene);
    FILE* fh = std::fopen(path, "wb");
    if (!fh) return fail(DDGI_ERR_INVALID_ARGUMENT, "cannot open %s for writing", path);
    const int32_t hdr[8] = {1, scene, b.lo[0], b.lo[1], b.lo[2], b.dim[0], b.dim[1], b.dim[2]};
    const uint32_t nid = noise_id();
    bool ok = std::fwrite("DDGIVOX1", 1, 8, fh) == 8 && std::fwrite(hdr, 4, 8, fh) == 8 && std::fwrite(&nid, 4, 1, fh) == 1 &&
              std::fwrite(b.types.data(), 1, b.types.size(), fh) == b.types.size();
    ok = (std::fclose(fh) == 0) && ok;
    return ok ? DDGI_OK : fail(DDGI_ERR_INVALID_ARGUMENT, "short write to %s", path);
}

int ddgi_scene_load(ddgi_handle e, const char* path)
{
    if (!e || !path) return fail(DDGI_ERR_INVALID_ARGUMENT, "null handle/path");
    FILE* fh = std::fopen(path, "rb");
    if (!fh) return fail(DDGI_ERR_INVALID_ARGUMENT, "cannot open %s", path);
    char magic[8];
    int32_t hdr[8];
    uint32_t nid = 0;
    std::vector<uint8_t> types;
    int rc = DDGI_OK;
    if (std::fread(magic, 1, 8, fh) != 8 || std::memcmp(magic, "DDGIVOX1", 8) != 0 || std::fread(hdr, 4, 8, fh) != 8 || std::fread(&nid, 4, 1, fh) != 1 ||
        hdr[0] != 1)
        rc = fail(DDGI_ERR_INVALID_ARGUMENT, "%s is not a DDGIVOX1 version-1 file", path);
    else if (hdr[1] == 0 && nid != noise_id())
        rc = fail(DDGI_ERR_INVALID_ARGUMENT, "%s was baked with a different noise arithmetic (id %08x, this build %08x)", path, nid, noise_id());
    else if (hdr[5] < 1 || hdr[6] < 1 || hdr[7] < 1 || 1ll * hdr[5] * hdr[6] * hdr[7] > (1ll << 23))
        rc = fail(DDGI_ERR_INVALID_ARGUMENT, "%s: bad dimensions", path);
    else
    {
        types.resize(static_cast<size_t>(hdr[5]) * hdr[6] * hdr[7]);
        if (std::fread(types.data(), 1, types.size(), fh) != types.size()) rc = fail(DDGI_ERR_INVALID_ARGUMENT, "%s is truncated", path);
    }
    std::fclose(fh);
    if (rc) return rc;
    const int lo[3] = {hdr[2], hdr[3], hdr[4]}, dim[3] = {hdr[5], hdr[6], hdr[7]};
    return fill_user_scene(e, lo, dim, types.data());
}

int ddgi_scene_block_at(int scene, int x, int y, int z)
{
    if (scene < 0 || scene > 2) return 0;
    return baked_scene(scene).block_at(x, y, z);
}

float ddgi_pinned_sinf(float x) { return pm::sinf_pinned(x); }
float ddgi_pinned_cosf(float x) { return pm::cosf_pinned(x); }
float ddgi_pinned_acosf(float x) { return pm::acosf_pinned(x); }

}  // extern "C"
