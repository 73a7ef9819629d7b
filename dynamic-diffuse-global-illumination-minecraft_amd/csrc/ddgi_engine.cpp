// ddgi_engine.cpp — the C ABI of include/ddgi_probe.h: handle, device memory, launches.
// There is no CPU compute path in this library: every compute entry point needs a gfx950 device.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <thread>
#include <cstdarg>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <set>
#include <string>
#include <utility>
#include <vector>

#include "ddgi_engine.h"
#include "ddgi_host.h"
#include "ddgi_pinned_math.h"
#include "ddgi_scene.h"
#include "ddgi_types.h"

namespace ddgi {
hipError_t launch_probe_trace_ref(const TraceArgs& args, int grid_blocks, hipStream_t stream);
hipError_t launch_probe_sample_ref(const SampleArgs& args, hipStream_t stream);
hipError_t trace_kernel_occupancy(int* blocks_per_cu, size_t lds_bytes);
int wf_pool_size(int nwords, bool multi_light, size_t lds_limit, int threads, int max_pool);
hipError_t launch_probe_trace_wf(const TraceArgs& args, int threads, int pool, int grid_blocks, uint32_t* work_counter, hipStream_t stream);
hipError_t launch_probe_blend(const BlendArgs& args, int num_cus, hipStream_t stream);
int aq_pool_size(int nwords, size_t lds_limit);
int aq_threads();
int aq_wgs_per_cu();
int aq_pool_usual();
int aq_pool_size_fast(int nwords_skip, size_t lds_limit, bool plain);
hipError_t launch_probe_trace_aq(const TraceArgs& args, int pool, int grid_blocks, int march_waves, const AqChain& chain, uint32_t* status, hipStream_t stream);
size_t blend_weights_floats(int n);
hipError_t launch_blend_weights(const BlendArgs& args, hipStream_t stream);
hipError_t launch_carry_tiles(void* dst, const void* src, const int32_t* map, uint32_t n_probes, uint32_t words_per_tile, hipStream_t stream);
hipError_t launch_probe_sample_ddgi(const SampleArgs& args, hipStream_t stream);
hipError_t launch_render_primary(const RenderArgs& args, hipStream_t stream);
size_t sample_group_scratch_words(uint32_t n, uint32_t n_probes);
hipError_t launch_sample_box_filter(const GridK& grid, const uint32_t* albedo, float4* box, int num_cus, hipStream_t stream);
hipError_t launch_sample_grouping(const GridK& grid, const float* pos, uint32_t n, uint32_t* scratch, const uint32_t** perm_out, const uint32_t** perm_off_out, hipStream_t stream);
hipError_t launch_light_visibility(const SceneK& scene, const float light_pos[3], const int32_t* list, int n_list, uint8_t* out, uint32_t* out_occ, hipStream_t stream);

hipError_t ensure_dynamic_lds(const void* kernel, int bytes)
{
    static std::mutex mu;
    static std::set<std::pair<int, const void*>> done;  // (device ordinal, kernel)
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(mu);
    if (done.count({dev, kernel})) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) done.insert({dev, kernel});
    return e;
}
}  // namespace ddgi

using namespace ddgi;
#define make_grid ddgi_make_grid
#define texture_bytes ddgi_texture_bytes
#define alloc_texture_pair ddgi_alloc_texture_pair

struct TuningKey
{
    const char* name;
    int Tuning::*field;
    const char* env;
};
static const TuningKey kTuningKeys[] = {
    {"trace_kernel", &Tuning::trace_kernel, nullptr},  // env DDGI_TRACE_KERNEL takes names, see read_env_tuning
    {"march_waves", &Tuning::march_waves, "DDGI_AQ_MARCH"},
    {"autotune", &Tuning::autotune, "DDGI_AUTOTUNE"},
    {"blend_kernel", &Tuning::blend_kernel, nullptr},
    {"blend_merge", &Tuning::blend_merge, "DDGI_BLEND_MERGE"},
    {"aq_pool", &Tuning::aq_pool, "DDGI_AQ_POOL"},
    {"wf_pool", &Tuning::wf_pool, "DDGI_WF_POOL"},
    {"wf_maxpool", &Tuning::wf_maxpool, "DDGI_WF_MAXPOOL"},
    {"wf_threads", &Tuning::wf_threads, "DDGI_WF_THREADS"},
    {"wf_fetch", &Tuning::wf_fetch, "DDGI_WF_FETCH"},
    {"wf_tail", &Tuning::wf_tail, "DDGI_WF_TAIL"},
    {"wf_chunk", &Tuning::wf_chunk, "DDGI_WF_CHUNK"},
    {"wf_drain", &Tuning::wf_drain, "DDGI_WF_DRAIN"},
    {"wait_threshold", &Tuning::wait_threshold, "DDGI_WAIT_THRESHOLD"},
    {"fast_march", &Tuning::fast_march, "DDGI_FAST_MARCH"},
    {"light_vis", &Tuning::light_vis, "DDGI_LIGHT_VIS"},
    {"sample_group", &Tuning::sample_group, "DDGI_SAMPLE_GROUP"},
    {"sample_box", &Tuning::sample_box, "DDGI_SAMPLE_BOX"},
    {"noise_lut", &Tuning::noise_lut, nullptr},
    {"lut_off", &Tuning::lut_off, "DDGI_LUT_OFF"},
    {"timing", &Tuning::timing, "DDGI_TIMING"},
    {"frames_in_flight", &Tuning::frames_in_flight, "DDGI_FRAMES_IN_FLIGHT"},
    {"reserve_cus", &Tuning::reserve_cus, "DDGI_RESERVE_CUS"},
    {"prep_stream", &Tuning::prep_stream, "DDGI_PREP_STREAM"},
    {"wait_timeout_ms", &Tuning::wait_timeout_ms, "DDGI_WAIT_TIMEOUT_MS"},
    {"verbose", &Tuning::verbose, "DDGI_VERBOSE"},
#ifdef DDGI_PROFILING
    {"ablate", &Tuning::ablate, "DDGI_ABLATE"},
#endif
};

static Tuning read_env_tuning()
{
    Tuning t;
    for (const TuningKey& k : kTuningKeys)
        if (k.env)
            if (const char* v = std::getenv(k.env)) t.*(k.field) = std::atoi(v);
    if (const char* v = std::getenv("DDGI_TRACE_KERNEL"))
        t.trace_kernel = !std::strcmp(v, "rounds") ? 1 : !std::strcmp(v, "lane") ? 2 : !std::strcmp(v, "queues") ? 3 : 0;
    if (const char* v = std::getenv("DDGI_BLEND_KERNEL")) t.blend_kernel = std::strcmp(v, "division") == 0 ? 2 : 1;
    if (std::getenv("DDGI_NO_NOISE_LUT")) t.noise_lut = 0;
    return t;
}

static_assert(sizeof(ddgi_irradiance_field) == 48, "IrradianceField must be 48 bytes (rvpt.h:82-90)");
static_assert(sizeof(ddgi_render_settings) == 32, "RenderSettings must be 32 bytes (rvpt.h:70-80)");
static_assert(sizeof(ddgi_probe_ray) == 48, "ProbeRay must be 48 bytes (probe.h:5-19)");
static_assert(offsetof(ddgi_irradiance_field, field_origin) == 32, "std140 layout");

// ---- error reporting ---------------------------------------------------------------------------------

static thread_local std::string g_last_error;

int ddgi_fail(int code, const char* fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

GridK ddgi_make_grid(const ddgi_engine* e)
{
    GridK g;
    g.cx = e->field.probe_count[0];
    g.cy = e->field.probe_count[1];
    g.cz = e->field.probe_count[2];
    g.sx = e->tile[0] > 0 ? e->tile[0] : e->field.sqrt_rays_per_probe;
    g.sy = e->tile[1] > 0 ? e->tile[1] : e->field.sqrt_rays_per_probe;
    g.n = g.sx * g.sy;
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

static int validate_tile(const ddgi_irradiance_field* f, int tx, int ty)
{
    if (tx < 1 || ty < 1 || tx > 4096 || ty > 4096) return fail(DDGI_ERR_INVALID_ARGUMENT, "ray tile %d x %d not in [1,4096]^2", tx, ty);
    const unsigned long long rays = 1ull * f->probe_count[0] * f->probe_count[1] * f->probe_count[2] * tx * ty;
    if (rays >= (1ull << 32)) return fail(DDGI_ERR_UNSUPPORTED, "more than 2^32 probe rays: the reference's uint RNG seed wraps");
    return DDGI_OK;
}

constexpr int kMaxDdgiRays = 4096;  // DDGI mode: rays per probe (the cross-check blend kernel keeps 28 B per ray in LDS)

static int check_kernel_status(ddgi_engine* e);
static int peers_done_before_host_copies(ddgi_engine* e);
static void free_vis_tables(ddgi_engine::DevScene& d);

// Byte sizes of the two probe textures of a configuration.
void ddgi_texture_bytes(int mode, const ddgi_irradiance_field& f, int rays_per_probe, size_t bytes[2])
{
    const size_t probes = static_cast<size_t>(f.probe_count[0]) * f.probe_count[1] * f.probe_count[2];
    if (mode == DDGI_MODE_DDGI)
    {
        bytes[0] = probes * 8 * 8 * 4 * sizeof(float);    // irradiance tiles
        bytes[1] = probes * 16 * 16 * 2 * sizeof(float);  // depth-moment tiles
    }
    else
        bytes[0] = bytes[1] = probes * static_cast<size_t>(rays_per_probe) * 4;
}

// One launch works on at most kChainRays rays: what a chain hides is a constant per launch (the drain, ~0.25 ms), and launches of
// a quarter of a second run SLOWER than their parts — measured in DDGI mode on one MI355X (tools/ddgi_chain_probe.py,
// profiles/r05_ddgi_chain_by_grid.txt): 64x32x64 probes x 256 rays, 8 updates per launch: 17.77 -> 17.53 ms per update; 64x64x64, 8 per
// launch (257 ms): 31.8 -> 33.0; 128x64x64, 4 per launch (268 ms): 63.2 -> 68.8; C5, 2 per launch (262 ms): 128.2 -> 134.0.
constexpr size_t kChainRays = static_cast<size_t>(64) << 20;
static int chain_len_for_rays(int len, size_t rays)
{
    while (len > 1 && rays * static_cast<size_t>(len) > kChainRays) len /= 2;
    return len;
}

// REF mode: updates one launch of the queue kernel may work on (tuning "frames_in_flight") = texture pairs a group of updates takes,
// on the handle's own textures only — a host that holds pointers to a pair (ddgi_bind_textures, ddgi_device_textures) expects the
// handle to stay on it.  The pair a ray writes travels in the top three bits of its texel index (ddgi_trace_wf.hip: kDstPairShift).
// (DDGI mode's trace writes ray records, not textures: its group is the ring of record buffers, rec_ring_len below.)
static int chain_len_for(const ddgi_engine* e, size_t albedo_bytes, bool pipelined)
{
    if (e->mode != DDGI_MODE_REF || e->pin_pair || e->caller_tex) return 1;
    if (albedo_bytes / 4 >= (static_cast<size_t>(1) << 29)) return 1;
    int len = std::min(kAqChainMax, std::max(1, e->tuning.frames_in_flight));
    // What a longer chain hides is a constant per launch (the drain: ~0.25 ms), worth nothing on updates of tens of milliseconds —
    // and a ring of 8 (16 under the pipelined exchange) pairs of a large grid is memory better left to the host: rings beyond 2 GiB
    // are halved down to two pairs.  (A function of the configuration only: every rank of a sharded grid comes to the same length.)
    // (the pipelined exchange doubles the ring: counted)
    while (len > 2 && albedo_bytes * 2 * static_cast<size_t>(len) * (pipelined ? 2 : 1) > (static_cast<size_t>(2) << 30)) len /= 2;
    len = chain_len_for_rays(len, albedo_bytes / 4);  // (REF: one texel per ray)
    return len;
}
int ddgi_chain_len(const ddgi_engine* e) { return chain_len_for(e, e->tex_bytes[0], e->xch.pipelined); }

// DDGI mode: ray-record buffers the handle keeps = updates one launch may work on.  The blend of update k reads the records of
// update k behind that update's own launch, while a launch may already be tracing k + 1 ..: every update of a group has its own
// buffer (20 B per ray: C3 84 MB, C4 1.3 GB, C5 5.4 GB — the ring stays below 16 GiB of the 288).
static int rec_ring_len(const ddgi_engine* e, size_t rec_bytes, size_t local_rays)
{
    if (local_rays >= (static_cast<size_t>(1) << 29)) return 1;  // (the update a ray belongs to travels in the top three bits of its index)
    int len = std::min(kAqChainMax, std::max(1, e->tuning.frames_in_flight));
    while (len > 1 && rec_bytes * static_cast<size_t>(len) > (static_cast<size_t>(16) << 30)) len /= 2;
    return chain_len_for_rays(len, local_rays);
}
int ddgi_group_len(const ddgi_engine* e)
{
    if (e->mode == DDGI_MODE_DDGI) return std::max(1, e->nrec);
    if (e->caller_tex || (e->pin_pair && !e->xch.pipelined)) return 1;
    return std::max(1, std::min(ddgi_chain_len(e), e->np));
}

// Pairs the handle's ring should hold: one per update a launch may work on; twice that (at least two) when the multi-GPU
// exchange is pipelined — the all-gathers of one group's pairs run while the next group's are written.
int ddgi_pairs_wanted(const ddgi_engine* e, bool pipelined)
{
    const int len = chain_len_for(e, e->tex_bytes[0], pipelined);
    return pipelined ? std::max(2, 2 * len) : len;
}

// Allocates and zero-fills a ring of np texture pairs (one allocation per texture); on failure nothing is left allocated.
// (the reference leaves the images undefined until the first probe pass, rvpt.cpp:873-890; here they start zeroed)
int ddgi_alloc_texture_pair(ddgi_engine* e, const size_t bytes[2], int np, void* out[2])
{
    out[0] = out[1] = nullptr;
    for (int i = 0; i < 2; ++i)
    {
        hipError_t he = hipMalloc(&out[i], bytes[i] * static_cast<size_t>(np));
        if (he == hipSuccess) he = hipMemsetAsync(out[i], 0, bytes[i] * static_cast<size_t>(np), e->stream);
        if (he != hipSuccess)
        {
            for (int k = 0; k < 2; ++k)
                if (out[k]) (void)hipFree(out[k]);
            out[0] = out[1] = nullptr;
            return fail(he == hipErrorOutOfMemory ? DDGI_ERR_OUT_OF_MEMORY : DDGI_ERR_HIP, "probe texture allocation (%zu B) failed: %s", bytes[i], hipGetErrorString(he));
        }
    }
    return DDGI_OK;
}

// (Re)creates the handle's own textures for its current field / tile / mode, zeroed, and makes them current.
// On failure the previous textures stay in place.
static int alloc_textures(ddgi_engine* e)
{
    e->box_of = nullptr;
    ddgi_exchange_release(e);  // a new configuration: the exchange is set up again by ddgi_exchange_init
    size_t bytes[2];
    texture_bytes(e->mode, e->field, make_grid(e).n, bytes);
    e->caller_tex = false;
    const int np = chain_len_for(e, bytes[0], false);  // (the ring's length follows the NEW configuration; ddgi_exchange_release above: no exchange)
    void* fresh[2];
    if (int rc = alloc_texture_pair(e, bytes, np, fresh)) return rc;
    for (int i = 0; i < 2; ++i)
    {
        if (e->own_tex[i]) (void)hipFree(e->own_tex[i]);
        e->own_tex[i] = e->tex[i] = fresh[i];
        e->tex_prev[i] = nullptr;
        e->tex_bytes[i] = bytes[i];
    }
    e->np = np, e->pair_cur = 0, e->ring_k = 0, e->chain_break = true;
    e->frame = 0;
    return DDGI_OK;
}

// A ring of another length (tuning "frames_in_flight" changed, the pipelined exchange was switched on or off).  Blocks.  The
// current pair's contents become pair 0 of the new ring — what consumers read and what a DDGI blend mixes with next —,
// the other pairs start zeroed.  Only for the handle's own textures.
int ddgi_resize_ring(ddgi_engine* e, int np)
{
    if (e->caller_tex) return fail(DDGI_ERR_INVALID_ARGUMENT, "the handle is on caller-bound textures: unbind them first");
    if (np == e->np) return DDGI_OK;
    DDGI_TRY(ddgi_sync_stream(e, e->stream));
    void* fresh[2];
    if (int rc = alloc_texture_pair(e, e->tex_bytes, np, fresh)) return rc;
    hipError_t he = hipSuccess;
    for (int i = 0; i < 2 && he == hipSuccess; ++i) he = hipMemcpyAsync(fresh[i], e->tex[i], e->tex_bytes[i], hipMemcpyDeviceToDevice, e->stream);
    if (he == hipSuccess) he = hipStreamSynchronize(e->stream);
    if (he != hipSuccess)
    {
        (void)hipFree(fresh[0]);
        (void)hipFree(fresh[1]);
        return fail(DDGI_ERR_HIP, "moving the probe textures to a ring of %d pairs failed: %s", np, hipGetErrorString(he));
    }
    for (int i = 0; i < 2; ++i)
    {
        (void)hipFree(e->own_tex[i]);
        e->own_tex[i] = e->tex[i] = fresh[i];
        e->tex_prev[i] = nullptr;
    }
    e->np = np, e->pair_cur = 0, e->ring_k = 0, e->chain_break = true;
    e->box_of = nullptr;
    return DDGI_OK;
}

int ddgi_rebase_ring(ddgi_engine* e)
{
    if (e->caller_tex) return DDGI_OK;
    DDGI_TRY(ddgi_sync_stream(e, e->stream));
    if (e->pair_cur != 0)
    {
        for (int i = 0; i < 2; ++i) HIP_TRY(hipMemcpyAsync(e->own_tex[i], ddgi_pair_ptr(e, e->pair_cur, i), e->tex_bytes[i], hipMemcpyDeviceToDevice, e->stream));
        DDGI_TRY(ddgi_sync_stream(e, e->stream));
    }
    for (int i = 0; i < 2; ++i) e->tex[i] = e->own_tex[i], e->tex_prev[i] = nullptr;
    e->pair_cur = 0, e->ring_k = 0, e->chain_break = true;
    e->box_of = nullptr;
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
    // the fast march's skip field (2 bits per voxel, addressed like the bitmap); built with the scene, used only on request
    std::vector<uint32_t> skip;
    build_skip_field(b, shift, skip);
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d.skip), skip.size() * sizeof(uint32_t)));
    HIP_TRY(hipMemcpy(d.skip, skip.data(), skip.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    d.k.skip = d.skip;
    d.k.nwords_skip = static_cast<int>(skip.size());
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

// The memoised lattice tables are 70 MB of device memory and the same for every handle: ONE copy per device, shared by the
// handles of the process on that device and freed with the last of them (a process driving 8 sharded handles on one device — the
// p2p tests — used to hold 560 MB of them).
namespace {
struct NoiseDevice
{
    float* d[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    int refs = 0;
};
std::mutex g_noise_mu;
std::map<int, NoiseDevice> g_noise_dev;
}  // namespace

static int ensure_noise(ddgi_engine* e)
{
    if (e->noise.n2 || !e->tuning.noise_lut) return DDGI_OK;
    std::lock_guard<std::mutex> lock(g_noise_mu);
    NoiseDevice& nd = g_noise_dev[e->device];
    if (nd.refs == 0)
    {
        const NoiseLutHost& h = noise_lut_host();
        const std::vector<float>* src[5] = {&h.n2, &h.n1, &h.wp, &h.wall, &h.r1};
        for (int i = 0; i < 5; ++i)
        {
            hipError_t he = hipMalloc(reinterpret_cast<void**>(&nd.d[i]), src[i]->size() * sizeof(float));
            if (he == hipSuccess) he = hipMemcpy(nd.d[i], src[i]->data(), src[i]->size() * sizeof(float), hipMemcpyHostToDevice);
            if (he != hipSuccess)
            {
                for (auto& p : nd.d)
                {
                    if (p) (void)hipFree(p);
                    p = nullptr;
                }
                return fail(he == hipErrorOutOfMemory ? DDGI_ERR_OUT_OF_MEMORY : DDGI_ERR_HIP, "uploading the noise lattice tables failed: %s", hipGetErrorString(he));
            }
        }
    }
    nd.refs += 1;
    e->noise_shared = true;
    e->noise.n2 = nd.d[0];
    e->noise.n1 = nd.d[1];
    e->noise.wp = nd.d[2];
    e->noise.wall = nd.d[3];
    e->noise.r1 = nd.d[4];
    if (e->tuning.lut_off & 1) e->noise.wall = nullptr;  // profiling
    if (e->tuning.lut_off & 2) e->noise.r1 = nullptr;
    return DDGI_OK;
}

static void release_noise(ddgi_engine* e)
{
    if (!e->noise_shared) return;
    std::lock_guard<std::mutex> lock(g_noise_mu);
    auto it = g_noise_dev.find(e->device);
    if (it != g_noise_dev.end() && --it->second.refs == 0)
    {
        for (auto& p : it->second.d)
            if (p) (void)hipFree(p);
        g_noise_dev.erase(it);
    }
    e->noise_shared = false;
    e->noise = NoiseLut{};
}

// The host copy of the rays (what RVPT::probe_rays holds) is page-locked while it keeps its place: the per-frame upload of a host that does what the
// reference's does (rvpt.cpp:285: the whole buffer, every frame) then goes over PCIe by DMA straight from it instead of through the runtime's staging
// of pageable memory.  Registered lazily by ddgi_upload_probe_rays (201 MB on C3: tens of milliseconds, once), dropped before anything may move the vector.
static void unpin_host_rays(ddgi_engine* e)
{
    if (!e->host_rays_pinned) return;
    (void)hipHostUnregister(e->host_rays_pinned);
    e->host_rays_pinned = nullptr;
}

static void pin_host_rays(ddgi_engine* e)
{
    if (e->host_rays.empty() || e->host_rays_pinned == e->host_rays.data()) return;
    unpin_host_rays(e);
    if (hipHostRegister(e->host_rays.data(), e->host_rays.size() * sizeof(ddgi_probe_ray), hipHostRegisterDefault) == hipSuccess)
        e->host_rays_pinned = e->host_rays.data();
    else
        (void)hipGetLastError();  // (a locked-memory limit: the copies below still work, through the runtime's staging)
}

// fn(c) for every c in [0, n_chunks) on up to 16 host threads (each takes the next chunk nobody has taken); returns when all are done
template <class Fn>
static void parallel_chunks(size_t n_chunks, Fn fn)
{
    const size_t hw = std::max(1u, std::thread::hardware_concurrency());
    const size_t t = std::max<size_t>(1, std::min<size_t>({16, hw, n_chunks}));
    std::atomic<size_t> next{0};
    auto worker = [&] {
        for (size_t c = next.fetch_add(1); c < n_chunks; c = next.fetch_add(1)) fn(c);
    };
    std::vector<std::thread> pool;
    pool.reserve(t - 1);
    for (size_t k = 1; k < t; ++k) pool.emplace_back(worker);
    worker();
    for (auto& th : pool) th.join();
}

static int upload_local_rays(ddgi_engine* e)
{
    e->chain_break = true;  // new rays: the next update is not a continuation of the last (frames in flight)
    e->n_local_rays = 0;    // (until the copies below are through: a failed upload leaves a handle WITHOUT rays, not with rays its host copy does not match)
    const GridK g = make_grid(e);
    const size_t n = static_cast<size_t>(g.n);
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
    DDGI_TRY(ddgi_sync_stream(e, e->stream));
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
    e->tuning = read_env_tuning();
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
    if (e->stream) (void)ddgi_sync_stream(e, e->stream);  // (bounded while an exchange is attached: a peer that is gone must not keep the handle from being destroyed)
    ddgi_exchange_release(e);
    for (int i = 0; i < 2; ++i)
        if (e->own_tex[i]) (void)hipFree(e->own_tex[i]);
    unpin_host_rays(e);
    if (e->d_rays) (void)hipFree(e->d_rays);
    if (e->d_stats) (void)hipFree(e->d_stats);
    if (e->d_sample_scratch) (void)hipFree(e->d_sample_scratch);
    if (e->d_box) (void)hipFree(e->d_box);
    if (e->d_blend_w) (void)hipFree(e->d_blend_w);
    if (e->d_work) (void)hipFree(e->d_work);
    if (e->pub) (void)hipHostFree(e->pub);
    if (e->prep_stream2)
    {
        (void)hipStreamSynchronize(e->prep_stream2);
        (void)hipStreamDestroy(e->prep_stream2);
    }
    if (e->prep_stream)
    {
        (void)hipStreamSynchronize(e->prep_stream);
        (void)hipStreamDestroy(e->prep_stream);
        (void)hipEventDestroy(e->prep_after);
        (void)hipEventDestroy(e->prep_done);
        (void)hipEventDestroy(e->prep_w_done);
    }
    if (e->upd_host) (void)hipHostFree(e->upd_host);
    if (e->upd_dev) (void)hipFree(e->upd_dev);
    for (auto& m : e->milestone)
        if (m) (void)hipEventDestroy(m);
    if (e->d_wf_cold) (void)hipFree(e->d_wf_cold);
    if (e->d_wf_dir) (void)hipFree(e->d_wf_dir);
    if (e->d_radiance) (void)hipFree(e->d_radiance);
    release_noise(e);
    for (auto& d : e->dev_scene)
    {
        if (d.bits) (void)hipFree(d.bits);
        if (d.types) (void)hipFree(d.types);
        if (d.skip) (void)hipFree(d.skip);
        free_vis_tables(d);
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
    e->tex_ops_since_update = true;  // (ddgi_exchange: something besides the update may be using the textures on the stream)
    if (int rc = validate_config(field, settings, e->world)) return rc;
    HIP_TRY(hipSetDevice(e->device));
    DDGI_TRY(ddgi_sync_stream(e, e->stream));
    e->field = *field;
    e->tile[0] = e->tile[1] = 0;  // the new field's square tile; ddgi_set_ray_tile changes it
    e->settings = *settings;
    unpin_host_rays(e);
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
    e->tex_ops_since_update = true;  // (ddgi_exchange: something besides the update may be using the textures on the stream)
    if (!carry_over) return ddgi_configure(e, field, settings);
    if (int rc = validate_config(field, settings, e->world)) return rc;
    HIP_TRY(hipSetDevice(e->device));
    DDGI_TRY(ddgi_sync_stream(e, e->stream));
    const ddgi_irradiance_field old = e->field;
    // a tile can be carried over only if it has the same size: always in DDGI mode (8x8 / 16x16
    // octahedral tiles), in REF mode when the rays per probe did not change
    const GridK old_grid = make_grid(e);
    const bool same_tiles = e->mode == DDGI_MODE_DDGI || (old_grid.sx == field->sqrt_rays_per_probe && old_grid.sy == field->sqrt_rays_per_probe);
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
    // New textures first, into temporaries: the handle's state is committed only when everything worked.
    const size_t old_probes = static_cast<size_t>(old.probe_count[0]) * old.probe_count[1] * old.probe_count[2];
    const size_t old_words[2] = {e->tex_bytes[0] / 4 / old_probes, e->tex_bytes[1] / 4 / old_probes};
    size_t bytes[2];
    texture_bytes(e->mode, *field, field->sqrt_rays_per_probe * field->sqrt_rays_per_probe, bytes);
    void* fresh[2];
    const int np = chain_len_for(e, bytes[0], false);  // (the exchange, if any, is set up again afterwards: ddgi_exchange_release below)
    if (int rc = alloc_texture_pair(e, bytes, np, fresh)) return rc;
    if (carried > 0)
    {
        // (a pipelined exchange may still be writing the other ranks' slabs into the pair the carry reads)
        if (int rc = ddgi_exchange_wait_latest(e))
        {
            (void)hipFree(fresh[0]);
            (void)hipFree(fresh[1]);
            return rc;
        }
        int32_t* d_map = nullptr;
        hipError_t he = hipMalloc(reinterpret_cast<void**>(&d_map), map.size() * sizeof(int32_t));
        if (he == hipSuccess) he = hipMemcpyAsync(d_map, map.data(), map.size() * sizeof(int32_t), hipMemcpyHostToDevice, e->stream);
        for (int i = 0; i < 2 && he == hipSuccess; ++i)
            he = launch_carry_tiles(fresh[i], e->tex[i], d_map, static_cast<uint32_t>(map.size()), static_cast<uint32_t>(old_words[i]), e->stream);
        if (he == hipSuccess) he = hipStreamSynchronize(e->stream);
        if (d_map) (void)hipFree(d_map);
        if (he != hipSuccess)
        {
            (void)hipFree(fresh[0]);
            (void)hipFree(fresh[1]);
            return fail(he == hipErrorOutOfMemory ? DDGI_ERR_OUT_OF_MEMORY : DDGI_ERR_HIP, "carrying the probe tiles over failed: %s (the handle keeps its previous configuration)",
                        hipGetErrorString(he));
        }
    }
    // commit.  Caller-bound textures (ddgi_bind_textures) are unbound: the handle uses its own from here on.
    e->box_of = nullptr;
    ddgi_exchange_release(e);
    for (int i = 0; i < 2; ++i)
    {
        if (e->own_tex[i]) (void)hipFree(e->own_tex[i]);
        e->own_tex[i] = e->tex[i] = fresh[i];
        e->tex_prev[i] = nullptr;
        e->tex_bytes[i] = bytes[i];
    }
    e->caller_tex = false;
    e->np = np, e->pair_cur = 0, e->ring_k = 0, e->chain_break = true;
    e->field = *field;
    e->tile[0] = e->tile[1] = 0;
    e->settings = *settings;
    unpin_host_rays(e);
    e->host_rays.clear();
    e->n_local_rays = 0;
    e->updates = 0;  // (the DDGI frame sequence — ray rotation, RNG keys — goes on: e->frame is kept)
    return DDGI_OK;
}

int ddgi_set_mode(ddgi_handle e, int mode)
{
    if (!e) return fail(DDGI_ERR_INVALID_ARGUMENT, "null handle");
    e->tex_ops_since_update = true;  // (ddgi_exchange: something besides the update may be using the textures on the stream)
    if (mode != DDGI_MODE_REF && mode != DDGI_MODE_DDGI) return fail(DDGI_ERR_INVALID_ARGUMENT, "unknown mode %d", mode);
    if (mode == e->mode) return DDGI_OK;
    HIP_TRY(hipSetDevice(e->device));
    DDGI_TRY(ddgi_sync_stream(e, e->stream));
    if (e->caller_tex) return fail(DDGI_ERR_INVALID_ARGUMENT, "unbind caller textures before changing the mode");
    e->mode = mode;
    e->updates = 0;
    return alloc_textures(e);  // the two modes keep differently shaped textures; both start zeroed
}

int ddgi_set_ray_tile(ddgi_handle e, int tile_x, int tile_y)
{
    if (!e) return fail(DDGI_ERR_INVALID_ARGUMENT, "null handle");
    const int s = e->field.sqrt_rays_per_probe;
    if (tile_x == 0 && tile_y == 0) tile_x = tile_y = s;
    if (int rc = validate_tile(&e->field, tile_x, tile_y)) return rc;
    const GridK g = make_grid(e);
    if (g.sx == tile_x && g.sy == tile_y) return DDGI_OK;
    HIP_TRY(hipSetDevice(e->device));
    DDGI_TRY(ddgi_sync_stream(e, e->stream));
    if (e->caller_tex) return fail(DDGI_ERR_INVALID_ARGUMENT, "unbind caller textures before changing the ray tile");
    e->tile[0] = tile_x, e->tile[1] = tile_y;
    unpin_host_rays(e);
    e->host_rays.clear();
    e->n_local_rays = 0;
    e->updates = 0;
    return alloc_textures(e);  // REF: the texel tile follows the ray tile; both modes restart from zeroed textures
}

int ddgi_get_ray_tile(ddgi_handle e, int* tile_x, int* tile_y)
{
    if (!e) return fail(DDGI_ERR_INVALID_ARGUMENT, "null handle");
    const GridK g = make_grid(e);
    if (tile_x) *tile_x = g.sx;
    if (tile_y) *tile_y = g.sy;
    return DDGI_OK;
}

int ddgi_get_texture_size(ddgi_handle e, int* width, int* height)
{
    if (!e) return fail(DDGI_ERR_INVALID_ARGUMENT, "null handle");
    const GridK g = make_grid(e);
    if (width) *width = g.cx * g.cz * g.sx;
    if (height) *height = g.cy * g.sy;
    return DDGI_OK;
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
    const GridK g = make_grid(e);
    unpin_host_rays(e);  // (the generator may grow the vector)
    generate_probe_rays(e->field, g.sx, g.sy, e->rand, e->host_rays);
    return upload_local_rays(e);
}

int ddgi_upload_probe_rays(ddgi_handle e, const ddgi_probe_ray* rays, size_t n)
{
    if (!e || !rays) return fail(DDGI_ERR_INVALID_ARGUMENT, "null handle/rays");
    HIP_TRY(hipSetDevice(e->device));
    const GridK g = make_grid(e);
    const size_t probes = static_cast<size_t>(g.cx) * g.cy * g.cz;
    const size_t expect = probes * g.n;
    if (n != expect) return fail(DDGI_ERR_INVALID_ARGUMENT, "expected %zu rays (full grid), got %zu", expect, n);
    // The reference's host copies its WHOLE ray buffer to the device every frame (probe_buffer.copy_to, rvpt.cpp:285) — the same rays: it generates them
    // at start-up and after a reconfiguration only (main.cpp:47, rvpt.cpp:721-726).  So, per chunk of 64 Ki rays, on up to 16 host threads: a chunk whose bytes equal the handle's own host copy — which the
    // device holds — is left alone (checked when it came in, already where it belongs); any other is checked — the reference's shader trusts probe_info
    // blindly (Q13), an out-of-range tile would write outside the texture: rejected here, nothing changed —, copied into the host copy (page-locked: the DMA's
    // source) and sent.  An unchanged buffer costs two reads of it and touches neither the GPU nor the frames in flight (the next update goes on with its
    // predecessor's launch); C3's 201 MB, all new: 6 ms = 33 GB/s (15 ms before round 6: one thread, pageable memory).
    constexpr size_t kChunk = 65536;
    const size_t n_chunks = (n + kChunk - 1) / kChunk;
    const bool have_copy = e->host_rays.size() == n && e->n_local_rays > 0 && e->d_rays;  // (the device holds what the host copy holds)
    const ddgi_probe_ray* held = have_copy ? e->host_rays.data() : nullptr;
    std::vector<uint8_t> dirty(n_chunks, 1);
    std::atomic<size_t> first_bad{n};
    const float fp = static_cast<float>(probes), fx = static_cast<float>(g.sx), fy = static_cast<float>(g.sy);
    parallel_chunks(n_chunks, [&](size_t c) {
        const size_t a = c * kChunk, b = std::min(n, a + kChunk);
        if (held && std::memcmp(held + a, rays + a, (b - a) * sizeof(ddgi_probe_ray)) == 0)
        {
            dirty[c] = 0;
            return;
        }
        for (size_t i = a; i < b; ++i)
        {
            const float p = rays[i].probe_info[0], tx = rays[i].probe_info[1], ty = rays[i].probe_info[2];
            if (!(p >= 0.0f && p < fp && tx >= 0.0f && tx < fx && ty >= 0.0f && ty < fy))
            {
                size_t seen = first_bad.load();
                while (i < seen && !first_bad.compare_exchange_weak(seen, i)) {}
                return;
            }
        }
    });
    if (first_bad.load() < n)
    {
        const size_t i = first_bad.load();  // (the first one of its chunk; the lowest of those)
        return fail(DDGI_ERR_INVALID_ARGUMENT, "ray %zu: probe_info (%g,%g,%g) outside the grid", i, rays[i].probe_info[0], rays[i].probe_info[1], rays[i].probe_info[2]);
    }
    size_t n_dirty = 0;
    for (uint8_t d : dirty) n_dirty += d;
    if (n_dirty == 0) return DDGI_OK;  // the device holds these rays already
    if (e->host_rays.size() != n)
    {
        unpin_host_rays(e);
        e->host_rays.resize(n);
    }
    ddgi_probe_ray* own = e->host_rays.data();
    parallel_chunks(n_chunks, [&](size_t c) {
        const size_t a = c * kChunk, b = std::min(n, a + kChunk);
        if (dirty[c]) std::memcpy(own + a, rays + a, (b - a) * sizeof(ddgi_probe_ray));
    });
    pin_host_rays(e);
    if (!have_copy || e->world != 1) return upload_local_rays(e);
    // an unsharded handle's device order is the host's: only the runs of chunks that changed go over PCIe
    e->chain_break = true;  // new rays: the next update is not a continuation of the last (frames in flight)
    const uint32_t n_held = e->n_local_rays;
    e->n_local_rays = 0;    // (as in upload_local_rays: a copy that fails leaves the handle without rays)
    for (size_t c = 0; c < n_chunks;)
    {
        if (!dirty[c])
        {
            ++c;
            continue;
        }
        size_t c_end = c;
        while (c_end < n_chunks && dirty[c_end]) ++c_end;
        const size_t a = c * kChunk, b = std::min(n, c_end * kChunk);
        HIP_TRY(hipMemcpyAsync(reinterpret_cast<ddgi_probe_ray*>(e->d_rays) + a, own + a, (b - a) * sizeof(ddgi_probe_ray), hipMemcpyHostToDevice, e->stream));
        c = c_end;
    }
    DDGI_TRY(ddgi_sync_stream(e, e->stream));
    e->n_local_rays = n_held;
    return DDGI_OK;
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
    mix(static_cast<unsigned>(a.grid.sx)), mix(static_cast<unsigned>(a.grid.sy));
    for (int i = 0; i < 3; ++i)
    {
        unsigned u;
        std::memcpy(&u, &e->field.field_origin[i], 4);
        mix(u);
    }
    mix(static_cast<unsigned>(a.scene_id)), mix(static_cast<unsigned>(a.max_bounces)), mix(static_cast<unsigned>(a.nl));
    mix(ddgi_mode ? 1u : 0u), mix(static_cast<unsigned>(e->rank)), mix(static_cast<unsigned>(e->world)), mix(a.n_rays);
    mix(e->scene_epoch), mix(static_cast<unsigned>(a.fast_march));
    return h | 1ull;
}

static void free_vis_tables(ddgi_engine::DevScene& d)
{
    for (auto& set : d.vis_set)
    {
        for (auto& v : set.vis)
            if (v) (void)hipFree(v);
        if (set.occ) (void)hipFree(set.occ);
        set = ddgi_engine::DevScene::VisSet{};
    }
    if (d.vis_list) (void)hipFree(d.vis_list);
    d.vis_list = nullptr, d.n_vis_list = -1;
}

// ---- light-feeler classes (ddgi_visibility.hip) ------------------------------------------------------------------------
// Which feelers need no march: one table per light (the first kVisLights) and light position, in sets (ddgi_engine.h: VisSet).

static bool vis_set_holds(const ddgi_engine::DevScene::VisSet& set, const LightK* lights, int nl)
{
    for (int li = 0; li < nl && li < kVisLights; ++li)
        if (!set.vis[li] || !set.valid[li] || std::memcmp(set.light[li], lights[li].pos, sizeof(set.light[li])) != 0) return false;
    return true;
}

// A set that holds the tables of exactly these light positions, complete before launch `before_seq` starts (computed by kernels
// that stand in front of it in the stream); -1: none.  any_seq: computed at all (the caller's own launch stands behind them).
static int find_vis_set(const ddgi_engine* e, int scene, const LightK* lights, int nl, bool any_seq, uint32_t before_seq)
{
    const ddgi_engine::DevScene& d = e->dev_scene[scene];
    for (int k = 0; k < 2 * kAqChainMax; ++k)
        if (!d.vis_set[k].on_prep_stream && vis_set_holds(d.vis_set[k], lights, nl) && (any_seq || static_cast<int32_t>(d.vis_set[k].launch_seq - before_seq) <= 0)) return k;
    return -1;
}

// Makes set `k` hold the tables of these light positions: allocates what is missing, runs k_light_visibility for every light whose
// table is missing or stale (on the handle's stream: complete before the next launch starts).
// How many of each half's kAqChainMax sets of light-feeler tables a scene may fill: a set is 8 bytes per voxel and light plus light 0's lists of up to
// 512 MB, allocated on first use and kept until the scene changes — with animated lights all sixteen fill within a few frames (the cave: 16 x 29 MB;
// a 4 M-voxel user scene would come to 8.7 GB).  Each half gets 1 GiB; at least one set per half (a scene that large traces its moving lights' updates
// one per launch: no predictions, nothing wrong).
static int vis_sets_allowed(const SceneK& sk, int nl)
{
    const size_t n_vox = static_cast<size_t>(sk.hi[0] - sk.lo[0] + 1) * (sk.hi[1] - sk.lo[1] + 1) * (sk.hi[2] - sk.lo[2] + 1);
    const size_t occ_bytes = n_vox * 8 * kVisListMax * sizeof(uint32_t);
    const size_t set_bytes = n_vox * 8 * static_cast<size_t>(std::max(1, std::min(nl, static_cast<int>(kVisLights)))) + (occ_bytes <= (static_cast<size_t>(512) << 20) ? occ_bytes : 0);
    const size_t fit = (static_cast<size_t>(1) << 30) / std::max<size_t>(1, set_bytes);
    return static_cast<int>(std::max<size_t>(1, std::min<size_t>(fit, kAqChainMax)));
}

static int fill_vis_set(ddgi_engine* e, int scene, const SceneK& sk, const LightK* lights, int nl, int k, hipStream_t stream = nullptr)
{
    if (!stream) stream = e->stream;
    ddgi_engine::DevScene& d = e->dev_scene[scene];
    const int n_vox = (sk.hi[0] - sk.lo[0] + 1) * (sk.hi[1] - sk.lo[1] + 1) * (sk.hi[2] - sk.lo[2] + 1);
    if (d.n_vis_list < 0)
    {
        // the voxels a feeler can start in, once per scene: empty with an occupied face neighbour (clamped lookups: outside
        // the box the world is the extrusion of the border layer)
        const SceneBake& bk = scene == 3 ? e->user_scene : baked_scene(scene);
        std::vector<int32_t> list;
        for (int z = bk.lo[2]; z <= bk.hi[2]; ++z)
            for (int y = bk.lo[1]; y <= bk.hi[1]; ++y)
                for (int x = bk.lo[0]; x <= bk.hi[0]; ++x)
                {
                    if (bk.block_at(x, y, z) > 0) continue;
                    const int r = ((z - bk.lo[2]) * bk.dim[1] + (y - bk.lo[1])) * bk.dim[0] + (x - bk.lo[0]);
                    const int nb[6][3] = {{-1, 0, 0}, {1, 0, 0}, {0, -1, 0}, {0, 1, 0}, {0, 0, -1}, {0, 0, 1}};  // face = 2 axis + (+ side)
                    for (int fc = 0; fc < 6; ++fc)
                        if (bk.block_at(x + nb[fc][0], y + nb[fc][1], z + nb[fc][2]) > 0) list.push_back(r * 8 + fc);
                }
        if (!list.empty())
        {
            HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d.vis_list), list.size() * sizeof(int32_t)));
            HIP_TRY(hipMemcpy(d.vis_list, list.data(), list.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        }
        d.n_vis_list = static_cast<int>(list.size());
    }
    ddgi_engine::DevScene::VisSet& set = d.vis_set[k];
    for (int li = 0; li < nl && li < kVisLights; ++li)
    {
        if (!set.vis[li])
        {
            // one class byte per (voxel, face); light 0 also lists the occupied voxels of a kVisListed bundle (written and read only
            // for entries of that class: no initialisation; 128 B per voxel — a user scene of more than 4 M voxels goes without
            // lists, i.e. without the class).  Allocated into locals and committed together: a failure half way (a large user
            // scene) must not leave a table without its lists behind for the next update to trip over
            uint8_t* vis = nullptr;
            uint32_t* occ = nullptr;
            hipError_t he = hipMalloc(reinterpret_cast<void**>(&vis), static_cast<size_t>(n_vox) * 8);
            if (he == hipSuccess) he = hipMemsetAsync(vis, 0, static_cast<size_t>(n_vox) * 8, stream);
            const size_t occ_bytes = static_cast<size_t>(n_vox) * 8 * kVisListMax * sizeof(uint32_t);
            if (he == hipSuccess && li == 0 && !set.occ && occ_bytes <= (static_cast<size_t>(512) << 20)) he = hipMalloc(reinterpret_cast<void**>(&occ), occ_bytes);
            if (he != hipSuccess)
            {
                if (vis) (void)hipFree(vis);
                if (occ) (void)hipFree(occ);
                return fail(he == hipErrorOutOfMemory ? DDGI_ERR_OUT_OF_MEMORY : DDGI_ERR_HIP, "allocating the light-feeler classes failed: %s", hipGetErrorString(he));
            }
            set.vis[li] = vis;
            if (occ) set.occ = occ;
            set.valid[li] = false;
        }
        if (!set.valid[li] || std::memcmp(set.light[li], lights[li].pos, sizeof(set.light[li])) != 0)
        {
            // (lights 1 ..: classes only — no lists of occupied voxels: the event of a hit under several lights does not use them)
            HIP_TRY(launch_light_visibility(sk, lights[li].pos, d.vis_list, d.n_vis_list, set.vis[li], li == 0 ? set.occ : nullptr, stream));
            std::memcpy(set.light[li], lights[li].pos, sizeof(set.light[li]));
            set.valid[li] = true;
            set.launch_seq = e->launch_seq;  // (launches before this one stand IN FRONT of the table's kernel in the stream)
            set.on_prep_stream = stream != e->stream;  // (... once the handle's stream has waited for the preparation stream: ddgi_probe_update)
        }
    }
    return DDGI_OK;
}

// One probe update's trace launch, fully decided: arguments, kernel, pool, grid.  Built by plan_trace (which
// also makes sure every buffer the launch needs exists), used by ddgi_probe_update and ddgi_tune.
struct TracePlan
{
    TraceArgs a{};
    bool ddgi_mode = false;
    int pool = 0;            // ray pool of the wavefront kernels; 0: the ray-per-lane kernel
    bool use_async = false;  // k_probe_trace_aq (queues) rather than k_probe_trace_wf (rounds)
    bool fast = false;       // ... its tolerance-mode build (tuning "fast_march")
    int wf_threads = 1024;
    uint32_t grid = 0;
    size_t rec_rgb = 0;      // DDGI mode: floats in the rgb part of the ray-record buffer (the (d, d*d) part follows)
    unsigned long long key = 0;
};

static int plan_trace(ddgi_engine* e, TracePlan& p)
{
    const Tuning& tn = e->tuning;
    p.ddgi_mode = e->mode == DDGI_MODE_DDGI;
    if (!p.ddgi_mode && e->n_local_rays == 0) return fail(DDGI_ERR_NOT_READY, "ddgi_probe_update before any probe rays were generated/uploaded");
    const int scene = e->settings.scene;
    if (int rc = ensure_scene(e, scene)) return rc;
    if (int rc = ensure_noise(e)) return rc;

    TraceArgs& a = p.a;
    std::memset(&a, 0, sizeof a);  // (padding too: plan_hash compares launches byte by byte)
    a.grid = make_grid(e);
    a.scene = e->dev_scene[scene].k;
    a.scene_id = scene;
    a.max_bounces = e->settings.max_bounces;
    a.nl = e->n_lights[scene];
    for (int i = 0; i < a.nl; ++i) a.lights[i] = e->lights[scene][i];
    a.rays = e->d_rays;
    a.n_rays = e->n_local_rays;
    if (p.ddgi_mode && a.grid.n > kMaxDdgiRays)
        return fail(DDGI_ERR_UNSUPPORTED, "DDGI mode: %d rays per probe; at most %d are supported", a.grid.n, kMaxDdgiRays);
    if (p.ddgi_mode)
    {
        // rays are generated in the kernel; lights follow update_lights(time) (probe_pass.comp:217-251)
        const size_t local_rays = static_cast<size_t>(a.grid.cx) * a.grid.cy * a.grid.czl * a.grid.n;
        const uint32_t local_probes = static_cast<uint32_t>(a.grid.cx) * a.grid.cy * a.grid.czl;
        p.rec_rgb = rec_rgb_floats(local_probes, static_cast<uint32_t>(a.grid.n));
        const size_t rec_floats = p.rec_rgb + rec_dd_floats(local_probes, static_cast<uint32_t>(a.grid.n));
        int want = rec_ring_len(e, rec_floats * sizeof(float), local_rays);
        // (a ring that did not fit once is not asked for again on every update — each try would free, wait for the device and fail a
        //  multi-GB hipMalloc: the fallback to one buffer is remembered until the records' size changes)
        if (e->nrec_cap > 0 && rec_floats == e->rec_stride && e->d_radiance_rays == a.grid.n && want > e->nrec_cap) want = e->nrec_cap;
        if (!e->d_radiance || rec_floats != e->rec_stride || want != e->nrec || e->d_radiance_rays != a.grid.n)
        {
            // (re)allocated zeroed: the records of padding rays / padding probes are never written and must read as 0.  A ring of
            // `want` buffers (frames in flight); if that much memory is not to be had, one buffer and no continuation.
            if (e->d_radiance) (void)hipFree(e->d_radiance);  // (waits for the device: no launch is still writing records)
            e->d_radiance = nullptr;
            e->d_radiance_capacity = 0, e->rec_stride = 0, e->nrec = 1, e->nrec_cap = 0;
            hipError_t he = hipMalloc(reinterpret_cast<void**>(&e->d_radiance), rec_floats * sizeof(float) * static_cast<size_t>(want));
            if (he == hipErrorOutOfMemory && want > 1)
            {
                (void)hipGetLastError();
                want = 1;
                e->nrec_cap = 1;
                he = hipMalloc(reinterpret_cast<void**>(&e->d_radiance), rec_floats * sizeof(float));
            }
            HIP_TRY(he);
            HIP_TRY(hipMemsetAsync(e->d_radiance, 0, rec_floats * sizeof(float) * static_cast<size_t>(want), e->stream));
            e->d_radiance_capacity = rec_floats * static_cast<size_t>(want);
            e->d_radiance_rays = a.grid.n;
            e->rec_stride = rec_floats, e->nrec = want;
            e->chain_break = true;
        }
        animate_lights(scene, e->settings.time, e->lights[scene], a.nl, a.lights);
        a.ddgi = 1;
        a.frame_key = frame_key(e->frame);
        frame_rotation(e->frame, a.rot);
        a.rad_rgb = e->d_radiance;  // (buffer 0; ddgi_probe_update moves the update to the buffer of its number)
        a.rad_dd = e->d_radiance + p.rec_rgb;
        a.rays = nullptr;
        a.n_rays = static_cast<uint32_t>(local_rays);
    }
    // (the light-feeler classes — a.vis, a.vis_occ, a.vis_more — are assigned by assign_vis, once the caller knows whether the
    // update continues its predecessor)
    a.albedo = static_cast<uint32_t*>(e->tex[0]);
    a.distance = static_cast<uint32_t*>(e->tex[1]);
    a.wait_threshold = tn.wait_threshold;
    a.stats = e->d_stats;
    a.noise = e->noise;
    a.ablate = tn.ablate;  // always 0 in the release library (Tuning)
    a.wf_tail = tn.wf_tail, a.wf_fetch = tn.wf_fetch, a.wf_chunk = tn.wf_chunk, a.wf_drain = tn.wf_drain;

    // Kernel choice: the wavefront kernel (one persistent 1024-lane workgroup per CU, ray pool in
    // LDS) whenever its pool fits next to the scene bitmap; the ray-per-lane kernel otherwise
    // (or when tuning "trace_kernel" = 2 asks for it, e.g. to cross-check the two).
    const bool force_lane = tn.trace_kernel == 2;
    if (p.ddgi_mode && (force_lane || a.max_bounces < 1)) return fail(DDGI_ERR_UNSUPPORTED, "DDGI mode needs the wavefront trace kernel and max_bounces >= 1");
    p.wf_threads = tn.wf_threads == 512 ? 512 : 1024;
    const int wf_blocks_per_cu = 1024 / p.wf_threads;
    p.pool = (force_lane || a.max_bounces < 1) ? 0 : wf_pool_size(a.scene.nwords, a.nl > 1, 160 * 1024 / wf_blocks_per_cu, p.wf_threads, tn.wf_maxpool);
    if (tn.wf_pool > 0) p.pool = p.pool ? std::min(p.pool, std::max(p.wf_threads, tn.wf_pool / 64 * 64)) : 0;
    // the barrier-free queue kernel (k_probe_trace_aq) whenever its pool fits; "trace_kernel" = 1 asks for the
    // round-based k_probe_trace_wf (cross-check), which also serves the utilisation counters
    const bool force_rounds = tn.trace_kernel == 1 || (a.stats != nullptr && tn.trace_kernel != 3) || p.wf_threads != 1024;
    const size_t aq_lds = 160 * 1024 / static_cast<size_t>(aq_wgs_per_cu());
    p.use_async = p.pool > 0 && !force_rounds && aq_pool_size(a.scene.nwords, aq_lds) > 0;
    // "fast_march": the tolerance-mode march (ddgi_device.h: fast_march_step) — the queue kernel only, and only when its pool
    // fits next to the scene's skip field; otherwise the update runs the exact march ("fast_march_active" tells)
    p.fast = false;
    if (p.use_async && tn.fast_march && a.stats == nullptr && tn.ablate == 0)
    {
        const int fast_pool = aq_pool_size_fast(a.scene.nwords_skip, aq_lds, a.nl == 1);
        if (fast_pool > 0)
        {
            p.fast = true;
            p.pool = fast_pool;
            a.fast_march = 1;
        }
    }
    e->fast_march_active = p.fast;
    if (p.use_async && !p.fast)
    {
        p.pool = std::min(aq_pool_usual(), aq_pool_size(a.scene.nwords, aq_lds));  // what fits next to the cave's bitmap at 72 B per slot; more buys nothing (C3: 1024: 2.56 ms, 1152: 2.37, 1344: 2.11, 1472: 2.13)
        if (tn.aq_pool > 0) p.pool = std::min(aq_pool_size(a.scene.nwords, aq_lds), std::max(aq_wgs_per_cu() == 1 ? 1024 : 320, tn.aq_pool / 64 * 64));
    }
    if (p.ddgi_mode && p.pool <= 0) return fail(DDGI_ERR_UNSUPPORTED, "DDGI mode: the ray pool does not fit in LDS next to the scene bitmap");
    if (p.pool > 0)
    {
        if (!e->d_work)
        {
            // [1] kernel status, [3] the round kernel's counter, [8 .. 8 + kAqCounters) the queue kernel's ray counters: launch s uses [8 + s % kAqCounters] and zeroes
            // the one of launch s + kAqChainMax (ddgi_types.h: AqChain).  `pub`: the host's ring of published continuations, read by running launches.
            HIP_TRY(hipMalloc(reinterpret_cast<void**>(&e->d_work), (8 + kAqCounters) * sizeof(uint32_t)));
            HIP_TRY(hipMemsetAsync(e->d_work, 0, (8 + kAqCounters) * sizeof(uint32_t), e->stream));
            e->launch_seq = 0;
        }
        if (!e->pub)
        {
            HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&e->pub), kAqPubRing * sizeof(uint32_t), hipHostMallocMapped | hipHostMallocCoherent));
            std::memset(e->pub, 0, kAqPubRing * sizeof(uint32_t));
            HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void**>(&e->pub_dev), e->pub, 0));
            for (auto& m : e->milestone) HIP_TRY(hipEventCreateWithFlags(&m, hipEventDisableTiming));
        }
        if (!e->upd_host)
        {
            // the per-update records (ddgi_types.h: UpdK): written here by the host, copied into the device ring by the kernel's workgroups
            HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&e->upd_host), kAqPubRing * sizeof(UpdK), hipHostMallocMapped | hipHostMallocCoherent));
            std::memset(e->upd_host, 0, kAqPubRing * sizeof(UpdK));
            HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void**>(&e->upd_host_dev), e->upd_host, 0));
        }
        if (!e->upd_dev)
        {
            HIP_TRY(hipMalloc(reinterpret_cast<void**>(&e->upd_dev), kAqCounters * sizeof(UpdK)));
            HIP_TRY(hipMemsetAsync(e->upd_dev, 0, kAqCounters * sizeof(UpdK), e->stream));
        }
        // one persistent workgroup per CU unless the launch is tiny (the launcher sizes the ray claims
        // so that every workgroup gets several: launch_probe_trace_wf)
        const uint32_t chunks = (a.n_rays + 255u) / 256u;
        p.grid = static_cast<uint32_t>(e->num_cus * (p.use_async ? aq_wgs_per_cu() : wf_blocks_per_cu));
        if (p.use_async && tn.reserve_cus > 0) p.grid = static_cast<uint32_t>(std::max(1, static_cast<int>(p.grid) - tn.reserve_cus * aq_wgs_per_cu()));
        if (p.grid > chunks) p.grid = chunks;
        const size_t slots = static_cast<size_t>(p.grid) * p.pool;
        if (slots > e->wf_cold_slots)
        {
            if (e->d_wf_cold) (void)hipFree(e->d_wf_cold);
            e->d_wf_cold = nullptr;
            e->wf_cold_slots = 0;
            HIP_TRY(hipMalloc(&e->d_wf_cold, slots * kWfColdBytes));
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
    }
    else
    {
        const size_t lds = static_cast<size_t>(a.scene.nwords) * sizeof(uint32_t);
        int per_cu = 0;
        HIP_TRY(trace_kernel_occupancy(&per_cu, lds));
        if (per_cu < 1) return fail(DDGI_ERR_UNSUPPORTED, "scene bitmap (%zu B) does not fit in LDS", lds);
        const uint32_t chunks = (a.n_rays + kTraceBlock - 1) / kTraceBlock;
        p.grid = static_cast<uint32_t>(e->num_cus) * static_cast<uint32_t>(per_cu);
        if (p.grid > chunks) p.grid = chunks;
    }
    p.key = aq_config_key(e, a, p.ddgi_mode);
    return DDGI_OK;
}

// The light-feeler classes of plan p's update (a.vis, a.vis_occ, a.vis_more).
//   follows (in / out; null: a launch outside the sequence of updates, ddgi_tune): the update would be published as the
//     continuation of its predecessor.  A launch that goes on with this update's rays reads the update's tables — so they must be
//     complete before the CHAIN'S FIRST launch starts, i.e. computed by kernels submitted in front of it.  Are there none (the
//     lights have moved to where nobody expected them), the update is not continued: *follows = false, it starts a chain of its
//     own behind the kernels that compute its tables — what every update with moved lights did before round 5.
//   chain_ahead: the updates a chain that starts with this update may go on to.  Their tables are made NOW, for the light
//     positions they are expected to have: the same lights (REF mode, ddgi_set_lights apart), or update_lights at the times the
//     host's steps so far lead to (DDGI mode: the reference adds 2 to RenderSettings::time per frame, src/rvpt/rvpt.cpp:281).
//     As many updates ahead as the host has lately been running ahead (ddgi_engine::runahead): a host that waits for every
//     update pays for no prediction.  A prediction that does not come true costs its kernel (34 us on the cave), never a result.
static int join_prepared(ddgi_engine* e);
static int assign_vis(ddgi_engine* e, TracePlan& p, bool* follows, int chain_ahead)
{
    TraceArgs& a = p.a;
    a.vis = nullptr, a.vis_occ = nullptr;
    for (auto& v : a.vis_more) v = nullptr;
    const Tuning& tn = e->tuning;
    if (!(a.nl >= 1 && tn.light_vis && tn.trace_kernel != 2)) return DDGI_OK;
    const int scene = a.scene_id;
    ddgi_engine::DevScene& d = e->dev_scene[scene];
    int k = -1;
    if (follows && *follows)
    {
        k = find_vis_set(e, scene, a.lights, a.nl, false, e->chain_first_seq);
        if (k < 0)
        {
            *follows = false;
            if (int rc = join_prepared(e)) return rc;  // (a chain starts here after all: what the preparation stream has made counts now)
        }
    }
    if (k < 0)
    {
        k = find_vis_set(e, scene, a.lights, a.nl, true, 0u);
        unsigned used = 0u;
        const int n_sets = vis_sets_allowed(a.scene, a.nl);  // (of this half's kAqChainMax: a large scene's tables are bounded in memory)
        auto victim = [&]() {  // a set this call has not used: the least recently computed one
            int best = -1;
            for (int i = 0; i < n_sets; ++i)
                if (!((used >> i) & 1u) && (best < 0 || static_cast<int32_t>(d.vis_set[i].launch_seq - d.vis_set[best].launch_seq) < 0)) best = i;
            return best;
        };
        if (k < 0)
        {
            k = victim();
            if (int rc = fill_vis_set(e, scene, a.scene, a.lights, a.nl, k)) return rc;
        }
        used |= 1u << k;
        const int n_pred = follows ? std::min(e->runahead, std::min(chain_ahead, n_sets - 1)) : 0;
        if (n_pred > 0 && p.ddgi_mode)
        {
            const float dt = e->have_last_time ? e->settings.time - e->last_time : 0.0f;
            float t = e->settings.time;
            LightK pl[kMaxLights];
            for (int j = 1; j <= n_pred; ++j)
            {
                t += dt;  // (as the host's own `time += step` rounds)
                animate_lights(scene, t, e->lights[scene], a.nl, pl);
                int kk = find_vis_set(e, scene, pl, a.nl, true, 0u);
                if (kk < 0)
                {
                    kk = victim();
                    if (kk < 0) break;
                    if (int rc = fill_vis_set(e, scene, a.scene, pl, a.nl, kk)) return rc;
                }
                used |= 1u << kk;
            }
        }
    }
    const ddgi_engine::DevScene::VisSet& set = d.vis_set[k];
    a.vis = set.vis[0];
    a.vis_occ = set.occ;
    for (int li = 1; li < a.nl && li < kVisLights; ++li) a.vis_more[li - 1] = set.vis[li];
    return DDGI_OK;
}

// What the preparation stream was given at the previous update (prepare_ahead) becomes the handle's: its stream waits for it —
// beside the previous blend it is long done — and the tables made there count as complete from the next launch on.
static int join_prepared_weights(ddgi_engine* e)
{
    if (!e->prep_w_pending) return DDGI_OK;
    HIP_TRY(hipStreamWaitEvent(e->stream, e->prep_w_done, 0));
    e->prep_w_pending = false;
    return DDGI_OK;
}
// (the tables: only where a chain starts — an update that continues its predecessor could not use them anyway, and its own launch,
// empty, would wait for kernels that have the whole group's blends to hide behind)
static int join_prepared(ddgi_engine* e)
{
    if (!e->prep_pending) return DDGI_OK;
    HIP_TRY(hipStreamWaitEvent(e->stream, e->prep_done, 0));
    e->prep_pending = false;
    for (auto& d : e->dev_scene)
        for (auto& set : d.vis_set)
            if (set.on_prep_stream) set.on_prep_stream = false, set.launch_seq = e->launch_seq;
    return DDGI_OK;
}

// DDGI mode, behind an update's trace launch: the next frame's weight tiles and the light-feeler tables of the updates to come, on
// the preparation stream — beside this update's blend instead of in front of the next update's trace.  `after`: an event already
// recorded behind the launch (the update's timing event), or null.
//   weights  frame f + 1's rotation is known (frame_rotation); buffer (f + 1) & 1 was last read by the blend of frame f - 1, which
//            ended before this update's launch started.
//   tables   for updates u = 1 .. nrec ahead, at the light positions update_lights gives for time + u x (this update's time step) —
//            the host's own `time += step` (src/rvpt/rvpt.cpp:281) —, unless a set holds them already; into set kAqChainMax +
//            (number of that update) % kAqChainMax, whose previous contents belonged to an update traced by launches that ended
//            before this one did.  A prediction that does not come true is never used (assign_vis compares the positions).
static int prepare_ahead(ddgi_engine* e, const TracePlan& p, const BlendArgs& b, hipEvent_t after)
{
    if (!e->prep_stream)
    {
        HIP_TRY(hipStreamCreateWithFlags(&e->prep_stream, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&e->prep_after, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&e->prep_done, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&e->prep_w_done, hipEventDisableTiming));
    }
    if (!after)
    {
        HIP_TRY(hipEventRecord(e->prep_after, e->stream));
        after = e->prep_after;
    }
    HIP_TRY(hipStreamWaitEvent(e->prep_stream, after, 0));
    // THE TABLES' OWN STREAM (tuning "prep_stream" 2, the default).  On one preparation stream the feeler tables' kernel stands behind the weight
    // kernels, which start with the blend: it starts when the blend ends — and the NEXT update's launch, which needs every CU EMPTY for a
    // moment (a persistent 1 024-lane workgroup takes a whole CU's registers and LDS) even when it has nothing left to trace, waits for it to
    // drain: a continued update cost 60 us on a slab of an 8-way sharded C3 grid (blend 25, then the tables 29 with the empty launch stuck
    // behind them: tools/slab_timeline.sh).  On a stream of their own the tables start beside the blend and are nearly done when it is:
    // 60 -> 50 us per continued update, a G = 8 DDGI slab 0.341 -> 0.334 ms.  Only where the blend is the small merged kernel (few probes per
    // rank: launch_probe_blend's own test) — beside the whole grid's persistent depth workgroups the tables' kernel takes their CUs and the
    // update gets 1.5 % SLOWER (profiles/r05_k_prep_stream_ab.txt); "prep_stream" 3: always, 1: never.
    const bool small_blend = ((b.n_local_probes + 15u) / 16u) * 2u <= static_cast<uint32_t>(e->num_cus) * b.merge_below;
    hipStream_t tables_stream = e->prep_stream;
    if (e->tuning.prep_stream >= 3 || (e->tuning.prep_stream == 2 && small_blend))
    {
        if (!e->prep_stream2) HIP_TRY(hipStreamCreateWithFlags(&e->prep_stream2, hipStreamNonBlocking));
        HIP_TRY(hipStreamWaitEvent(e->prep_stream2, after, 0));
        tables_stream = e->prep_stream2;
    }
    // (the stream that serves the tables can change between updates — tuning "prep_stream", a slab that becomes a whole grid —, and prep_done is
    //  ONE event: whatever an earlier update put on the other stream and nobody has joined yet is ordered in front of this call's record)
    if (e->prep_pending && e->prep_tables_last && e->prep_tables_last != tables_stream) HIP_TRY(hipStreamWaitEvent(tables_stream, e->prep_done, 0));
    e->prep_tables_last = tables_stream;
    const TraceArgs& a = p.a;
    {
        BlendArgs nb = b;
        frame_rotation(e->frame + 1u, nb.rot);
        const int wb = static_cast<int>((e->frame + 1u) & 1u);
        nb.w_sum = e->d_blend_w + static_cast<size_t>(wb) * e->d_blend_w_floats;
        nb.w = nb.w_sum + 256;
        HIP_TRY(launch_blend_weights(nb, e->prep_stream));
        auto& made = e->w_made[wb];
        made.valid = true, made.frame = e->frame + 1u, made.n = a.grid.n;
        std::memcpy(made.rot, nb.rot, sizeof made.rot);
        HIP_TRY(hipEventRecord(e->prep_w_done, e->prep_stream));
        e->prep_w_pending = true;
    }
    if (a.vis && e->have_last_time)
    {
        const int scene = a.scene_id;
        const float dt = e->settings.time - e->last_time;
        float t = e->settings.time;
        LightK pl[kMaxLights];
        for (int u = 1; u <= std::max(1, e->nrec); ++u)
        {
            t += dt;
            animate_lights(scene, t, e->lights[scene], a.nl, pl);
            if (find_vis_set(e, scene, pl, a.nl, true, 0u) >= 0) continue;
            bool coming = false;  // (already on its way on the preparation stream)
            for (int k = kAqChainMax; k < 2 * kAqChainMax; ++k) coming = coming || (e->dev_scene[scene].vis_set[k].on_prep_stream && vis_set_holds(e->dev_scene[scene].vis_set[k], pl, a.nl));
            if (coming) continue;
            // (a scene whose tables are bounded in memory gets NO sets on this stream: with fewer than kAqChainMax of them a set would be rewritten here, beside
            //  the launches, while a launch that went on with an earlier update may still read it — its moving lights' updates start chains of their own, as before round 5)
            if (vis_sets_allowed(a.scene, a.nl) < kAqChainMax) break;
            const int k = kAqChainMax + static_cast<int>((e->updates + static_cast<unsigned long long>(u)) % kAqChainMax);
            if (e->dev_scene[scene].vis_set[k].on_prep_stream) continue;  // (still waiting for the next chain to start: left alone)
            if (int rc = fill_vis_set(e, scene, a.scene, pl, a.nl, k, tables_stream)) return rc;
        }
    }
    HIP_TRY(hipEventRecord(e->prep_done, tables_stream));  // (the tables; the weights have their own event)
    e->prep_pending = true;
    return DDGI_OK;
}

// The part of a launch's arguments that may differ between the updates one launch works on (ddgi_types.h: UpdK).
static UpdK upd_of(const TraceArgs& a)
{
    UpdK u;
    std::memset(&u, 0, sizeof u);
    for (int i = 0; i < kMaxLights; ++i) u.lights[i] = a.lights[i];
    for (int i = 0; i < 9; ++i) u.rot[i] = a.rot[i];
    u.frame_key = a.frame_key;
    u.rays = a.rays, u.rad_rgb = a.rad_rgb, u.rad_dd = a.rad_dd;
    u.vis = a.vis, u.vis_occ = a.vis_occ;
    for (int i = 0; i < kVisLights - 1; ++i) u.vis_more[i] = a.vis_more[i];
    return u;
}

// What a launch is, for "can the predecessor's launch go on with this update's rays": every argument of the kernel but the textures
// it writes, and the kernel's shape.  DDGI mode: also without the per-update record (upd_of: lights, rotation, key, ray / record
// buffers, feeler classes) — its kernels read those per update.  REF mode's kernels read them from their own arguments: equal
// byte for byte, or no continuation.  (What the other arguments POINT at — scene, noise tables, the rays — is covered by
// ddgi_engine::chain_break: every entry point that can change any of it sets it.)
static unsigned long long plan_hash(const TracePlan& p, int march_waves)
{
    TraceArgs a = p.a;
    a.albedo = a.distance = nullptr;
    if (p.ddgi_mode)
    {
        std::memset(a.lights, 0, sizeof a.lights);
        std::memset(a.rot, 0, sizeof a.rot);
        a.frame_key = 0;
        a.rays = nullptr, a.rad_rgb = a.rad_dd = nullptr;
    }
    // (the feeler classes follow from the scene and the lights' positions: assign_vis gives equal lights the same tables, or says no)
    a.vis = nullptr, a.vis_occ = nullptr;
    for (auto& v : a.vis_more) v = nullptr;
    unsigned long long h = 1469598103934665603ull;
    const unsigned char* b = reinterpret_cast<const unsigned char*>(&a);
    for (size_t i = 0; i < sizeof a; ++i) h = (h ^ b[i]) * 1099511628211ull;
    const unsigned extra[6] = {static_cast<unsigned>(p.pool), p.grid, static_cast<unsigned>(march_waves), p.use_async ? 1u : 0u, p.fast ? 1u : 0u, static_cast<unsigned>(p.wf_threads)};
    for (unsigned v : extra) h = (h ^ v) * 1099511628211ull;
    return h | 1ull;
}

// One launch of the queue kernel with the handle's next sequence number.  chain_max: updates after its own the launch may go
// on with; publish: this launch's update continues its predecessor's (ddgi_probe_update) — said so in the host ring BEFORE the
// launch, so that a predecessor still running can pick the rays up.
static int launch_aq_numbered(ddgi_engine* e, const TracePlan& p, int march_waves, int chain_max, bool publish)
{
    if (e->counters_dirty)
    {
        // a launch failed after its number was handed out: the counter it was to zero may be stale.  Start over on a clean ring.
        DDGI_TRY(ddgi_sync_stream(e, e->stream));
        HIP_TRY(hipMemsetAsync(e->d_work + 8, 0, kAqCounters * sizeof(uint32_t), e->stream));
        e->launch_seq = (e->launch_seq + 2u * kAqCounters - 1u) & ~(kAqCounters - 1u);
        e->counters_dirty = false;
        publish = false, chain_max = 0;
    }
    const uint32_t seq = e->launch_seq;
    // the host ring is reused every kAqPubRing launches: never run further ahead of the device than 48 launches
    if ((seq & 15u) == 0u)
    {
        const int m = static_cast<int>((seq >> 4) & 1u);
        if (seq >= 32u) DDGI_TRY(ddgi_sync_event(e, e->milestone[m]));  // (recorded 32 launches ago)
        HIP_TRY(hipEventRecord(e->milestone[m], e->stream));
    }
    // The update's record (ddgi_types.h: UpdK), written before the release store that publishes the update: its own launch's
    // workgroups read it, and those of a predecessor's launch that goes on with its rays.
    if (!publish) e->chain_first_seq = seq;
    *(reinterpret_cast<UpdK*>(e->upd_host) + static_cast<size_t>(seq & (kAqPubRing - 1u))) = upd_of(p.a);
    // (a slot still holds what launch seq - 64 left there, which is never seq + 1)
    __atomic_store_n(&e->pub[seq & (kAqPubRing - 1u)], publish ? seq + 1u : 0u, __ATOMIC_RELEASE);
    AqChain c;
    c.upd_host = e->upd_host_dev;
    c.upd_dev = e->upd_dev;
    c.counters = e->d_work + 8;
    c.continued = e->d_work + 4;
    c.pub = e->pub_dev;
    c.seq = seq;
    c.chain_max = static_cast<uint32_t>(chain_max);
    e->launch_seq = seq + 1u;
    const hipError_t he = launch_probe_trace_aq(p.a, p.pool, static_cast<int>(p.grid), march_waves, c, e->d_work + 1, e->stream);
    if (he != hipSuccess)
    {
        e->counters_dirty = true;
        return fail(DDGI_ERR_HIP, "launching the trace kernel failed: %s", hipGetErrorString(he));
    }
    return DDGI_OK;
}

// How many of the queue kernel's 16 waves march (the rest run events) is the one knob the balance of a scene
// moves: C3 is fastest at 5 (4: 3.44 ms, 5: 2.96, 6: 3.14, 8: 3.7), Cornell and the house at 6, a sparser cave
// grid at 3.  Measured per configuration: a few extra launches of the same (idempotent) trace, hill-climbing
// from `start`.  BLOCKS the host (hipEventSynchronize); some tens of milliseconds.
static int measure_march_waves(ddgi_engine* e, const TracePlan& p, int start, int* out)
{
    hipEvent_t* ev = e->ev[e->updates % ddgi_engine::kRing];
    int march_waves = start;
    auto timed = [&](int mw, float* ms) -> int {  // the faster of two launches
        *ms = 0.0f;
        for (int rep = 0; rep < 2; ++rep)
        {
            float t = 0.0f;
            HIP_TRY(hipEventRecord(ev[0], e->stream));
            if (int rc = launch_aq_numbered(e, p, mw, 0, false)) return rc;
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
        for (int mw = march_waves + dir; mw >= 2 && mw <= std::min(12, aq_threads() / 64 - 2); mw += dir)
        {
            if (int rc = timed(mw, &ms)) return rc;
            if (ms >= best_ms) break;
            best_ms = ms, march_waves = mw, moved = true;
        }
        if (moved) break;  // downhill in this direction: the other one was uphill
    }
    if (e->tuning.verbose) std::fprintf(stderr, "[ddgi] queue kernel: %d march waves / %d event waves for this configuration\n", march_waves, aq_threads() / 64 - march_waves);
    *out = march_waves;
    return DDGI_OK;
}

// The split for plan p: pinned ("march_waves" tuning) > measured earlier for this configuration (ddgi_tune, or "autotune") >
// measured now (when `may_block` and "autotune" allow it) > the last measured value > 5 (the fast march: 4).
static int choose_march_waves(ddgi_engine* e, const TracePlan& p, bool may_block, bool force_measure, int* out)
{
    const Tuning& tn = e->tuning;
    if (tn.march_waves > 0 && !force_measure)
    {
        *out = std::min(15, std::max(1, tn.march_waves));
        return DDGI_OK;
    }
    auto it = e->aq_split.find(p.key);
    if (it != e->aq_split.end() && !force_measure)
    {
        *out = it->second;
        return DDGI_OK;
    }
    int mw = e->aq_last > 0 ? e->aq_last : std::max(2, (p.fast ? 4 : 5) * aq_threads() / 1024);
    if (force_measure || (may_block && tn.autotune && tn.ablate == 0))
    {
        if (int rc = measure_march_waves(e, p, mw, &mw)) return rc;
        if (e->aq_split.size() >= 64) e->aq_split.erase(e->aq_split.begin());  // bounded: an app cycling through more configurations re-measures
        e->aq_split[p.key] = mw;
        e->aq_last = mw;
    }
    *out = mw;
    return DDGI_OK;
}

int ddgi_tune(ddgi_handle e)
{
    if (!e) return fail(DDGI_ERR_INVALID_ARGUMENT, "null handle");
    e->tex_ops_since_update = true;  // (ddgi_exchange: something besides the update may be using the textures on the stream)
    e->chain_break = true;
    HIP_TRY(hipSetDevice(e->device));
    TracePlan p;
    if (int rc = plan_trace(e, p)) return rc;
    if (!p.use_async) return DDGI_OK;  // nothing to tune for the other trace kernels
    if (int rc = assign_vis(e, p, nullptr, 0)) return rc;
    int mw = 0;
    return choose_march_waves(e, p, true, true, &mw);
}

int ddgi_probe_update(ddgi_handle e, const ddgi_render_settings* settings)
{
    if (!e) return fail(DDGI_ERR_INVALID_ARGUMENT, "null handle");
    if (settings)
    {
        if (settings->scene < 0 || settings->scene > 3) return fail(DDGI_ERR_INVALID_ARGUMENT, "scene %d not in {0,1,2,3}", settings->scene);
        e->settings = *settings;
    }
    HIP_TRY(hipSetDevice(e->device));
    TracePlan p;
    if (int rc = plan_trace(e, p)) return rc;
    TraceArgs& a = p.a;
    // (the queue kernel's wave split: pinned, measured earlier for this configuration, or — tuning "autotune" — measured below, by
    // launches into the pair this update writes)
    int march_waves = 5;
    const bool measure_split = p.pool > 0 && p.use_async && e->tuning.autotune && e->tuning.ablate == 0 && e->tuning.march_waves <= 0 && e->aq_split.find(p.key) == e->aq_split.end();
    if (p.pool > 0 && p.use_async && !measure_split)
        if (int rc = choose_march_waves(e, p, false, false, &march_waves)) return rc;

    // ---- which texture pair of the ring this update writes -------------------------------------------------------------
    // Frames in flight (REF mode, the queue kernel): updates come in groups of up to `len`.  The first of a group takes the next
    // aligned run of `len` pairs; an update that is THE SAME WORK as its predecessor (same launch, byte for byte, and nothing but
    // probe updates, exchanges and consumers on the handle in between) continues the group into the next pair — and is published,
    // so that the predecessor's launch, if it is still running when its own rays are used up, goes on with this update's rays
    // instead of draining (k_probe_trace_aq).  Each update still has its own launch, in stream order behind its predecessor's:
    // whatever that one left is traced there, and everything stream-ordered behind an update's launch sees the update complete.
    // Which pair an update writes and where it stands in its group depend on the NUMBER of the update only (ring_k: updates since
    // the ring was made) — every rank of a sharded grid must pick the same pair whatever it decides locally about continuing.
    // DDGI mode (round 5): the trace writes ray records, not textures — a group is the ring of record buffers (plan_trace), update k
    // writes buffer k % nrec, and what differs from update to update (rotation, key, animated lights, the buffer) travels in the
    // update's record (ddgi_types.h: UpdK), which the kernel reads per update.
    const int group = ddgi_group_len(e);
    const bool ring_fits = p.ddgi_mode || e->np % group == 0;
    const int pos = ring_fits ? static_cast<int>(e->ring_k % static_cast<unsigned long long>(group)) : 0;
    const bool can_chain = group > 1 && ring_fits && p.pool > 0 && p.use_async && a.stats == nullptr && !measure_split;
    if (p.ddgi_mode)
    {
        a.rad_rgb = e->d_radiance + static_cast<size_t>(pos) * e->rec_stride;
        a.rad_dd = a.rad_rgb + p.rec_rgb;
    }
    const unsigned long long hash = plan_hash(p, march_waves);
    // (how far the host runs ahead of its blocking calls: updates that arrive with nothing but updates / exchanges / consumers since
    // their predecessor raise it, a blocking call that found nothing to continue lowers it — assign_vis predicts that far)
    if (!e->chain_break) e->runahead = std::min(kAqChainMax - 1, e->runahead + 1);
    else if (e->chain_published == 0) e->runahead = std::max(0, e->runahead - 1);
    if (e->chain_break) e->chain_published = 0;
    bool follows = can_chain && pos > 0 && !e->chain_break && hash == e->chain_hash;
    if (int rc = join_prepared_weights(e)) return rc;
    if (!follows)
        if (int rc = join_prepared(e)) return rc;
    if (int rc = assign_vis(e, p, &follows, can_chain ? group - 1 - pos : 0)) return rc;
    // (what the handle goes back to when the update cannot be launched: consumers keep reading the latest finished update)
    struct Saved
    {
        ddgi_engine* e;
        void *tex[2], *prev[2];
        int pair_cur;
        bool launched = false;
        ~Saved()
        {
            if (launched) return;
            for (int i = 0; i < 2; ++i) e->tex[i] = tex[i], e->tex_prev[i] = prev[i];
            e->pair_cur = pair_cur;
            e->chain_break = true;
            if (e->xch.transport) e->xch.desync = true;  // (this rank's count of updates has fallen behind its peers': ddgi_exchange says so)
        }
    } saved{e, {e->tex[0], e->tex[1]}, {e->tex_prev[0], e->tex_prev[1]}, e->pair_cur};
    if (!e->caller_tex)
    {
        const bool stay = e->pin_pair && !e->xch.pipelined;  // (the host holds pointers to this pair: ddgi_device_textures)
        const int pair = stay ? e->pair_cur : static_cast<int>(e->ring_k % static_cast<unsigned long long>(e->np));
        // pipelined exchange: the pairs this launch may write — its own and the ones it may continue into — must have left for the
        // other ranks (the launch that may continue waits for the rest of its group's pairs as well: a continued update has no
        // wait of its own that the predecessor's launch would see)
        // (DDGI mode: the textures are written by the update's own blend, behind this wait, whoever traces its rays)
        if (!follows || p.ddgi_mode)
            if (int rc = ddgi_exchange_before_update(e, pair, (can_chain && !p.ddgi_mode) ? group - pos : 1)) return rc;
        for (int i = 0; i < 2; ++i)
        {
            e->tex_prev[i] = pair != e->pair_cur ? e->tex[i] : nullptr;  // DDGI blend: where the previous update's tiles are, when not in place
            e->tex[i] = ddgi_pair_ptr(e, pair, i);
        }
        e->pair_cur = pair;
    }
    else
        e->tex_prev[0] = e->tex_prev[1] = nullptr;
    a.albedo = static_cast<uint32_t*>(e->tex[0]);
    a.distance = static_cast<uint32_t*>(e->tex[1]);
    const int chain_max = can_chain ? group - 1 - pos : 0;
    // (!= 0 also tells the kernel that rays carry their update in dst[31:29]; DDGI mode's rays write records: any non-zero value)
    a.pair_words = can_chain ? (p.ddgi_mode ? 1u : static_cast<uint32_t>(e->tex_bytes[0] / 4)) : 0u;
    if (measure_split)
        if (int rc = choose_march_waves(e, p, true, false, &march_waves)) return rc;

    hipEvent_t* ev = e->ev[e->updates % ddgi_engine::kRing];
    const bool timing = e->tuning.timing != 0;  // (two or three events per update: ~3 us of the stream's time each)
    if (timing) HIP_TRY(hipEventRecord(ev[0], e->stream));  // (DDGI mode: the trace time includes the blend's two small weight kernels)
    BlendArgs b{};
    if (p.ddgi_mode)
    {
        b.grid = a.grid;
        for (int i = 0; i < 9; ++i) b.rot[i] = a.rot[i];
        b.rad_rgb = a.rad_rgb;
        b.rad_dd = a.rad_dd;
        b.irradiance = static_cast<float*>(e->tex[0]);
        b.depth = static_cast<float*>(e->tex[1]);
        b.irradiance_old = static_cast<const float*>(e->tex_prev[0] ? e->tex_prev[0] : e->tex[0]);
        b.depth_old = static_cast<const float*>(e->tex_prev[1] ? e->tex_prev[1] : e->tex[1]);
        b.n_local_probes = static_cast<uint32_t>(a.grid.cx) * a.grid.cy * a.grid.czl;
        const size_t need = (blend_weights_floats(a.grid.n) + 256 + 63) / 64 * 64;
        if (need > e->d_blend_w_floats)
        {
            if (e->d_blend_w) (void)hipFree(e->d_blend_w);  // (waits for the device, the preparation stream included)
            e->d_blend_w = nullptr, e->d_blend_w_floats = 0;
            e->w_made[0].valid = e->w_made[1].valid = false;
            HIP_TRY(hipMalloc(reinterpret_cast<void**>(&e->d_blend_w), 2 * need * sizeof(float)));
            e->d_blend_w_floats = need;
        }
        // frame f's weight tiles live in buffer f & 1: they depend on the frame's ray directions only, and the NEXT frame's are made
        // beside this frame's blend (below) — here only if that did not happen (the first update, ddgi_set_frame, another ray count)
        const int wb = static_cast<int>(e->frame & 1u);
        b.w_sum = e->d_blend_w + static_cast<size_t>(wb) * e->d_blend_w_floats;  // (the buffers' stride is the ALLOCATION's — it never shrinks —, here and in
        b.w = b.w_sum + 256;                                                        //  prepare_ahead: `need` is smaller after a reconfiguration to fewer rays)
        if (e->tuning.blend_kernel == 1) b.w = b.w_sum = nullptr;  // one probe per workgroup, weights in place
        b.force_division = e->tuning.blend_kernel == 2 ? 1u : 0u;
        b.merge_below = static_cast<uint32_t>(e->tuning.blend_merge < 0 ? 0 : e->tuning.blend_merge);
        auto& made = e->w_made[wb];
        if (b.w && !(made.valid && made.frame == e->frame && made.n == a.grid.n && std::memcmp(made.rot, b.rot, sizeof made.rot) == 0))
        {
            HIP_TRY(launch_blend_weights(b, e->stream));  // (before the trace: off the critical path between the last ray and the first tile)
            made.valid = true, made.frame = e->frame, made.n = a.grid.n;
            std::memcpy(made.rot, b.rot, sizeof made.rot);
        }
    }
    if (p.pool > 0)
    {
        if (p.use_async)
        {
            if (int rc = launch_aq_numbered(e, p, march_waves, chain_max, follows)) return rc;
        }
        else
            HIP_TRY(launch_probe_trace_wf(a, p.wf_threads, p.pool, static_cast<int>(p.grid), e->d_work + 3, e->stream));
    }
    else
        HIP_TRY(launch_probe_trace_ref(a, static_cast<int>(p.grid), e->stream));
    if (timing) HIP_TRY(hipEventRecord(ev[1], e->stream));
    if (p.ddgi_mode)
    {
        if (e->tuning.prep_stream && p.pool > 0 && p.use_async && b.w)
            if (int rc = prepare_ahead(e, p, b, timing ? ev[1] : nullptr)) return rc;
        HIP_TRY(launch_probe_blend(b, e->num_cus, e->stream));
        e->frame += 1;
    }
    saved.launched = true;
    if (follows) e->chain_published += 1;
    e->last_dt = e->have_last_time ? e->settings.time - e->last_time : 0.0f;
    e->last_time = e->settings.time, e->have_last_time = true;
    e->chain_hash = hash, e->chain_break = measure_split || !can_chain;
    e->ring_k += 1;
    e->box_of = nullptr;  // the textures change: the sampler's per-texel table is stale
    // (REF mode: nothing follows the trace kernel, its end event is the update's end — an event costs the stream microseconds)
    e->ev_has_blend[e->updates % ddgi_engine::kRing] = p.ddgi_mode;
    e->ev_valid[e->updates % ddgi_engine::kRing] = timing;
    if (timing && p.ddgi_mode) HIP_TRY(hipEventRecord(ev[2], e->stream));
    e->updates += 1;
    e->tex_ops_since_update = false;
    return DDGI_OK;
}

// After a stream synchronisation: did a trace kernel trip its safety net (k_probe_trace_aq: a queue wait
// that never ended)?  Then the textures are not valid — say so instead of returning them.
static int check_kernel_status(ddgi_engine* e)
{
    if (!e->d_work) return DDGI_OK;
    uint32_t status = 0;
    HIP_TRY(hipMemcpy(&status, e->d_work + 1, sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (status != 0)
    {
        // reported once: the flag is cleared so that later (clean) updates on this handle are usable again
        (void)hipMemsetAsync(e->d_work + 1, 0, sizeof(uint32_t), e->stream);  // (in stream order: in front of the next launch, not beside it)
        return fail(DDGI_ERR_HIP, "the trace kernel aborted (status %u): the probe textures are not valid", status);
    }
    return DDGI_OK;
}

int ddgi_synchronize(ddgi_handle e)
{
    if (!e) return fail(DDGI_ERR_INVALID_ARGUMENT, "null handle");
    HIP_TRY(hipSetDevice(e->device));
    DDGI_TRY(ddgi_sync_stream(e, e->stream));
    e->chain_break = true;  // nothing is in flight: the next update starts a group of its own —
    {
        // — of its own in the ring as well: the next update's number moves up to the next group's first
        // — unless the handle exchanges its textures with other ranks: which pair an update writes is a function of its number on
        // EVERY rank, and one rank may synchronise where another does not
        const unsigned long long g = static_cast<unsigned long long>(ddgi_group_len(e));
        if (!e->xch.transport && !e->xch.p2p && (e->mode == DDGI_MODE_DDGI || e->np % static_cast<int>(g) == 0)) e->ring_k = (e->ring_k + g - 1) / g * g;
    }
    return check_kernel_status(e);
}

int ddgi_last_update_ms(ddgi_handle e, float* trace_ms, float* blend_ms, float* total_ms)
{
    if (!e) return fail(DDGI_ERR_INVALID_ARGUMENT, "null handle");
    if (e->updates == 0) return fail(DDGI_ERR_NOT_READY, "no probe update has been issued yet");
    HIP_TRY(hipSetDevice(e->device));
    const size_t slot = (e->updates - 1) % ddgi_engine::kRing;
    if (!e->ev_valid[slot]) return fail(DDGI_ERR_NOT_READY, "the last update was not timed (tuning \"timing\" is 0)");
    hipEvent_t* ev = e->ev[slot];
    const bool blend = e->ev_has_blend[slot];
    DDGI_TRY(ddgi_sync_event(e, ev[blend ? 2 : 1]));
    float t01 = 0.f, t12 = 0.f, t02 = 0.f;
    HIP_TRY(hipEventElapsedTime(&t01, ev[0], ev[1]));
    if (blend) HIP_TRY(hipEventElapsedTime(&t12, ev[1], ev[2]));
    HIP_TRY(hipEventElapsedTime(&t02, ev[0], ev[blend ? 2 : 1]));
    if (trace_ms) *trace_ms = t01;
    if (blend_ms) *blend_ms = t12;
    if (total_ms) *total_ms = t02;
    return DDGI_OK;
}

int ddgi_update_history_ms(ddgi_handle e, float* trace_ms, float* blend_ms, int capacity, int* n_out)
{
    if (!e || !n_out || capacity < 0) return fail(DDGI_ERR_INVALID_ARGUMENT, "null handle/n_out");
    HIP_TRY(hipSetDevice(e->device));
    DDGI_TRY(ddgi_sync_stream(e, e->stream));
    unsigned long long have = e->updates < static_cast<unsigned long long>(ddgi_engine::kRing) ? e->updates : ddgi_engine::kRing;
    if (have > static_cast<unsigned long long>(capacity)) have = capacity;
    for (unsigned long long i = 0; i < have; ++i)
        if (!e->ev_valid[(e->updates - have + i) % ddgi_engine::kRing]) return fail(DDGI_ERR_NOT_READY, "an update of the history was not timed (tuning \"timing\" was 0)");
    for (unsigned long long i = 0; i < have; ++i)
    {
        const size_t slot = (e->updates - have + i) % ddgi_engine::kRing;
        hipEvent_t* ev = e->ev[slot];
        float t01 = 0.f, t12 = 0.f;
        HIP_TRY(hipEventElapsedTime(&t01, ev[0], ev[1]));
        if (e->ev_has_blend[slot]) HIP_TRY(hipEventElapsedTime(&t12, ev[1], ev[2]));
        if (trace_ms) trace_ms[i] = t01;
        if (blend_ms) blend_ms[i] = t12;
    }
    *n_out = static_cast<int>(have);
    return DDGI_OK;
}

int ddgi_trace_stats(ddgi_handle e, int enable, unsigned long long* out64)
{
    if (!e) return fail(DDGI_ERR_INVALID_ARGUMENT, "null handle");
    HIP_TRY(hipSetDevice(e->device));
    DDGI_TRY(ddgi_sync_stream(e, e->stream));
    if (out64)
    {
        std::memset(out64, 0, 64 * sizeof(unsigned long long));
        if (e->d_stats) HIP_TRY(hipMemcpy(out64, e->d_stats, 64 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    }
    if (enable && !e->d_stats) HIP_TRY(hipMalloc(reinterpret_cast<void**>(&e->d_stats), 64 * sizeof(unsigned long long)));
    if (e->d_stats) HIP_TRY(hipMemsetAsync(e->d_stats, 0, 64 * sizeof(unsigned long long), e->stream));  // (in stream order)
    if (!enable && e->d_stats)
    {
        (void)hipFree(e->d_stats);
        e->d_stats = nullptr;
    }
    return DDGI_OK;
}

// The host-side consumers copy into the CALLER's (or a local) host buffer with hipMemcpyAsync and then wait.  With an exchange attached that
// wait has a deadline (ddgi_sync_stream) — and a call that gives up must not leave a copy in flight into memory it is about to hand
// back.  So: whatever depends on another rank is waited for FIRST, with the deadline, before any host copy is enqueued; after it nothing
// on the stream waits for anybody but this GPU.
static int peers_done_before_host_copies(ddgi_engine* e)
{
    if (!e->xch.transport && !e->xch.p2p) return DDGI_OK;
    return ddgi_sync_stream(e, e->stream);
}

int ddgi_read_textures(ddgi_handle e, uint8_t* albedo, uint8_t* distance)
{
    if (!e) return fail(DDGI_ERR_INVALID_ARGUMENT, "null handle");
    e->tex_ops_since_update = true;  // (ddgi_exchange: something besides the update may be using the textures on the stream)
    if (e->mode != DDGI_MODE_REF) return fail(DDGI_ERR_UNSUPPORTED, "ddgi_read_textures is REF-mode only; use ddgi_read_tiles");
    HIP_TRY(hipSetDevice(e->device));
    const GridK g = make_grid(e);
    const size_t s = g.sx, sh = g.sy, s2 = g.n;
    const size_t W = static_cast<size_t>(g.cx) * g.cz * s;
    if (int rc = ddgi_exchange_wait_latest(e)) return rc;
    if (int rc = peers_done_before_host_copies(e)) return rc;
    std::vector<uint32_t> slab(e->tex_bytes[0] / 4);
    uint8_t* outs[2] = {albedo, distance};
    for (int t = 0; t < 2; ++t)
    {
        if (!outs[t]) continue;
        HIP_TRY(hipMemcpyAsync(slab.data(), e->tex[t], e->tex_bytes[t], hipMemcpyDeviceToHost, e->stream));
        DDGI_TRY(ddgi_sync_stream(e, e->stream));
        if (int rc = check_kernel_status(e)) return rc;
        uint32_t* raster = reinterpret_cast<uint32_t*>(outs[t]);
        // slab-major [z][y][x][ty][tx] -> reference raster: tile of probe p at ((p mod cx*cz)*s, (p div cx*cz)*s)
        for (int z = 0; z < g.cz; ++z)
            for (int y = 0; y < g.cy; ++y)
                for (int x = 0; x < g.cx; ++x)
                {
                    const uint32_t* tile = slab.data() + ((static_cast<size_t>(z) * g.cy + y) * g.cx + x) * s2;
                    const size_t col0 = (static_cast<size_t>(z) * g.cx + x) * s;
                    const size_t row0 = static_cast<size_t>(y) * sh;
                    for (size_t ty = 0; ty < sh; ++ty)
                        std::memcpy(raster + (row0 + ty) * W + col0, tile + ty * s, s * 4);
                }
    }
    return DDGI_OK;
}

int ddgi_read_tiles(ddgi_handle e, float* irradiance, float* depth)
{
    if (!e) return fail(DDGI_ERR_INVALID_ARGUMENT, "null handle");
    e->tex_ops_since_update = true;  // (ddgi_exchange: something besides the update may be using the textures on the stream)
    if (e->mode != DDGI_MODE_DDGI) return fail(DDGI_ERR_UNSUPPORTED, "ddgi_read_tiles is DDGI-mode only; use ddgi_read_textures");
    HIP_TRY(hipSetDevice(e->device));
    const GridK g = make_grid(e);
    if (int rc = ddgi_exchange_wait_latest(e)) return rc;
    if (int rc = peers_done_before_host_copies(e)) return rc;
    float* outs[2] = {irradiance, depth};
    const size_t per_probe[2] = {8 * 8 * 4, 16 * 16 * 2};
    for (int t = 0; t < 2; ++t)
    {
        if (!outs[t]) continue;
        std::vector<float> slab(e->tex_bytes[t] / sizeof(float));
        HIP_TRY(hipMemcpyAsync(slab.data(), e->tex[t], e->tex_bytes[t], hipMemcpyDeviceToHost, e->stream));
        DDGI_TRY(ddgi_sync_stream(e, e->stream));
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

int ddgi_set_tuning(ddgi_handle e, const char* name, int value)
{
    if (!e || !name) return fail(DDGI_ERR_INVALID_ARGUMENT, "null handle/name");
    for (const TuningKey& k : kTuningKeys)
        if (!std::strcmp(k.name, name))
        {
            e->chain_break = true;
            if (k.field == &Tuning::frames_in_flight)
            {
                // the ring of texture pairs follows (blocks; the current textures carry over).  The multi-GPU exchange has the
                // ring's addresses (RCCL calls in flight, IPC mappings on the peers): detach it first.
                if (value < 1 || value > kAqChainMax) return fail(DDGI_ERR_INVALID_ARGUMENT, "frames_in_flight %d not in [1,%d]", value, kAqChainMax);
                if (e->xch.transport || e->xch.p2p) return fail(DDGI_ERR_INVALID_ARGUMENT, "set frames_in_flight before the exchange is set up (ddgi_exchange_init(h, NULL, 0) detaches it)");
                const int before = e->tuning.frames_in_flight;
                e->tuning.frames_in_flight = value;
                if (!e->caller_tex)
                {
                    HIP_TRY(hipSetDevice(e->device));
                    if (int rc = ddgi_resize_ring(e, ddgi_pairs_wanted(e, false)))
                    {
                        e->tuning.frames_in_flight = before;
                        return rc;
                    }
                }
                return DDGI_OK;
            }
            e->tuning.*(k.field) = value;
            return DDGI_OK;
        }
    return fail(DDGI_ERR_INVALID_ARGUMENT, "unknown tuning key '%s'", name);
}

int ddgi_get_tuning(ddgi_handle e, const char* name, int* value)
{
    if (!e || !name || !value) return fail(DDGI_ERR_INVALID_ARGUMENT, "null handle/name/value");
    if (!std::strcmp(name, "march_waves_measured"))  // the split in use for the last planned configuration (0: none yet)
    {
        *value = e->aq_last;
        return DDGI_OK;
    }
    if (!std::strcmp(name, "continued_workgroups"))  // frames in flight: workgroups that went on with a later update's rays, so far (synchronises)
    {
        uint32_t v = 0;
        if (e->d_work)
        {
            HIP_TRY(hipSetDevice(e->device));
            DDGI_TRY(ddgi_sync_stream(e, e->stream));
            HIP_TRY(hipMemcpy(&v, e->d_work + 4, sizeof v, hipMemcpyDeviceToHost));
        }
        *value = static_cast<int>(v);
        return DDGI_OK;
    }
    if (!std::strcmp(name, "texture_pairs"))  // pairs in the handle's ring
    {
        *value = e->np;
        return DDGI_OK;
    }
    if (!std::strcmp(name, "p2p_landing_zones") || !std::strcmp(name, "p2p_exported_mb"))  // peer-to-peer exchange: textures whose pushes land in zones (0: in the ring itself); MB a peer maps of this rank
    {
        int zones = 0, mb = 0;
        ddgi_exchange_p2p_info(e, &zones, &mb);
        *value = name[4] == 'l' ? zones : mb;
        return DDGI_OK;
    }
    if (!std::strcmp(name, "fast_march_active"))  // did the most recent update run the fast march ("fast_march" is a request)
    {
        *value = e->fast_march_active ? 1 : 0;
        return DDGI_OK;
    }
    for (const TuningKey& k : kTuningKeys)
        if (!std::strcmp(k.name, name))
        {
            *value = e->tuning.*(k.field);
            return DDGI_OK;
        }
    return fail(DDGI_ERR_INVALID_ARGUMENT, "unknown tuning key '%s'", name);
}

int ddgi_set_frame(ddgi_handle e, uint32_t frame)
{
    if (!e) return fail(DDGI_ERR_INVALID_ARGUMENT, "null handle");
    e->frame = frame;
    return DDGI_OK;
}

// REF mode: the per-texel table of sample_probe for the handle's current textures (k_sample_box_filter), rebuilt when they have
// changed since it was last built — one pass over the texels, about what sampling 100 000 points directly costs.
// The table costs 16 B per texel (4 x the albedo texture) and a pass over every texel; it pays when the batch would otherwise
// read more texels than that pass does: a point evaluates 8 x 26 texels directly, the pass about 26 per texel — so from
// texels / 8 points on (C3: 0.5 M points; never below 65 536).  *usable = false (and DDGI_OK): no table for this batch — too few
// points for this grid, or the table could not be allocated (a C5-sized grid's is 4.3 GB): the caller evaluates sample_probe per
// point, as it would for a small batch.
// (Measured, round 5: that count holds while the albedo texture stays in the caches — C3's 16 MB: the direct path costs 0.26 ns per point, the pass
// 11 ps per texel.  On a texture of hundreds of megabytes a point's 8 x 26 gathers miss and cost 1.55 ns (C4, 268 MB: 2.23 ms per 1.44 M points),
// six times as much, while the pass streams: there the table pays from texels / 64 points on — C4: 1.05 M points, one frame's pixels.)
static bool sample_box_pays(const ddgi_engine* e, size_t n_points)
{
    const size_t texels = e->tex_bytes[0] / 4;
    const size_t per_point = e->tex_bytes[0] > (static_cast<size_t>(64) << 20) ? 64 : 8;
    return n_points >= std::max<size_t>(65536, texels / per_point);
}
static int ensure_sample_box(ddgi_engine* e, const GridK& grid, bool* usable)
{
    *usable = false;
    // (one entry per texel and table slot: the table's 2x2x2 bricks round the probe counts up to even, ddgi_types.h)
    const size_t texels = static_cast<size_t>(box_slots(grid.cx, grid.cy, grid.cz)) * static_cast<size_t>(grid.n);
    if (texels > e->box_texels)
    {
        DDGI_TRY(ddgi_sync_stream(e, e->stream));
        if (e->d_box) (void)hipFree(e->d_box);
        e->d_box = nullptr, e->box_texels = 0, e->box_of = nullptr;
        if (hipMalloc(reinterpret_cast<void**>(&e->d_box), texels * sizeof(float4)) != hipSuccess)
        {
            (void)hipGetLastError();  // not an error of the call: the direct path needs no table
            e->d_box = nullptr;
            return DDGI_OK;
        }
        e->box_texels = texels;
    }
    if (e->box_of != e->tex[0])
    {
        HIP_TRY(launch_sample_box_filter(grid, static_cast<const uint32_t*>(e->tex[0]), e->d_box, e->num_cus, e->stream));
        e->box_of = e->tex[0];
    }
    *usable = true;
    return DDGI_OK;
}
// textures the host can write behind the handle's back — its own (ddgi_bind_textures), or the handle's through the pointers
// ddgi_device_textures gave out (an all-gather in place): their table is never reused
static void release_sample_box_if_borrowed(ddgi_engine* e)
{
    if (e->caller_tex || e->pin_pair) e->box_of = nullptr;
}

// try_box: the batch may go through the per-texel table (false: the caller — the chunk loop below — has found there is none)
static int sample_device(ddgi_engine* e, const float* d_pos, const float* d_nrm, size_t n, float* d_rgb, int32_t* d_cage, bool try_box);

int ddgi_sample_device(ddgi_handle e, const float* d_pos, const float* d_nrm, size_t n, float* d_rgb, int32_t* d_cage)
{
    if (!e) return fail(DDGI_ERR_INVALID_ARGUMENT, "null handle");
    return sample_device(e, d_pos, d_nrm, n, d_rgb, d_cage, true);
}

static int sample_device(ddgi_engine* e, const float* d_pos, const float* d_nrm, size_t n, float* d_rgb, int32_t* d_cage, bool try_box)
{
    e->tex_ops_since_update = true;  // (ddgi_exchange: something besides the update may be using the textures on the stream)
    if (n == 0) return DDGI_OK;
    if (!d_pos || !d_nrm || !d_rgb) return fail(DDGI_ERR_INVALID_ARGUMENT, "null device pointer");
    if (n > 0xffffffffull) return fail(DDGI_ERR_INVALID_ARGUMENT, "too many points");
    HIP_TRY(hipSetDevice(e->device));
    if (int rc = ddgi_exchange_wait_latest(e)) return rc;
    SampleArgs a{};
    a.grid = make_grid(e);
    a.albedo = static_cast<const uint32_t*>(e->tex[0]);
    a.distance = static_cast<const uint32_t*>(e->tex[1]);
    a.pos = d_pos;
    a.nrm = d_nrm;
    a.rgb = d_rgb;
    a.cage = d_cage;
    a.n = static_cast<uint32_t>(n);
    // REF mode, a large batch (or the table is there already): sample_probe comes from its per-texel table — built now if the
    // textures have changed since it was last built (one pass over the texels, ~ the cost of sampling 100 000 points directly)
    if (try_box && e->mode == DDGI_MODE_REF && e->tuning.sample_box && (sample_box_pays(e, n) || e->box_of == e->tex[0]))
    {
        bool usable = false;
        if (int rc = ensure_sample_box(e, a.grid, &usable)) return rc;
        if (usable)
        {
            a.box = e->d_box;
            HIP_TRY(launch_probe_sample_ref(a, e->stream));  // 8 table entries per point: nothing to gain from grouping
            release_sample_box_if_borrowed(e);
            return DDGI_OK;
        }
    }
    // The grouping kernels count a contiguous run of the batch per workgroup in 16-bit LDS counters (256 runs of fewer than 65 536
    // points): a larger batch is sampled in pieces of at most kGroupChunk points, one after the other on the stream.
    constexpr size_t kGroupChunk = (static_cast<size_t>(1) << 24) - 65536;
    if (e->tuning.sample_group && n > kGroupChunk)
    {
        for (size_t off = 0; off < n; off += kGroupChunk)
        {
            const size_t m = std::min(kGroupChunk, n - off);
            // (whether the per-texel table serves this batch was decided above, once: a table that could not be allocated is not
            // asked for again per chunk — a stream synchronisation and a failing multi-GB allocation each time)
            if (int rc = sample_device(e, d_pos + 3 * off, d_nrm + 3 * off, m, d_rgb + 3 * off, d_cage ? d_cage + 8 * off : nullptr, false)) return rc;
        }
        return DDGI_OK;
    }
    if (e->tuning.sample_group && n >= 4096)  // a batch worth grouping by cage (small ones are launch-latency bound anyway)
    {
        const size_t words = sample_group_scratch_words(a.n, static_cast<uint32_t>(a.grid.cx) * a.grid.cy * a.grid.cz);
        if (words > e->sample_scratch_words)
        {
            DDGI_TRY(ddgi_sync_stream(e, e->stream));  // (an earlier batch may still be using the old scratch)
            if (e->d_sample_scratch) (void)hipFree(e->d_sample_scratch);
            e->d_sample_scratch = nullptr, e->sample_scratch_words = 0;
            HIP_TRY(hipMalloc(reinterpret_cast<void**>(&e->d_sample_scratch), words * sizeof(uint32_t)));
            e->sample_scratch_words = words;
        }
        HIP_TRY(launch_sample_grouping(a.grid, d_pos, a.n, e->d_sample_scratch, &a.perm, &a.perm_off, e->stream));
    }
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
    e->tex_ops_since_update = true;  // (ddgi_exchange: something besides the update may be using the textures on the stream)
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
    if (int rc0 = ddgi_exchange_wait_latest(e)) return rc0;
    if (int rc0 = peers_done_before_host_copies(e)) return rc0;
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
    rc = ddgi_sync_stream(e, e->stream);
#undef TRY_OR_CLEAN
    if (rc == DDGI_ERR_TIMEOUT) return rc;  // (the device buffers are LEFT: whatever still stands on the stream may write them)
    cleanup();
    return rc;
}

int ddgi_render_device(ddgi_handle e, const ddgi_camera* cam, const ddgi_render_settings* st, uint32_t* d_rgba8, float* d_rgb_f32)
{
    if (!e || !cam || !st || !d_rgba8) return fail(DDGI_ERR_INVALID_ARGUMENT, "null handle/camera/settings/output");
    e->tex_ops_since_update = true;  // (ddgi_exchange: something besides the update may be using the textures on the stream)
    if (st->scene < 0 || st->scene > 3) return fail(DDGI_ERR_INVALID_ARGUMENT, "scene %d not in {0,1,2,3}", st->scene);
    if (st->screen_width < 1 || st->screen_height < 1 || 1ll * st->screen_width * st->screen_height > 0x7fffffffll)
        return fail(DDGI_ERR_INVALID_ARGUMENT, "bad image size %d x %d", st->screen_width, st->screen_height);
    if (st->camera_mode != 0 && st->camera_mode != 1) return fail(DDGI_ERR_UNSUPPORTED, "camera_mode %d (only 0 pinhole, 1 ortho)", st->camera_mode);
    HIP_TRY(hipSetDevice(e->device));
    const int scene = st->scene;
    if (int rc = ensure_scene(e, scene)) return rc;
    if (int rc = ensure_noise(e)) return rc;
    if ((256 + static_cast<size_t>(e->dev_scene[scene].k.nwords)) * sizeof(uint32_t) > 160 * 1024)
        return fail(DDGI_ERR_UNSUPPORTED, "ddgi_render: the scene's occupancy bitmap (%d words) does not fit in the 160 KB of LDS", e->dev_scene[scene].k.nwords);
    if (int rc = ddgi_exchange_wait_latest(e)) return rc;
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
    r.visualize_probes = st->visualize_probes;
    r.width = st->screen_width;
    r.height = st->screen_height;
    if (e->mode == DDGI_MODE_DDGI)
    {
        r.irradiance = static_cast<const float*>(e->tex[0]);
        r.depth = static_cast<const float*>(e->tex[1]);
    }
    else
    {
        r.albedo = static_cast<const uint32_t*>(e->tex[0]);
        // only for the views that read the probe field (integrator_DDGI, integrator_indirect, the cage colours), and when the frame's
        // pixels are worth a pass over the grid's texels (sample_box_pays) or the table is there already
        const size_t pixels = static_cast<size_t>(st->screen_width > 0 ? st->screen_width : 0) * (st->screen_height > 0 ? st->screen_height : 0);
        if (e->tuning.sample_box && !(st->render_mode == 1 || (st->render_mode >= 3 && st->render_mode <= 6)) && (sample_box_pays(e, pixels) || e->box_of == e->tex[0]))
        {
            bool usable = false;
            if (int rc = ensure_sample_box(e, r.trace.grid, &usable)) return rc;
            if (usable) r.box = e->d_box;
        }
    }
    r.rgba8 = d_rgba8;
    r.rgb_f32 = d_rgb_f32;
    HIP_TRY(launch_render_primary(r, e->stream));
    if (r.box) release_sample_box_if_borrowed(e);
    return DDGI_OK;
}

int ddgi_render(ddgi_handle e, const ddgi_camera* cam, const ddgi_render_settings* st, uint8_t* rgba8, float* rgb_f32)
{
    if (!e || !cam || !st || !rgba8) return fail(DDGI_ERR_INVALID_ARGUMENT, "null handle/camera/settings/output");
    e->tex_ops_since_update = true;  // (ddgi_exchange: something besides the update may be using the textures on the stream)
    HIP_TRY(hipSetDevice(e->device));
    const size_t n = static_cast<size_t>(st->screen_width > 0 ? st->screen_width : 0) * (st->screen_height > 0 ? st->screen_height : 0);
    uint32_t* d_img = nullptr;
    float* d_f = nullptr;
    if (n == 0) return fail(DDGI_ERR_INVALID_ARGUMENT, "bad image size");
    if (int rc0 = ddgi_exchange_wait_latest(e)) return rc0;
    if (int rc0 = peers_done_before_host_copies(e)) return rc0;
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
    if (rc == DDGI_OK && he == hipSuccess) rc = ddgi_sync_stream(e, e->stream);
    if (rc == DDGI_ERR_TIMEOUT) return rc;  // (the device image is LEFT: whatever still stands on the stream may write it)
    (void)hipFree(d_img);
    if (d_f) (void)hipFree(d_f);
    if (rc != DDGI_OK) return rc;
    if (he != hipSuccess) return fail(DDGI_ERR_HIP, "render readback failed: %s", hipGetErrorString(he));
    return DDGI_OK;
}

int ddgi_set_stream(ddgi_handle e, void* hip_stream)
{
    if (!e) return fail(DDGI_ERR_INVALID_ARGUMENT, "null handle");
    e->tex_ops_since_update = true;  // (ddgi_exchange: something besides the update may be using the textures on the stream)
    HIP_TRY(hipSetDevice(e->device));
    DDGI_TRY(ddgi_sync_stream(e, e->stream));
    if (e->own_stream) (void)hipStreamDestroy(e->stream);
    e->own_stream = false;
    e->stream = static_cast<hipStream_t>(hip_stream);
    e->chain_break = true;
    return DDGI_OK;
}

int ddgi_device_textures(ddgi_handle e, void** tex0, size_t* tex0_bytes, void** tex1, size_t* tex1_bytes,
                         size_t* slab_offset0, size_t* slab_bytes0, size_t* slab_offset1, size_t* slab_bytes1)
{
    if (!e) return fail(DDGI_ERR_INVALID_ARGUMENT, "null handle");
    e->tex_ops_since_update = true;  // (ddgi_exchange: something besides the update may be using the textures on the stream)
    // The host may keep these pointers (an all-gather of its own in place, Vulkan interop): from here on the handle stays on
    // this pair — no frames in flight — unless the pipelined exchange alternates pairs by contract (include/ddgi_probe.h); and
    // what the host writes through them is invisible to the sampler's per-texel table, which is therefore not reused.
    if (!e->xch.pipelined) e->pin_pair = true;
    e->chain_break = true;
    e->box_of = nullptr;
    // A pinned handle writes one pair only: the other pairs of a frames-in-flight ring would be memory held for nothing (eight pairs
    // by default) — the ring shrinks to what a pinned handle uses BEFORE the pointers go out (blocks once; the textures carry over).
    if (e->pin_pair && !e->caller_tex && !e->xch.transport && !e->xch.p2p && e->np > ddgi_pairs_wanted(e, false))
    {
        HIP_TRY(hipSetDevice(e->device));
        if (int rc = ddgi_resize_ring(e, ddgi_pairs_wanted(e, false))) return rc;
    }
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
    e->tex_ops_since_update = true;  // (ddgi_exchange: something besides the update may be using the textures on the stream)
    if ((tex0 == nullptr) != (tex1 == nullptr)) return fail(DDGI_ERR_INVALID_ARGUMENT, "bind both textures or neither");
    if (e->xch.pipelined || e->xch.p2p) return fail(DDGI_ERR_INVALID_ARGUMENT, "the pipelined / peer-to-peer exchange owns the texture pairs: ddgi_exchange_init(h, NULL, 0) first");
    HIP_TRY(hipSetDevice(e->device));
    DDGI_TRY(ddgi_sync_stream(e, e->stream));
    e->caller_tex = tex0 != nullptr;
    e->tex[0] = tex0 ? tex0 : ddgi_pair_ptr(e, e->pair_cur, 0);
    e->tex[1] = tex1 ? tex1 : ddgi_pair_ptr(e, e->pair_cur, 1);
    e->tex_prev[0] = e->tex_prev[1] = nullptr;
    e->chain_break = true;
    e->box_of = nullptr;
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

int ddgi_generate_probe_rays_host_tile(const ddgi_irradiance_field* f, int tx, int ty, uint32_t seed, int skip_calls, ddgi_probe_ray* rays, size_t n)
{
    if (!f || !rays) return fail(DDGI_ERR_INVALID_ARGUMENT, "null field/rays");
    ddgi_render_settings st{};
    if (int rc = validate_config(f, &st, 1)) return rc;
    if (int rc = validate_tile(f, tx, ty)) return rc;
    const size_t expect = static_cast<size_t>(f->probe_count[0]) * f->probe_count[1] * f->probe_count[2] * tx * ty;
    if (n != expect) return fail(DDGI_ERR_INVALID_ARGUMENT, "expected room for %zu rays, got %zu", expect, n);
    GlibcRand rng;
    rng.seed(seed);
    // each generate_probe_rays() call draws 2*s*s values (rvpt.cpp:1161-1162)
    for (long long i = 0; i < 2ll * skip_calls * tx * ty; ++i) (void)rng.next();
    std::vector<ddgi_probe_ray> tmp;
    generate_probe_rays(*f, tx, ty, rng, tmp);
    std::memcpy(rays, tmp.data(), tmp.size() * sizeof(ddgi_probe_ray));
    return DDGI_OK;
}

int ddgi_generate_probe_rays_host(const ddgi_irradiance_field* f, uint32_t seed, int skip_calls, ddgi_probe_ray* rays, size_t n)
{
    if (!f) return fail(DDGI_ERR_INVALID_ARGUMENT, "null field/rays");
    return ddgi_generate_probe_rays_host_tile(f, f->sqrt_rays_per_probe, f->sqrt_rays_per_probe, seed, skip_calls, rays, n);
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
    e->scene_epoch += 1;  // the trace kernel's wave split is measured again for the new scene
    e->chain_break = true;
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
    DDGI_TRY(ddgi_sync_stream(e, e->stream));
    ddgi_engine::DevScene& d = e->dev_scene[3];
    if (d.bits) (void)hipFree(d.bits);
    if (d.types) (void)hipFree(d.types);
    if (d.skip) (void)hipFree(d.skip);
    free_vis_tables(d);
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

int ddgi_scene_skip_field(int scene, int32_t lo[3], int32_t dim[3], uint8_t* codes, size_t capacity)
{
    if (scene < 0 || scene > 2 || !lo || !dim) return fail(DDGI_ERR_INVALID_ARGUMENT, "scene %d not in {0,1,2} / null output", scene);
    const SceneBake& b = baked_scene(scene);
    for (int a = 0; a < 3; ++a) lo[a] = b.lo[a], dim[a] = b.dim[a];
    if (!codes) return DDGI_OK;  // (size query)
    if (capacity < b.types.size()) return fail(DDGI_ERR_INVALID_ARGUMENT, "capacity %zu < %zu voxels", capacity, b.types.size());
    std::vector<uint32_t> words;
    build_skip_field(b, 0, words);
    for (size_t i = 0; i < b.types.size(); ++i) codes[i] = static_cast<uint8_t>((words[i >> 4] >> ((i & 15) * 2)) & 3u);
    return DDGI_OK;
}

float ddgi_pinned_sinf(float x) { return pm::sinf_pinned(x); }
float ddgi_pinned_cosf(float x) { return pm::cosf_pinned(x); }
float ddgi_pinned_acosf(float x) { return pm::acosf_pinned(x); }
int ddgi_pinned_sincos_small(float x, float* s, float* c)
{
    float sn, cs;
    pm::sincos_small(x, sn, cs);
    if (s) *s = sn;
    if (c) *c = cs;
    return DDGI_OK;
}

}  // extern "C"
