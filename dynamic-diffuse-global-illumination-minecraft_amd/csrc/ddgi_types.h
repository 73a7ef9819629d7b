// ddgi_types.h — host/device shared argument blocks of the probe-path kernels.
#pragma once

#include <cstddef>
#include <cstdint>

#include <hip/hip_runtime.h>

#include "../../include/ddgi_probe.h"
#include "ddgi_scene.h"

namespace ddgi {

constexpr int kMaxLights = DDGI_MAX_LIGHTS;
constexpr int kMarchIters = 125;  // grid_march's loop bound, intersection.glsl:1059
constexpr int kTraceBlock = 256;  // threads per workgroup = 4 wave64
constexpr int kWfColdBytes = 32;  // wavefront trace kernels: global scratch per pool slot (ddgi_trace_wf.hip: WfColdGlobal)

struct LightK
{
    float intensity;
    float col[3];
    float pos[3];
};

// A baked scene (ddgi_scene_bake.cpp): 1 bit/voxel occupancy + 1 byte/voxel block type over an
// inclusive voxel-id box that contains everything that is not an extrusion of its border layer.
struct SceneK
{
    int lo[3];
    int hi[3];
    int nx;   // cells per row (x extent)
    int nxy;  // cells per z-slice (x extent * y extent)
    int bias;    // (lo.z*ny + lo.y)*nx + lo.x : block-type index = z*nxy + y*nx + x - bias
    int bias32;  // bias rounded down to a multiple of 32; the occupancy bitmap is stored shifted by
                 // (bias - bias32) bits so that bit (raw & 31) of word (raw >> 5) - (bias32 >> 5) is voxel raw
    float lo_f[3], hi_f[3], nx_f, nxy_f;  // lo, hi, nx, nxy as floats (exact small integers) for march_step
    int nwords;            // 32-bit words in the occupancy bitmap
    unsigned face_empty;   // bit (2*axis + side): that border layer is entirely empty
    const uint32_t* bits;  // device
    const uint8_t* types;  // device
    const uint32_t* skip;  // device: the fast march's skip field, 2 bits per voxel, addressed like `bits` (ddgi_host.h: build_skip_field)
    int nwords_skip;       // 32-bit words in it (16 voxels each)
};

// Probe-grid geometry shared by all kernels.
struct GridK
{
    int cx, cy, cz;  // probe counts
    int sx, sy;      // ray tile: sx strata along z (texel columns) x sy strata along phi (texel rows); the reference's
                     // tile is square (sx == sy == sqrt_rays_per_probe), ddgi_set_ray_tile makes it sx x sy
    int n;           // rays per probe = sx * sy
    int side;        // integer probe spacing
    float origin[3];
    float hysteresis;
    int z0, czl;  // this rank's z-slab: probes with z in [z0, z0+czl)
};

// ---- the REF sampler's per-texel table (k_sample_box_filter, ddgi_sampler.h): where a probe's entry sits inside a texel's plane ----
// DDGI_BOX_LAYOUT  0: tile-major [slab slot][texel];  1: texel-major [texel][slab slot];  2: texel-major in 2x2x2 BRICKS of probes —
// the 8 entries of the probes (x..x+1, y..y+1, z..z+1), x, y, z even, are ONE 128-byte line (a 2x2 z-layer per 64-byte half), so a cage's
// 8 entries lie in 1.5^3 = 3.4 lines on average instead of 4.5 (slab order: 4 x-pairs of 32 bytes, an eighth of them across a line end).
#ifndef DDGI_BOX_LAYOUT
#define DDGI_BOX_LAYOUT 2
#endif
// table slots per texel plane (bricks: the counts rounded up to even)
inline __host__ __device__ uint32_t box_slots(int cx, int cy, int cz)
{
    return DDGI_BOX_LAYOUT >= 2 ? 8u * static_cast<uint32_t>((cx + 1) / 2) * static_cast<uint32_t>((cy + 1) / 2) * static_cast<uint32_t>((cz + 1) / 2)
                                : static_cast<uint32_t>(cx) * static_cast<uint32_t>(cy) * static_cast<uint32_t>(cz);
}
// the table slot of probe (x, y, z) (z: the whole grid's, as in a slab slot (z cy + y) cx + x)
inline __host__ __device__ uint32_t box_slot_xyz(int cx, int cy, int x, int y, int z)
{
    if (DDGI_BOX_LAYOUT < 2) return static_cast<uint32_t>((z * cy + y) * cx + x);
    const int cxh = (cx + 1) >> 1, cyh = (cy + 1) >> 1;
    return static_cast<uint32_t>((((z >> 1) * cyh + (y >> 1)) * cxh + (x >> 1)) * 8 + ((z & 1) << 2 | (y & 1) << 1 | (x & 1)));
}
// ... and back: table slot -> slab slot, or -1 for a slot of the padding (odd counts)
inline __host__ __device__ int box_slot_to_slab_slot(int cx, int cy, int cz, uint32_t b)
{
    if (DDGI_BOX_LAYOUT < 2) return static_cast<int>(b);
    const int cxh = (cx + 1) >> 1, cyh = (cy + 1) >> 1;
    const int sub = static_cast<int>(b & 7u), br = static_cast<int>(b >> 3);
    const int x = 2 * (br % cxh) + (sub & 1), y = 2 * ((br / cxh) % cyh) + ((sub >> 1) & 1), z = 2 * (br / (cxh * cyh)) + (sub >> 2);
    return (x < cx && y < cy && z < cz) ? (z * cy + y) * cx + x : -1;
}

// Opts a kernel in to `bytes` of dynamic LDS (more than the 64 KB default) on the CURRENT device, once per
// (device, kernel): a process may hold handles on several devices (ddgi_engine.cpp).
hipError_t ensure_dynamic_lds(const void* kernel, int bytes);

// k_light_visibility's classes (ddgi_visibility.hip): what a light feeler that starts in a voxel is certain to find
enum : uint8_t
{
    kVisUnknown = 0,  // march it
    kVisLit = 1,      // reaches the light
    kVisShadow = 2,   // lands in an occupied voxel before the light
    kVisListed = 3,   // neither for the whole start patch, but the bundle holds at most kVisListMax occupied voxels: they are listed,
                      // and the event tests the feeler's own ray against them (wf_event: listed_feeler_outcome)
};
constexpr int kVisListMax = 4;
constexpr int kVisLights = 4;  // lights that get a table (the commented 4-light cave table of the reference, structs.glsl:65-68); further lights' feelers are marched
constexpr uint32_t kVisListEnd = 0xffffffffu;
// a listed voxel: its id minus the start voxel's, per axis, biased by 512 into 10 bits (the table's range is 400 voxels)
inline __host__ __device__ uint32_t vis_pack_offset(int dx, int dy, int dz) { return static_cast<uint32_t>(dx + 512) | (static_cast<uint32_t>(dy + 512) << 10) | (static_cast<uint32_t>(dz + 512) << 20); }

struct TraceArgs
{
    GridK grid;
    SceneK scene;
    int scene_id;
    int max_bounces;
    int nl;
    LightK lights[kMaxLights];
    const float4* rays;   // local slab's ProbeRay records (3 x float4 each), reference p order
    uint32_t n_rays;      // local ray count
    uint32_t* albedo;     // full-grid slab-major rgba8 texels [cz][cy][cx][s][s]
    uint32_t* distance;   // same shape
    int wait_threshold;   // event batching: handle finished marches once this many lanes wait
    NoiseLut noise;       // memoised lattice hashes (device pointers)
    unsigned long long* stats;  // profiling aid (null = off): [0] march-loop trips, [1] lane-steps, [2] event rounds, [3] lane-events, [4] waves
    void* wf_cold;        // wavefront kernel: per-slot shading state, grid x pool x kWfColdBytes (device scratch)
    float4* wf_dir;       // wavefront kernel: per-slot accumulated direct light (only with > 1 light)
    int wf_tail;          // wavefront kernel: straggler steps after the march list is drained (0 = default)
    int wf_fetch;         // wavefront kernel: idle lanes that trigger a task fetch (0 = default)
    int wf_drain;         // wavefront kernel: straggler trips while the pool drains (0 = default)
    int wf_chunk;         // wavefront kernel: rays a workgroup claims at a time (set by the launcher)
    int ablate;           // profiling build only (tuning "ablate"; 0 = exact): 1 constant albedo, 2 constant bounce direction, 4 no dead-feeler elimination, 8 queue kernel: posted marches are dropped (tests the safety net), 16 count feeler classes into stats (slow)
    // DDGI mode (ddgi != 0): rays are generated in the kernel (spherical Fibonacci set rotated by
    // rot, origin = probe position), the RNG seed is ray index ^ frame_key, and each ray writes
    // its radiance and clamped first-hit distance to the ray records (below) instead of an rgba8 texel
    int ddgi;
    uint32_t frame_key;
    float rot[9];
    float* rad_rgb;  // ray records, see RayRecords
    float* rad_dd;
    int fast_march;      // tolerance mode (ddgi_set_tuning "fast_march"): marches skip empty space (ddgi_device.h: fast_march_step)
    const uint8_t* vis;  // single light: feeler classes per (voxel of the baked box, face): [voxel * 8 + face] (k_light_visibility), or null
    const uint32_t* vis_occ;  // ... and for class kVisListed the occupied voxels of the bundle: [(voxel * 8 + face) * kVisListMax + k]
    const uint8_t* vis_more[kVisLights - 1];  // several lights: the same table for lights 1 .. kVisLights - 1 (classes kVisLit / kVisShadow only), or null
    uint32_t pair_words;  // REF mode, frames in flight: 32-bit words between two consecutive texture pairs of the handle's ring (albedo + k * pair_words
                          // is the pair k updates after this launch's own); 0: the launch writes its own pair only (ddgi_trace_wf.hip: wf_finish_ray)
};

// k_probe_trace_aq's place in the handle's sequence of launches (ddgi_engine.cpp: "frames in flight").  Launch `seq` claims its
// rays from counters[seq % kAqCounters]; when they are used up and the host has published launch seq + 1 as a CONTINUATION (the same
// work into the next texture pair: pub[(seq + 1) % kAqPubRing] == seq + 2), the launch's workgroups go on with that update's
// rays instead of draining — up to chain_max updates ahead.
constexpr uint32_t kAqPubRing = 64;  // entries of the host's ring of published continuations (pinned host memory)
constexpr int kAqChainMax = 8;       // updates one launch can work on (its own + 7): the counters ring holds 2 x that
constexpr uint32_t kAqCounters = 2 * kAqChainMax;
// What may DIFFER between two updates that one launch works on (round 5): the lights (ddgi_set_lights, or animated by
// RenderSettings::time — probe_pass.comp:217-251), DDGI mode's per-frame ray rotation and RNG key, where the rays come from
// and where the ray records go, and the light-feeler classes that belong to the lights' positions.  The queue kernel reads these
// from a ring indexed by the update's sequence number instead of from its own arguments: a ray carries in dst[31:29] which
// update after the launch's own it belongs to, an event group runs with the record of ITS update (k_probe_trace_aq).
// The host writes an update's record into pinned memory before it publishes the update (AqChain::upd_host); a workgroup copies it
// into the device ring (AqChain::upd_dev) when it starts (its launch's own update) / when it moves on to the update (a continued
// one) and reads it from there with scalar loads.  The feeler classes a record points at must be complete before the first launch
// that may read it starts: ddgi_engine.cpp: assign_vis.
struct UpdK
{
    LightK lights[kMaxLights];
    float rot[9];
    uint32_t frame_key;
    const float4* rays;
    float* rad_rgb;
    float* rad_dd;
    const uint8_t* vis;
    const uint32_t* vis_occ;
    const uint8_t* vis_more[kVisLights - 1];
    uint32_t pad[128 - 7 * kMaxLights - 10 - 2 * (5 + kVisLights - 1)];
};
constexpr uint32_t kUpdWords = 128;
static_assert(sizeof(UpdK) == kUpdWords * 4, "one record of the per-update ring is 128 dwords: a wave copies it with two dwords per lane");
static_assert(offsetof(UpdK, rays) % 8 == 0, "pointers of the record are read as 64-bit words");

struct AqChain
{
    uint32_t* counters;   // device: kAqCounters ray counters
    uint32_t* continued;  // device: counts the workgroups that went on with a later update (ddgi_get_tuning "continued_workgroups")
    const uint32_t* pub;  // pinned host memory, device-visible: kAqPubRing words
    uint32_t seq;
    uint32_t chain_max;   // 0: this launch works on its own update only
    const uint32_t* upd_host;  // pinned host memory, device-visible: kAqPubRing records (UpdK) — [seq % kAqPubRing]
    uint32_t* upd_dev;         // device: kAqCounters records — [seq % kAqCounters]
};

// DDGI-mode ray records, laid out as the B operand of the blend's MFMA contraction (ddgi_blend_sample.hip): for local
// probe pl (the slab's (y, zl, x) enumeration) and ray i of n_pad (n rounded up to kRecRayPad; the padding stays zero)
//     rad_dd [(pl >> 4) * n_pad * 32 + mfma_operand_offset(i, ch * 16 + (pl & 15))]   ch: 0 = d, 1 = d * d,  d = min(first-hit distance, 1.5 * spacing)
//     rad_rgb[((pl >> 5) * 3 + ch) * n_pad * 32 + mfma_operand_offset(i, pl & 31)]    ch: r, g, b
// One v_mfma_f32_32x32x2_f32 consumes, per lane l, the operand element (k = ray 2q + (l >> 5), column l & 31) of ray
// pair q.  mfma_operand_offset stores, for every lane, the elements of FOUR consecutive ray pairs next to each other:
// a lane fetches 4 MFMAs' worth of operand with one 16-byte load and a wave reads 1 KB contiguous.
constexpr uint32_t kRecRayPad = 32;  // 16 ray pairs: the chunk the blend kernels stage and prefetch
inline __host__ __device__ uint32_t rec_ray_pad(uint32_t n) { return (n + kRecRayPad - 1) / kRecRayPad * kRecRayPad; }
// element (k index i, row-or-column j in [0, 32)) of a 32-wide MFMA operand stream: [i / 8][lane = (i & 1) * 32 + j][(i / 2) & 3]
inline __host__ __device__ size_t mfma_operand_offset(uint32_t i, uint32_t j)
{
    return ((static_cast<size_t>(i >> 3) * 64 + (i & 1u) * 32 + j) * 4) + ((i >> 1) & 3u);
}
inline __host__ __device__ size_t rec_dd_index(uint32_t pl, uint32_t i, uint32_t n_pad, uint32_t ch)
{
    return static_cast<size_t>(pl >> 4) * n_pad * 32 + mfma_operand_offset(i, ch * 16 + (pl & 15u));
}
inline __host__ __device__ size_t rec_rgb_index(uint32_t pl, uint32_t i, uint32_t n_pad, uint32_t ch)
{
    return (static_cast<size_t>(pl >> 5) * 3 + ch) * n_pad * 32 + mfma_operand_offset(i, pl & 31u);
}
inline size_t rec_dd_floats(uint32_t n_local_probes, uint32_t n) { return static_cast<size_t>((n_local_probes + 15u) / 16u) * rec_ray_pad(n) * 32; }
inline size_t rec_rgb_floats(uint32_t n_local_probes, uint32_t n) { return static_cast<size_t>((n_local_probes + 31u) / 32u) * 3 * rec_ray_pad(n) * 32; }

// k_probe_blend: per-probe update of the octahedral irradiance / depth-moment tiles
struct BlendArgs
{
    GridK grid;
    float rot[9];
    const float* rad_rgb;  // ray records of the local slab (probes in (y, zl, x) order), see rec_rgb_index / rec_dd_index
    const float* rad_dd;
    float* irradiance;       // full-grid slab-major [cz][cy][cx][8][8][4]: the tiles this update writes
    float* depth;            // full-grid slab-major [cz][cy][cx][16][16][2]
    const float* irradiance_old;  // the previous update's tiles the hysteresis mixes with: the same buffers, or — when the
    const float* depth_old;       // multi-GPU exchange is pipelined (ddgi_exchange.cpp) — the other buffer pair
    uint32_t n_local_probes;
    float* w;      // per-update texel weights as MFMA A tiles (k_blend_weights; ddgi_blend_sample.hip: weight_index), or null
    float* w_sum;  // [256] weight sum per texel column: [0,196) depth texels, [196,232) irradiance texels
    uint32_t merge_below;     // HALF depth groups per CU up to which depth and irradiance run as ONE launch (k_probe_blend_mfma); tuning "blend_merge"
    uint32_t force_division;  // 1: the MFMA kernels divide with the compiler's `/` everywhere (what they do for a group whose sums lie
                              // outside pm::div_prepared's domain) — tuning "blend_kernel" 2, the cross-check of that path
};

#ifndef DDGI_SAMPLE_COHERENCE
#define DDGI_SAMPLE_COHERENCE 1  // the grouping kernels notice a batch that comes in cage order and write no permutation for it (k_sample_place)
#endif
struct SampleArgs
{
    GridK grid;
    const uint32_t* albedo;
    const uint32_t* distance;
    const float* pos;  // n*3
    const float* nrm;  // n*3
    float* rgb;        // n*3
    int32_t* cage;     // n*8 or null
    uint32_t n;
    const float* irradiance;  // DDGI mode tiles (slab-major), else null
    const float* depth;
    const uint32_t* perm;     // processing order: lane k handles point perm[k] (points grouped by cage, see k_sample_*), or null
    const uint32_t* perm_off; // with perm: *perm_off != 0 = the batch came in cage order already (k_sample_place found so and wrote no permutation): perm is not used
    const float4* box;        // REF mode: sample_probe tabulated per texel (k_sample_box_filter), or null: evaluate it per point
};

// k_render_primary: camera rays + integrators over the probe field
struct RenderArgs
{
    TraceArgs trace;       // scene, lights, noise, grid (rays/textures of the trace kernels unused)
    float cam_matrix[16];  // Camera UBO: mat4, column-major (compute_pass.comp:30-35)
    float cam_params[4];   // aspect, hfov, ortho scale, 0
    float pinhole_w;       // 1 / tan(hfov / 2), pinned tan = sin / cos, evaluated on the host
    int camera_mode;       // 0 pinhole, 1 ortho
    int render_mode;       // integrator index, compute_pass.comp:58-87; 6 probe-texture blit, 7 cage-index colours (debug views)
    int visualize_probes;  // RenderSettings::visualize_probes: probes drawn as spheres (integrators.glsl:45-65)
    int width, height;
    const uint32_t* albedo;   // REF mode probe texture (slab-major)
    const float4* box;        // ... and sample_probe tabulated per texel of it (k_sample_box_filter), or null
    const float* irradiance;  // DDGI mode tiles; null in REF mode
    const float* depth;
    uint32_t* rgba8;  // width*height
    float* rgb_f32;   // optional, width*height*3
};

}  // namespace ddgi
