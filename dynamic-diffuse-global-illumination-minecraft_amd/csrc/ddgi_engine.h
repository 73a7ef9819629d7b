// ddgi_engine.h — the handle behind include/ddgi_probe.h (private to the library's host side:
// ddgi_engine.cpp = configuration, launches, outputs; ddgi_exchange.cpp = the multi-GPU exchange).
#pragma once

#include <hip/hip_runtime.h>

#include <map>
#include <string>
#include <vector>

#include "ddgi_host.h"
#include "ddgi_scene.h"
#include "ddgi_types.h"

using namespace ddgi;  // (private header of the library's own translation units)

// Tuning / diagnostic switches of a handle (ddgi_set_tuning / ddgi_get_tuning).  The environment variables of
// the same meaning are read ONCE, when the handle is created — never on the per-frame path.
struct Tuning
{
    int trace_kernel = 0;   // 0 auto (queues when the pool fits), 1 rounds, 2 ray per lane, 3 queues even with counters on
    int march_waves = 0;    // queue kernel: waves that march; 0 = per configuration (measured, see "autotune")
    int autotune = 0;       // 1: measure the march/event split on the first update of a configuration (that update BLOCKS the host
                            // once); default off — ddgi_probe_update never blocks, ddgi_tune() measures on request
    int blend_kernel = 0;   // DDGI blend: 0 auto, 1 one probe per workgroup (cross-check), 2 auto with the compiler's division (cross-check)
    int aq_pool = 0, wf_pool = 0, wf_maxpool = 0, wf_threads = 1024;
    int wf_fetch = 0, wf_tail = 0, wf_chunk = 0, wf_drain = 0, wait_threshold = 64;
    int blend_merge = 1;    // DDGI blend: up to this many HALF depth groups (of 16 probes) per CU, depth and irradiance run as one launch
    int frames_in_flight = 8;  // REF mode: the MOST probe updates one launch may work on (how many it does is up to the host: an update is continued only
                               // if it was submitted while its predecessor still ran — the reference's host runs MAX_FRAMES_IN_FLIGHT = 2 ahead,
                               // src/rvpt/rvpt.h:23, a loop that never waits runs as far ahead as this allows): such an update, with the same inputs, is
                               // traced by the predecessor's workgroups instead of waiting for their drain (ddgi_engine.cpp: ddgi_probe_update;
                               // k_probe_trace_aq); 1: every launch traces its own update only
    int timing = 1;         // per-update events for ddgi_last_update_ms / ddgi_update_history_ms (0: none — saves the stream ~6 us per update)
    int fast_march = 0;     // tolerance mode: marches skip empty space (NOT bit-exact; tests/test_gpu_fast_march.py states the tolerance)
    int light_vis = 1;      // per-voxel light-feeler classes (k_light_visibility): 0 = march every feeler
    int sample_box = 1;     // REF ddgi_sample*: large batches go through the per-texel table of sample_probe (0: every point evaluates its 8 x 26 texels)
    int sample_group = 1;   // ddgi_sample*: handle the points of a batch cage by cage (0: in the order given)
    int noise_lut = 1;      // memoised lattice hashes (0: compute every hash)
    int lut_off = 0;        // profiling: 1 = no wall table, 2 = no random1 table
    int reserve_cus = 0;    // the queue kernel leaves this many CUs without a workgroup — its persistent workgroups fill a CU's registers and LDS, so a copy or
                            // collective KERNEL of a multi-GPU exchange otherwise finds no CU until a launch ends (copy-engine transfers need none)
    int prep_stream = 2;    // DDGI mode: the next frame's weight tiles and the predicted light-feeler tables are made on a second stream beside the blend (0: in line;
                            // 2: the tables on a THIRD stream where the blend is the small merged kernel — a rank's slab —, 3: always, 1: never)
    int wait_timeout_ms = 10000;  // the deadline of every host wait of a handle with a multi-GPU exchange attached (ddgi_exchange.cpp: ddgi_sync_stream); 0: none
    int verbose = 0;
    int ablate = 0;         // profiling build only (-DDDGI_PROFILING): ablations / fault injection
};

// ---- error reporting ---------------------------------------------------------------------------------

// Records the calling thread's last error message (ddgi_last_error) and returns `code`.
int ddgi_fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
#define fail ddgi_fail

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
    int tile[2] = {0, 0};  // ddgi_set_ray_tile: non-square ray tile (0 = the field's sqrt_rays_per_probe)
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
        uint32_t* skip = nullptr;           // the fast march's skip field (ddgi_host.h: build_skip_field)
        SceneK k{};
        bool ready = false;
        // light-feeler classes (k_light_visibility): kAqChainMax SETS of tables, each for one position of the scene's lights — the one
        // the latest update uses and, with frames in flight and lights that move (DDGI mode: update_lights(time)), the ones
        // PREDICTED for the updates a launch may go on with: a continued update's tables must be complete before the launch that
        // traces its rays starts (ddgi_engine.cpp: assign_vis)
        struct VisSet
        {
            uint8_t* vis[kVisLights] = {};  // one class byte per (voxel, face) and light
            uint32_t* occ = nullptr;        // light 0: the listed voxels of class kVisListed, kVisListMax per (voxel, face)
            float light[kVisLights][3] = {};
            bool valid[kVisLights] = {};
            uint32_t launch_seq = 0;        // the handle's launch_seq when a table of the set was last (re)computed: launches from this one on see it complete
            bool on_prep_stream = false;    // ... by a kernel on the handle's preparation stream that the handle's stream has not waited for yet: not to be used
        } vis_set[2 * kAqChainMax];         // [0, kAqChainMax): filled on the handle's stream; the rest: predictions made on the preparation stream
        int32_t* vis_list = nullptr;        // the (voxel, face) pairs that are classified: empty voxel, occupied on the face's other side
        int n_vis_list = -1;                // (-1: not built yet)
    } dev_scene[4];

    // memoised lattice hashes on device: one copy per device, shared by the process's handles (ddgi_engine.cpp: ensure_noise)
    bool noise_shared = false;
    NoiseLut noise{};

    // rays
    GlibcRand rand;
    bool rand_seeded = false;
    std::vector<ddgi_probe_ray> host_rays;  // full grid (what RVPT::probe_rays holds)
    void* host_rays_pinned = nullptr;       // host_rays.data() while that memory is page-locked (hipHostRegister): the per-frame upload's DMA source
    float4* d_rays = nullptr;               // local slab
    size_t d_rays_capacity = 0;             // in rays
    uint32_t n_local_rays = 0;

    // textures (REF: rgba8 texels, slab-major).  The handle owns a RING of `np` texture pairs, one allocation per texture
    // (pair k of texture i starts at own_tex[i] + k * tex_bytes[i]): np = 1 unless updates are in flight together — REF mode's
    // frames in flight (a launch continues the next update into the next pair) and/or the pipelined multi-GPU exchange (the
    // all-gather of one pair runs while the next update writes another).
    void* own_tex[2] = {nullptr, nullptr};
    int np = 1;
    int pair_cur = 0;                        // ring index of the pair in `tex` (while tex is the handle's own)
    void* tex[2] = {nullptr, nullptr};       // the textures the latest update wrote and consumers read
    bool caller_tex = false;                 // ... are the caller's (ddgi_bind_textures), not a pair of the ring
    void* tex_prev[2] = {nullptr, nullptr};  // DDGI blend: where the previous update's tiles are, when not in tex (pipelined exchange)
    size_t tex_bytes[2] = {0, 0};

    static constexpr int kMaxPairs = 16;  // 2 x the most updates one launch works on (ddgi_types.h: kAqChainMax)
    static constexpr int kRing = 64;  // timing history: one event triple per recent update
    hipEvent_t ev[kRing][3] = {};
    bool ev_has_blend[kRing] = {};    // the update recorded ev[2] (DDGI mode: after the blend); otherwise ev[1] is its end
    bool tex_ops_since_update = true;  // an API call other than ddgi_probe_update / ddgi_exchange* may have put work that touches the textures on the stream
    bool ev_valid[kRing] = {};        // the update recorded its events at all (tuning "timing")
    unsigned long long updates = 0;
    int wait_threshold = 64;
    uint32_t* d_work = nullptr;             // ray counters and status word of the wavefront trace kernels (ddgi_engine.cpp: plan_trace)
    // frames in flight (ddgi_types.h: AqChain): the queue kernel's launches are numbered; an update that continues its predecessor is published in `pub`
    uint32_t launch_seq = 0;                // sequence number of the next k_probe_trace_aq launch
    uint32_t* pub = nullptr;                // pinned host memory, kAqPubRing words (the kernel reads it through pub_dev)
    uint32_t* pub_dev = nullptr;
    hipEvent_t milestone[2] = {nullptr, nullptr};  // recorded every 16 launches: bounds how far the host runs ahead of the ring
    // the per-update records of the queue kernel (ddgi_types.h: UpdK): what may differ between the updates one launch works on
    uint32_t* upd_host = nullptr;           // pinned host memory: kAqPubRing x 2 records — [seq % kAqPubRing][own view, continued view]
    uint32_t* upd_host_dev = nullptr;       // ... as the device sees it
    uint32_t* upd_dev = nullptr;            // device: 2 x kAqCounters records, filled by the kernel's workgroups
    uint32_t chain_first_seq = 0;           // the launch that started the current chain: the earliest launch that may go on with the next update's rays
    int chain_published = 0;                // updates published as continuations since the handle last had nothing in flight
    int runahead = 0;                       // how far the host has lately been submitting updates ahead of a blocking call (0 .. kAqChainMax - 1): how many
                                            // updates ahead the light-feeler tables are predicted when a chain starts
    float last_time = 0.0f, last_dt = 0.0f;  // RenderSettings::time of the latest update and its step (DDGI mode: the lights' animation is predicted with it)
    bool have_last_time = false;
    int nrec = 1;                           // DDGI mode: ray-record buffers in d_radiance — an update writes buffer (its number % nrec), so that the blend of
                                            // update k can read its records while a launch is already tracing update k + 1 (frames in flight)
    int nrec_cap = 0;                       // > 0: a longer ring did not fit in memory (remembered until the records' size changes)
    size_t rec_stride = 0;                  // ... floats per buffer
    unsigned long long ring_k = 0;          // updates since the ring of texture pairs was made: update k writes pair k % np
    bool chain_break = true;                // something other than a probe update has touched the handle since: the next update starts a group
    unsigned long long chain_hash = 0;      // what the latest update's launch was (ddgi_engine.cpp: plan_hash)
    bool counters_dirty = false;            // a launch failed after its sequence number was handed out: re-zero before the next
    bool pin_pair = false;                  // the host holds pointers to the current pair (ddgi_device_textures): the handle stays on it
    void* d_wf_cold = nullptr;              // wavefront kernel scratch: per-slot shading state
    float4* d_wf_dir = nullptr;
    size_t wf_cold_slots = 0, wf_dir_slots = 0;
    float* d_radiance = nullptr;            // DDGI mode ray records: rgb part then (d, d*d) part (ddgi_types.h: rec_rgb_index)
    size_t d_radiance_capacity = 0;         // in floats
    int d_radiance_rays = 0;                // rays per probe the buffer was zeroed for (its padding layout)
    uint32_t frame = 0;                     // DDGI mode: updates done so far (seeds the ray rotation)
    Tuning tuning;
    bool fast_march_active = false;              // the most recent update ran the fast march
    std::map<unsigned long long, int> aq_split;  // configuration key -> measured march/event wave split of the queue kernel
    int aq_last = 0;                             // the most recently measured split (starting point of the next measurement)
    unsigned scene_epoch = 0;                    // bumped when the user scene changes (part of the configuration key)
    float* d_blend_w = nullptr;  // k_blend_weights output: [256 sums][n][256] — two of them: frame f's is buffer f & 1 (the next frame's is made beside this frame's blend)
    size_t d_blend_w_floats = 0; // ... floats per buffer
    // DDGI mode's preparation stream: what the NEXT updates need and this one does not depend on — the next frame's weight tiles (they
    // follow from its ray rotation alone) and the light-feeler tables of the light positions the coming updates are expected to have —
    // is made beside this update's blend, whose kernels leave room on every CU (the trace kernel's persistent workgroups do not)
    hipStream_t prep_stream = nullptr;
    hipStream_t prep_stream2 = nullptr;  // the feeler tables' own (tuning "prep_stream" 2)
    hipStream_t prep_tables_last = nullptr;  // the stream the latest prep_done was recorded on
    hipEvent_t prep_after = nullptr;  // handle's stream: this update's trace launch has ended (the preparation starts behind it)
    hipEvent_t prep_w_done = nullptr; // preparation stream: the next frame's weight tiles are made (the next update waits for it — its blend reads them)
    hipEvent_t prep_done = nullptr;   // preparation stream: everything given to it so far is done — the tables too (the next CHAIN's first launch waits for it)
    bool prep_w_pending = false, prep_pending = false;
    struct
    {
        bool valid = false;
        uint32_t frame = 0;
        int n = 0;
        float rot[9] = {};
    } w_made[2];                      // what each weight buffer holds
    // multi-GPU exchange of the blended textures (ddgi_exchange.cpp): one interface, two transports
    struct P2P;  // peer-to-peer transport state (ddgi_exchange.cpp)
    struct Exchange
    {
        int transport = 0;      // 0 none, 1 RCCL all-gather, 2 peer-to-peer pushes (DDGI_EXCHANGE_*)
        void* comm = nullptr;   // RCCL: ncclComm_t; caller-owned unless made by ddgi_comm_create
        P2P* p2p = nullptr;     // peer-to-peer: mapped peer buffers, flags, per-peer streams
        bool pipelined = false;                 // the exchange of a pair runs on comm_stream while later updates write other pairs of the ring
        bool broken = false;                    // a wait for another rank ran into its deadline (DDGI_ERR_TIMEOUT): refuses until attached again
        bool desync = false;                    // an update failed while attached: this rank's update count no longer matches its peers' (ddgi_exchange refuses)
        hipStream_t comm_stream = nullptr;
        hipEvent_t written = nullptr;           // handle's stream: the update's kernels have finished
        hipEvent_t sent[kMaxPairs] = {};        // comm stream: pair i's last exchange is over (RCCL) / this rank's slab has left (p2p)
        bool sent_valid[kMaxPairs] = {};
    } xch;
    float4* d_box = nullptr;                // REF mode: sample_probe per texel of the current textures (k_sample_box_filter), built on demand
    size_t box_texels = 0;
    const void* box_of = nullptr;           // ... the texture it was built from (null: stale — any update, exchange or rebind resets it)
    uint32_t* d_sample_scratch = nullptr;   // ddgi_sample_device: grouping of a batch by cage (ddgi_kernels.hip: k_sample_*)
    size_t sample_scratch_words = 0;
    unsigned long long* d_stats = nullptr;  // profiling aid, allocated on first ddgi_trace_stats(enable)
};

// shared by ddgi_engine.cpp and ddgi_exchange.cpp
GridK ddgi_make_grid(const ddgi_engine* e);
void ddgi_texture_bytes(int mode, const ddgi_irradiance_field& f, int rays_per_probe, size_t bytes[2]);
int ddgi_alloc_texture_pair(ddgi_engine* e, const size_t bytes[2], int np, void* out[2]);  // a ring of np pairs, zeroed
inline void* ddgi_pair_ptr(const ddgi_engine* e, int pair, int i) { return static_cast<uint8_t*>(e->own_tex[i]) + static_cast<size_t>(pair) * e->tex_bytes[i]; }
int ddgi_chain_len(const ddgi_engine* e);           // REF mode: updates one launch may work on = texture pairs a group takes (1: no continuation)
int ddgi_group_len(const ddgi_engine* e);           // updates one launch may work on in the handle's mode (REF: pairs of the ring; DDGI: ray-record buffers)
int ddgi_pairs_wanted(const ddgi_engine* e, bool pipelined);
void ddgi_exchange_p2p_info(const ddgi_engine* e, int* landing_textures, int* exported_mb);  // (ddgi_exchange.cpp; zeros without a peer-to-peer export)
int ddgi_resize_ring(ddgi_engine* e, int np);       // blocks; the current pair's contents move to pair 0 of the new ring
int ddgi_rebase_ring(ddgi_engine* e);               // blocks; the same ring, counted from update 0 again: the current pair's contents move to pair 0
// Host waits.  A handle on its own: hipStreamSynchronize / hipEventSynchronize.  With an exchange attached the stream may stand at a wait for
// ANOTHER RANK: polled against tuning "wait_timeout_ms"; on expiry DDGI_ERR_TIMEOUT with the lagging peer named, the exchange marked broken and
// (peer-to-peer) the handle's own waits released (ddgi_exchange.cpp).
int ddgi_sync_stream(ddgi_engine* e, hipStream_t s);
int ddgi_sync_event(ddgi_engine* e, hipEvent_t ev);
#define DDGI_TRY(expr)                  \
    do                                  \
    {                                   \
        if (int rc_ = (expr)) return rc_; \
    } while (0)
// exchange hooks called by the engine (no-ops without an initialised exchange)
int ddgi_exchange_before_update(ddgi_engine* e, int first_pair, int n_pairs);  // pipelined: the stream waits for the last exchanges of the pairs a launch may write
int ddgi_exchange_wait_latest(ddgi_engine* e);     // consumers: the handle's stream waits until the latest pair is complete
void ddgi_exchange_release(ddgi_engine* e);        // configuration changed / handle destroyed: drop pairs, stream, events
