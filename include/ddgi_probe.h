/*
 * ddgi_probe.h — C ABI of the MI355X-native DDGI probe-update engine (libddgi_probe.so).
 *
 * This is the drop-in boundary for the *probe path* of the reference renderer
 * (helenl9098/Dynamic-Diffuse-Global-Illumination-Minecraft, an RVPT fork).  The reference has no
 * plugin API: the boundary is a set of RVPT member functions plus one Vulkan descriptor-set
 * contract.  Every entry point below cites the reference interface (file:line, relative to the
 * reference root) it replaces.  Plain pointers and sizes only; no C++/torch types.
 *
 * Conventions
 *   - every function returns 0 on success or a negative ddgi_status; the message for the calling
 *     thread's last failure is ddgi_last_error().  Nothing throws across this ABI.
 *     (reference: bool returns + VK_CHECK_RESULT abort, src/rvpt/rvpt.cpp:499-592, vk_util.h:18-27)
 *   - the handle owns all device memory it allocates; caller owns every host array it passes;
 *     no caller pointer is retained past the call except by the explicit *_bind_* functions.
 *   - one handle = one GPU + one HIP stream.  Calls on one handle are not re-entrant.
 *     ddgi_probe_update is asynchronous on the handle's stream; ddgi_read_textures/ddgi_sample
 *     synchronise.  (reference: single-threaded frame loop, src/rvpt/main.cpp:80-96)
 *   - there is NO CPU fallback: every compute entry point fails with DDGI_ERR_NO_DEVICE when no
 *     gfx950 device is usable.
 */
#ifndef DDGI_PROBE_H
#define DDGI_PROBE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DDGI_ABI_VERSION 7 /* 7: the peer-to-peer transport publishes LANDING ZONES instead of a ring of 2 GiB or more (tuning "p2p_landing_zones", "p2p_exported_mb" tell), its flag
                              words live in fine-grained device memory; ddgi_upload_probe_rays leaves chunks equal to the handle's host copy alone (a buffer handed over every frame);
                              6: DDGI_ERR_TIMEOUT + tuning "wait_timeout_ms": every host wait of a handle with an exchange attached has a deadline;
                              5: ddgi_exchange_ranks; frames in flight in DDGI mode (per-update records, a ring of ray-record buffers); attaching an exchange
                              rebases the ring of texture pairs (every rank starts at pair 0, update 0); d_cage of ddgi_sample_device: 16-byte aligned or slower;
                              4: tuning "frames_in_flight" (default 8): the handle owns a ring of texture pairs, ddgi_device_textures pins the current one;
                              3: ddgi_exchange_p2p_*, ddgi_exchange_transport, ddgi_scene_skip_field; tuning "fast_march", "sample_box"; "autotune" off by default */

/* ---- wire formats: byte-identical to the reference's UBO/SSBO records ------------------------ */

/* RVPT::IrradianceField, src/rvpt/rvpt.h:82-90 == GLSL block probe_pass.comp:33-40 (std140), 48 B */
typedef struct ddgi_irradiance_field
{
    int32_t probe_count[3];      /* @0  probes along x, y, z (default 9,7,9)                    */
    int32_t side_length;         /* @12 integer probe spacing (rvpt.h:85)                       */
    float hysteresis;            /* @16 blend coefficient (dormant in the reference)            */
    int32_t sqrt_rays_per_probe; /* @20 s; rays per probe = s*s (ddgi_set_ray_tile: tile_x*tile_y) */
    int32_t _pad0[2];            /* @24                                                         */
    float field_origin[3];       /* @32 world position of the field centre (default 1.4,0,1)    */
    uint8_t visualize;           /* @44 host-only flag                                          */
    uint8_t _pad1[3];
} ddgi_irradiance_field;

/* RVPT::RenderSettings, src/rvpt/rvpt.h:70-80 == probe_pass.comp:15-26, 32 B */
typedef struct ddgi_render_settings
{
    int32_t screen_width;     /* unused by the probe path */
    int32_t screen_height;    /* unused by the probe path */
    int32_t max_bounces;      /* default 8 */
    int32_t camera_mode;      /* unused */
    int32_t render_mode;      /* unused by the probe update; ddgi_render: the integrator */
    int32_t scene;            /* 0 cave, 1 Cornell box, 2 house */
    float time;               /* +2 per frame (rvpt.cpp:281); only animated lights read it */
    int32_t visualize_probes; /* unused by the probe update; ddgi_render: probes drawn as spheres */
} ddgi_render_settings;

/* struct ProbeRay, src/rvpt/probe.h:5-19 == structs.glsl:22-27 (std430), 48 B */
typedef struct ddgi_probe_ray
{
    float origin[3];
    float _pad0;
    float direction[3]; /* unit length */
    float _pad1;
    float probe_info[3]; /* (probe_index, tile_x, tile_y) stored as floats */
    float _pad2;
} ddgi_probe_ray;

/* struct Light, assets/shaders/structs.glsl:54-59 (a shader constant in the reference) */
typedef struct ddgi_light
{
    float intensity;
    float col[3];
    float pos[3];
} ddgi_light;

#define DDGI_MAX_LIGHTS 8

typedef enum ddgi_status
{
    DDGI_OK = 0,
    DDGI_ERR_INVALID_ARGUMENT = -1,
    DDGI_ERR_NO_DEVICE = -2,   /* no usable gfx950 GPU: the product has no CPU path */
    DDGI_ERR_HIP = -3,         /* a HIP runtime call failed; see ddgi_last_error() */
    DDGI_ERR_OUT_OF_MEMORY = -4,
    DDGI_ERR_NOT_READY = -5,   /* e.g. probe_update before any rays were generated/uploaded */
    DDGI_ERR_UNSUPPORTED = -6,
    DDGI_ERR_TIMEOUT = -7      /* a wait that depends on ANOTHER RANK of a multi-GPU exchange did not end within tuning "wait_timeout_ms"
                                  (≙ the reference's fence timeout, vk_util.cpp:65,94-97: DEFAULT_FENCE_TIMEOUT); ddgi_last_error() names
                                  the peer, the flag and the exchange numbers expected and seen */
} ddgi_status;

/* Pipeline selection.
 *   REF  — exactly the reference's LIVE behaviour: host-supplied stratified rays, one texel per
 *          ray, rgba8 textures, distance texture all zero, 5x5 box-filter sampling.
 *   DDGI — the reference's dormant pieces switched on: in-kernel spherical-Fibonacci rays,
 *          octahedral irradiance + depth-moment tiles, hysteresis blend, Chebyshev visibility. */
typedef enum ddgi_mode
{
    DDGI_MODE_REF = 0,
    DDGI_MODE_DDGI = 1
} ddgi_mode;

typedef struct ddgi_engine* ddgi_handle;

/* ---- lifetime ------------------------------------------------------------------------------- */

/* Replaces RVPT::RVPT + initialize() for the probe path (src/rvpt/rvpt.cpp:212-264: context_init,
 * create_rendering_resources 757-924 [probe pipeline + the two probe images 873-890],
 * add_per_frame_data 926-1030 [probe SSBO 949-952]).  `device` is the HIP device ordinal. */
int ddgi_create(const ddgi_irradiance_field* field, const ddgi_render_settings* settings,
                int device, ddgi_handle* out);

/* Same, for one rank of a z-slab sharded grid (new — the reference is single-GPU).
 * The rank owns probes with z in [rank*cz/world, (rank+1)*cz/world); cz % world must be 0.
 * Textures are still allocated for the whole grid so an all-gather can fill in the other slabs. */
int ddgi_create_sharded(const ddgi_irradiance_field* field, const ddgi_render_settings* settings,
                        int device, int rank, int world, ddgi_handle* out);

/* Replaces RVPT::shutdown (rvpt.cpp:433-468). */
int ddgi_destroy(ddgi_handle h);

/* Replaces RVPT::recreate_probe_textures (rvpt.cpp:661-755): new counts / rays / spacing /
 * origin; textures are re-created zeroed, probe rays must be regenerated or re-uploaded. */
int ddgi_configure(ddgi_handle h, const ddgi_irradiance_field* field,
                   const ddgi_render_settings* settings);

/* ddgi_configure with carry-over (no reference counterpart — the reference
 * drops its textures, rvpt.cpp:661-755): with carry_over != 0 every new probe that stands exactly
 * where an old probe stood (same world position; REF mode additionally: same rays per probe) keeps
 * that probe's tiles, all others start zeroed, and in DDGI mode the frame sequence continues.
 * The handle uses its own textures afterwards (re-bind external ones with ddgi_bind_textures).
 * carry_over == 0 is ddgi_configure. */
int ddgi_reconfigure(ddgi_handle h, const ddgi_irradiance_field* field,
                     const ddgi_render_settings* settings, int carry_over);

int ddgi_set_mode(ddgi_handle h, int mode /* ddgi_mode */);

/* Non-square ray counts.  The reference only knows rays per probe = sqrt_rays_per_probe^2 (rvpt.h:87, UI
 * rvpt.cpp:342; generate_samples' s x s strata, rvpt.cpp:1147-1173); BASELINE's 8-GPU configuration asks for
 * 512.  The 48-byte field record keeps its layout; this setter makes the ray tile tile_x x tile_y:
 *   REF  tile_x strata along z (= texel columns of a probe's tile), tile_y strata along phi (= rows):
 *        u = (x + jitter)/tile_x, v = (y + jitter)/tile_y, ray i = y*tile_x + x; textures become
 *        W = cx*cz*tile_x by H = cy*tile_y; sample_probe inverts with tile_x / tile_y.  With
 *        tile_x == tile_y == sqrt_rays_per_probe every formula is the reference's.
 *   DDGI rays per probe n = tile_x*tile_y (the spherical Fibonacci set is defined for every n).
 * (0, 0) restores the field's square tile; ddgi_configure / ddgi_reconfigure also do.  Textures restart
 * zeroed and probe rays must be regenerated or re-uploaded. */
int ddgi_set_ray_tile(ddgi_handle h, int tile_x, int tile_y);
int ddgi_get_ray_tile(ddgi_handle h, int* tile_x, int* tile_y);
/* Reference raster size of the handle's REF-mode textures for its current ray tile
 * (ddgi_texture_size for the square tile). */
int ddgi_get_texture_size(ddgi_handle h, int* width, int* height);

/* Overrides the light table of one scene (the reference compiles them into the shader,
 * structs.glsl:61-89; defaults here are the shipped tables).  n <= DDGI_MAX_LIGHTS. */
int ddgi_set_lights(ddgi_handle h, int scene, const ddgi_light* lights, int n);

/* ---- probe rays (REF mode) -------------------------------------------------------------------- */

/* Replaces RVPT::generate_probe_rays + generate_samples (rvpt.cpp:1147-1224) followed by
 * probe_buffer.copy_to(probe_rays) (rvpt.cpp:285).  The reference draws its jitter from the
 * unseeded C rand(); here the glibc TYPE_3 generator is restated, `seed` seeds it on the first
 * call (seed 1 == an unseeded glibc process) and later calls continue the sequence exactly as
 * repeated calls do in the reference.  Pass reseed != 0 to restart it. */
int ddgi_generate_probe_rays(ddgi_handle h, uint32_t seed, int reseed);

/* Replaces probe_buffer.copy_to(probe_rays) (rvpt.cpp:285, vk_util.h:508-518) for caller-made
 * rays.  n must equal the (local slab's) probe count * s*s for the current configuration, in
 * the reference's order: probe-major (p = py*cx*cz + pz*cx + px), ray i = y*s + x.
 * For a sharded handle pass the FULL-grid array; the handle keeps only its slab.
 * Meant to be called every frame, as the reference does — with the same rays each time (it generates them at start-up and
 * after a reconfiguration only: main.cpp:47, rvpt.cpp:721-726): per chunk of 64 Ki rays, on host threads, a chunk whose bytes
 * equal the handle's host copy is left alone; any other is checked (probe_info inside the grid: the reference's shader trusts
 * it blindly — a bad ray rejects the call and leaves the previous rays in place), copied into the handle's page-locked host
 * copy and sent.  An unchanged buffer touches neither the GPU nor the frames in flight.  Synchronous: `rays` may be reused
 * on return. */
int ddgi_upload_probe_rays(ddgi_handle h, const ddgi_probe_ray* rays, size_t n);

/* Copies the handle's current host-side ray array (what generate produced) into `rays`
 * (capacity n records, full grid).  New: lets a caller inspect what the reference keeps in
 * RVPT::probe_rays (rvpt.h:113). */
int ddgi_get_probe_rays(ddgi_handle h, ddgi_probe_ray* rays, size_t n);

/* ---- the hot path ------------------------------------------------------------------------------ */

/* Replaces, per frame: RVPT::update()'s uploads (rvpt.cpp:281-287) + the probe half of
 * record_compute_command_buffer (barrier + vkCmdDispatch, rvpt.cpp:1105-1129) + Queue::submit
 * (rvpt.cpp:378-380).  Asynchronous on the handle's stream.  `settings` may be NULL to reuse the
 * last one; the function does NOT add +2 to time (the caller's RVPT::update does, rvpt.cpp:281).
 * Never blocks the host.  How the trace kernel splits its waves between marching and shading depends on the
 * configuration (grid, rays, scene, bounces, lights, mode): a host calls ddgi_tune() once at load time to
 * measure it (worth 5-10 %), or sets "autotune" to let the first update of a configuration measure it. */
int ddgi_probe_update(ddgi_handle h, const ddgi_render_settings* settings);

/* Measures, now and blocking, the trace kernel's march/event wave split for the handle's CURRENT configuration
 * and remembers it (up to 64 configurations per handle): a few extra launches of the same trace (idempotent) and
 * host synchronisations, some tens of milliseconds.  Call it after the rays are in place (REF mode). */
int ddgi_tune(ddgi_handle h);

/* Tuning switches of a handle, by name.  The environment variables in brackets are read ONCE, when the handle
 * is created, as initial values — never on the per-frame path.
 *   "autotune"      1 = the first update of a configuration measures the wave split itself (that one update blocks the
 *                   host, as ddgi_tune does); 0 (default) = ddgi_probe_update never blocks                    [DDGI_AUTOTUNE]
 *   "fast_march"    1 = TOLERANCE MODE: marches skip empty space through the scene's 2-bit skip field (csrc/ddgi_device.h:
 *                   fast_march_step).  Not bit-exact with the exact march — REF texels: |d| <= 1/255 on >= 99.9 % of the
 *                   channels, mean < 0.05/255 (measured on C3: 99.996 %, 0.0013/255); cage indices and the sampler are
 *                   untouched (tests/test_gpu_fast_march.py).  A request: where the skip field does not fit in LDS the update
 *                   runs the exact march; ddgi_get_tuning "fast_march_active" tells.  0 (default) = the exact march [DDGI_FAST_MARCH]
 *   "march_waves"   n > 0 pins the split (waves that march, of 16); 0 = per configuration            [DDGI_AQ_MARCH]
 *   "trace_kernel"  0 auto, 1 round-based, 2 ray per lane, 3 queues (cross-checks)   [DDGI_TRACE_KERNEL=rounds|lane|queues]
 *   "frames_in_flight"  REF mode.  The reference's host contract is MAX_FRAMES_IN_FLIGHT = 2 with a fence per frame (src/rvpt/rvpt.h:23,
 *                   rvpt.cpp:277-278): frame k + 1 is submitted while frame k runs.  n (default 8 = the most, 1 = off) is an UPPER BOUND on
 *                   the updates one launch works on: the handle keeps n texture pairs (2 n under the pipelined exchange), updates come
 *                   in aligned groups of n, and an update submitted as the SAME WORK as its predecessor (same configuration, rays, lights,
 *                   tuning — nothing but probe updates, exchanges, samples, renders and reads on the handle in between) while the
 *                   predecessor's launch still runs is continued by that launch's resident workgroups instead of waiting for their
 *                   drain (one ray's 8 bounces are a dependent chain: 0.25 ms of thinning occupancy per launch).  How many updates a
 *                   launch really continues is the host's doing: one that waits for frame k's fence before submitting frame k + 2,
 *                   like the reference, gets two per launch; a loop that never waits gets n.  Results are unchanged bit for bit; every
 *                   update still has its own launch in stream order, so everything enqueued behind an update sees it complete — but an
 *                   update's end event can fire up to n - 1 updates late (its launch went on with its successors' rays): a host that
 *                   consumes every update at once loses nothing and gains nothing.  Rings of more than 2 GiB (C4-sized grids and beyond, where
 *                   the drain is a fraction of a percent of an update) are halved, down to two pairs.  Changing it blocks and re-makes the ring (the
 *                   textures carry over); set it before the exchange is attached.                              [DDGI_FRAMES_IN_FLIGHT]
 *   "reserve_cus"   n > 0: the queue kernel launches n workgroups fewer than the device has CUs.  Its persistent workgroups take a
 *                   CU's registers and LDS whole, so a KERNEL of a multi-GPU exchange (an RCCL collective, a shader copy) otherwise
 *                   finds no CU until a launch ends; transfers by the copy engines need none.  0 (default)     [DDGI_RESERVE_CUS]
 *   "timing"        1 (default): every update records two (REF) or three (DDGI) events on its stream for ddgi_last_update_ms /
 *                   ddgi_update_history_ms; 0: none — the queries then fail with DDGI_ERR_NOT_READY, and a stream of
 *                   back-to-back updates loses ~6 us per update less to the command processor                [DDGI_TIMING]
 *   "blend_merge"   DDGI blend: depth and irradiance tiles in ONE launch up to this many half probe groups (8 probes) per CU
 *                   (default 1: grids / slabs of up to 8 probes per CU), in two launches above                  [DDGI_BLEND_MERGE]
 *   "blend_kernel"  0 auto, 1 one probe per workgroup (cross-check), 2 auto but every quotient by the compiler's
 *                   division — the path a probe group takes whose sums lie outside the short division's domain
 *                   (cross-check)                                                   [DDGI_BLEND_KERNEL=probe|division]
 *   "wait_timeout_ms"  the deadline of every host wait of a handle that has a multi-GPU exchange attached (ddgi_synchronize above);
 *                   default 10 000, 0 = none                                                        [DDGI_WAIT_TIMEOUT_MS]
 *   "verbose", "noise_lut", "aq_pool", "wf_pool", ... (profiling; see ddgi_engine.cpp: kTuningKeys)
 * ddgi_get_tuning also answers "march_waves_measured": the split most recently measured (0 = none yet), so a
 * host can persist it and pin it next time. */
int ddgi_set_tuning(ddgi_handle h, const char* name, int value);
int ddgi_get_tuning(ddgi_handle h, const char* name, int* value);

/* Waits for the stream (≙ Fence::wait, vk_util.cpp:94-97).  The reference's fences give up after DEFAULT_FENCE_TIMEOUT = 1 s
 * (vk_util.cpp:65).  Here: a handle on its own cannot wait for anybody but its own kernels (whose queue waits have their own safety
 * net) and waits without a limit; a handle with a multi-GPU exchange attached waits for OTHER RANKS — a peer that died, was never
 * started or stopped taking part must not turn this call into a silent forever: every host wait of such a handle (this call, the
 * consumers, the reads, the timing queries, detaching, destroying) polls with the deadline of tuning "wait_timeout_ms" (default
 * 10 000; 0 = no limit) and on expiry returns DDGI_ERR_TIMEOUT — ddgi_last_error() names the peer rank that is behind, the flag
 * (`ready`: it has not released the pair for exchange n; `arrived`: its slab of exchange n has not landed), the exchange number
 * expected and the one seen.  The exchange is then BROKEN: ddgi_exchange and the consumers refuse until it is attached again on every
 * rank; with the peer-to-peer transport the library releases its own streams' waits itself (it writes the flag words they stand at),
 * so the handle drains and can be detached, reconfigured or destroyed — the textures hold whatever had arrived.                         */
int ddgi_synchronize(ddgi_handle h);

/* Device time of the kernels of the most recent completed ddgi_probe_update, measured with HIP
 * events recorded on the handle's stream around each launch.  Any pointer may be NULL.
 * Synchronises.  blend_ms is 0 in REF mode. */
int ddgi_last_update_ms(ddgi_handle h, float* trace_ms, float* blend_ms, float* total_ms);

/* Per-kernel device times (HIP events on the handle's stream) of the most recent updates, oldest
 * first: up to `capacity` entries, at most the last 64 updates.  Synchronises.  Lets a caller time
 * the kernels of a whole timed region without synchronising inside it. */
int ddgi_update_history_ms(ddgi_handle h, float* trace_ms, float* blend_ms, int capacity, int* n_out);

/* Profiling aid (no reference counterpart): enables/disables and reads the trace kernel's
 * utilisation counters accumulated since the last call: out64 = {march-loop trips summed over
 * waves, lane-steps, event groups, lane-events, waves, rounds, task fetches, 0, shader-clock
 * cycles per phase (scan, march, march barrier, list, events, events barrier), 0, 0,
 * [16..22] cycles per event bucket, 0, [24..30] event groups per bucket, 0, reserved ...}.
 * Synchronises. */
int ddgi_trace_stats(ddgi_handle h, int enable, unsigned long long* out64);

/* ---- outputs ----------------------------------------------------------------------------------- */

/* New (the reference has no readback path): copies both probe textures to the
 * host in the reference's raster layout — R8G8B8A8_UNORM, W = cx*cz*s, H = cy*s, probe p's tile
 * at ((p mod cx*cz)*s, (p div cx*cz)*s)  (rvpt.cpp:873-890, probe_pass.comp:139-145).
 * REF mode only.  Each array holds 4*W*H bytes.  Either pointer may be NULL.  Synchronises. */
int ddgi_read_textures(ddgi_handle h, uint8_t* albedo_rgba8, uint8_t* distance_rgba8);

/* DDGI mode: copies the float tiles to the host, probe-major in reference probe order p:
 * irradiance [P][8][8][4] f32 (6x6 interior + 1 texel border, rgb + unused a) and
 * depth moments [P][16][16][2] f32 (14x14 interior + border; mean distance, mean squared).
 * Either pointer may be NULL.  Synchronises. */
int ddgi_read_tiles(ddgi_handle h, float* irradiance, float* depth);

/* DDGI mode: the frame index that seeds the next update's ray-set rotation and per-ray RNG
 * (starts at 0 after create/configure/set_mode and advances by one per ddgi_probe_update). */
int ddgi_set_frame(ddgi_handle h, uint32_t frame);

/* Replaces get_diffuse_gi (assets/shaders/intersection.glsl:1306-1409) as called from
 * integrator_DDGI / integrator_indirect (integrators.glsl:67,205) for a batch of n shading
 * points: pos_xyz/nrm_xyz are n*3 floats (Isect.pos / Isect.normal), rgb_out n*3 floats,
 * cage_idx8_out n*8 int32 = probe_index_1d of the 8 cage corners in the shader's order
 * (offset = (i>>2, i>>1, i)&1), or -1 for every corner when the shader returns magenta early.
 * Host pointers; synchronises.  cage_idx8_out may be NULL. */
int ddgi_sample(ddgi_handle h, const float* pos_xyz, const float* nrm_xyz, size_t n,
                float* rgb_out, int32_t* cage_idx8_out);

/* ---- the primary-visibility consumer ------------------------------------ */

/* The Camera UBO of the render pass (assets/shaders/compute_pass.comp:30-35; filled by
 * Camera::get_data, src/rvpt/camera.cpp:100-110): column-major mat4 + (aspect, hfov in radians,
 * ortho scale, 0).  80 bytes. */
typedef struct ddgi_camera
{
    float matrix[16];
    float params[4];
} ddgi_camera;

/* Replaces the render half of record_compute_command_buffer (rvpt.cpp:1131-1140) for the
 * integrators that consume the probe field: compute_pass.comp:main 162-191 = camera ray
 * (camera.glsl:29-74; settings->camera_mode 0 pinhole, 1 ortho) + eval_integrator
 * (compute_pass.comp:58-87; settings->render_mode 0 DDGI, 1 direct, 2 indirect, 3 colour, 4 normal,
 * 5 depth) over a screen_width x screen_height image, using the handle's current probe textures in
 * its current mode.  settings->visualize_probes != 0 draws the probes as spheres in modes 0 and 2
 * (intersect_probes, intersection.glsl:1102-1128; integrators.glsl:45-65, 180-199).  Two debug views
 * (SURVEY.md 8(f) row 2): render_mode 6 = the whole REF probe texture stretched over the screen (the
 * reference's dormant get_probe_image_coords blit, compute_pass.comp:116-124, 185-190), render_mode 7 =
 * the first probe index of every pixel's cage as a colour, magenta outside the field (README.md:89-91).
 * Output: rgba8 (the reference's result_image format), row 0 = top; optional unquantised rgb
 * (3 floats per pixel).  Synchronises. */
int ddgi_render(ddgi_handle h, const ddgi_camera* camera, const ddgi_render_settings* settings,
                uint8_t* rgba8_out, float* rgb_f32_out);
/* Same on device pointers, asynchronous on the handle's stream. */
int ddgi_render_device(ddgi_handle h, const ddgi_camera* camera, const ddgi_render_settings* settings,
                       uint32_t* d_rgba8_out, float* d_rgb_f32_out);

/* ---- device-pointer level (for hosts that own device memory / streams, e.g. PyTorch) ---------- */

/* Uses `hip_stream` (a hipStream_t) for all subsequent work of this handle; NULL = default. */
int ddgi_set_stream(ddgi_handle h, void* hip_stream);

/* Device addresses and byte sizes of the full-grid, slab-major buffers
 *   REF : tex0 = albedo   [cz][cy][cx][s][s] rgba8,  tex1 = distance, same shape
 *   DDGI: tex0 = irradiance [cz][cy][cx][8][8] 4xf32, tex1 = depth [cz][cy][cx][16][16] 2xf32
 * and of this rank's contiguous slab inside each (offset/bytes), which is what an all-gather
 * exchanges.  The handle owns a ring of texture pairs (tuning "frames_in_flight") and moves on to the next
 * one with every update; a host that asks for the addresses may keep them: from this call on the handle STAYS on the pair it
 * returns (no frames in flight; the sampler's per-texel table is no longer cached, since the host may write through the
 * pointers) — except under the pipelined exchange, which alternates pairs by contract (call this after every update there).
 * The first such call on a handle with a ring of several pairs blocks once: the ring shrinks to the one pair a pinned handle uses
 * (the textures carry over; the addresses returned are the new ones) instead of holding seven more pairs for nothing. */
int ddgi_device_textures(ddgi_handle h, void** tex0, size_t* tex0_bytes, void** tex1,
                         size_t* tex1_bytes, size_t* slab_offset0, size_t* slab_bytes0,
                         size_t* slab_offset1, size_t* slab_bytes1);

/* Makes the handle use caller-allocated full-grid device buffers (same layout/size as above)
 * instead of its own, so that e.g. torch.distributed.all_gather_into_tensor can run in place.
 * The caller keeps ownership and must keep them alive while bound.  NULLs rebind the internal
 * buffers.
 * DDGI mode: the update mixes every texel with the old value at its own place, and a tile's border
 * texels are copies of interior texels (octahedral wrap).  Buffers bound here must therefore hold
 * either zeros (fresh tiles) or tiles this library has written (whose borders equal their sources);
 * tiles with inconsistent borders would keep the inconsistency, decaying with the hysteresis. */
int ddgi_bind_textures(ddgi_handle h, void* tex0, void* tex1);

/* ddgi_sample on device pointers, asynchronous on the handle's stream.  d_cage_idx8_out (optional): 8 int32 per point; a
 * 16-byte aligned buffer (anything hipMalloc returns) is written with two 16-byte stores per point, any other with eight
 * 4-byte ones. */
int ddgi_sample_device(ddgi_handle h, const float* d_pos_xyz, const float* d_nrm_xyz, size_t n,
                       float* d_rgb_out, int32_t* d_cage_idx8_out);

/* ---- multi-GPU: the exchange of the blended textures ---------------------------------
 * New design — the reference is single-GPU (one vkQueueSubmit, rvpt.cpp:372-380) and has no collective.
 * Handles made by ddgi_create_sharded(rank, world) trace + blend their z-slab; ONE in-place all-gather
 * per texture (RCCL over xGMI; the slab-major layout makes a rank's contribution one contiguous chunk)
 * then gives every rank the whole field before the cage sample.  Per frame, on every rank:
 *       ddgi_probe_update(h, settings);  ddgi_exchange(h);          ... ddgi_sample_device / ddgi_render_device
 * RCCL is loaded at run time (the librccl.so.1 already in the process, else the system's); a host that never
 * shards needs none.  One process per GPU, or one process driving several handles (then bracket the
 * ddgi_exchange calls of one frame with ddgi_exchange_group_begin/end, ≙ ncclGroupStart/End). */

/* Attaches a communicator (an ncclComm_t whose size/rank equal the handle's world/rank; the caller keeps
 * ownership) and sets the exchange up; nccl_comm == NULL detaches.
 *   pipelined == 0  the all-gather runs on the handle's stream, in order after the update.
 *   pipelined != 0  the handle's ring of texture pairs is doubled (two pairs; 2 x "frames_in_flight" in REF mode) and the
 *                   all-gather of update k runs on a communication stream while later updates already trace into
 *                   other pairs (the DDGI blend reads its own slab's previous tiles from the pair it wrote last).
 *                   Every consumer call on the handle (sample / render / read) waits for the latest exchange.
 *                   Blocks (the ring is re-made; the textures so far carry over).
 * ddgi_configure / ddgi_reconfigure / ddgi_set_mode / ddgi_set_ray_tile detach: call this again after them. */
int ddgi_exchange_init(ddgi_handle h, void* nccl_comm, int pipelined);
/* Issues the all-gather of the most recent ddgi_probe_update (asynchronous). */
int ddgi_exchange(ddgi_handle h);
/* Makes the handle's stream wait for every exchange issued so far (needed only before work the caller
 * enqueues itself on that stream, e.g. a timing fence; the handle's own consumers wait by themselves). */
int ddgi_exchange_finish(ddgi_handle h);
/* One process driving several handles brackets the ddgi_exchange calls of a frame (≙ ncclGroupStart/End; per
 * thread).  Inside the bracket RCCL only records the collectives: consumers of the exchanged textures and the next
 * ddgi_probe_update come after ddgi_exchange_group_end. */
int ddgi_exchange_group_begin(void);
int ddgi_exchange_group_end(void);

/* The same exchange without RCCL: every rank pushes its slab straight into every other rank's textures (SURVEY.md
 * §8e's one-shot alternative to a ring; one copy stream per peer, rendezvous through flag words in device memory
 * that the command processor waits on).  One process per rank (the peers' textures are mapped with
 * hipIpcOpenMemHandle); several ranks may share ONE device — which RCCL refuses.  Set-up, on every rank:
 *     ddgi_exchange_p2p_export(h, pipelined, mine);          publish this handle's buffers (512 opaque bytes)
 *     ... the host gathers every rank's 512 bytes, in rank order, by whatever channel it has ...
 *     ddgi_exchange_p2p_init(h, all, world);                 map the peers' buffers
 * then ddgi_probe_update / ddgi_exchange / consumers exactly as with RCCL (`pipelined` means the same).  The ranks
 * must all still be alive when any of them detaches (ddgi_exchange_init(h, NULL, 0), reconfiguration, destroy):
 * put a host barrier before tearing down, as before ncclCommDestroy.
 * What a peer maps is the handle's ring of texture pairs — its push lands where this rank's consumers read —, unless a
 * ring reaches 2 GiB (a buffer of 2^31 bytes or more is not handed to another process reliably on the ROCm stack
 * measured): then the peers get LANDING ZONES — per texture and parity of the exchange's number a buffer of world - 1
 * slabs — and a stream of this rank's own copies what has landed into the pair (DDGI_P2P_LANDING=1 in the environment
 * asks for zones on grids of any size).  ddgi_get_tuning "p2p_landing_zones" (textures with zones, 0: none) and
 * "p2p_exported_mb" (MB of this rank a peer maps) tell which; every rank must come to the same answer (same grid, mode,
 * frames in flight, environment), ddgi_exchange_p2p_init checks it.  The flag words live in fine-grained device memory
 * where the runtime offers it (another GPU writes them, this GPU's command processor polls them). */
#define DDGI_P2P_ADDRESS_BYTES 512
int ddgi_exchange_p2p_export(ddgi_handle h, int pipelined, uint8_t address[DDGI_P2P_ADDRESS_BYTES]);
int ddgi_exchange_p2p_init(ddgi_handle h, const uint8_t* addresses_rank_major, int world);

#define DDGI_EXCHANGE_NONE 0
#define DDGI_EXCHANGE_RCCL 1
#define DDGI_EXCHANGE_P2P 2
/* Which transport the handle's exchange uses (DDGI_EXCHANGE_*), and whether it is pipelined. */
int ddgi_exchange_transport(ddgi_handle h, int* transport, int* pipelined);
/* How many ranks the attached transport really spans: RCCL — ncclCommCount of the communicator; peer-to-peer — the peers whose
 * buffers are mapped, plus this rank; 0 without an exchange.  (A benchmark line's proof that the transport saw N ranks.) */
int ddgi_exchange_ranks(ddgi_handle h, int* ranks);

/* Communicator bootstrap through the same RCCL instance (thin wrappers of ncclGetUniqueId /
 * ncclCommInitRank / ncclCommInitAll / ncclCommDestroy), for hosts that do not link RCCL themselves:
 * rank 0 makes the 128-byte id and hands it to the other ranks by whatever channel the host has. */
int ddgi_comm_unique_id(uint8_t id128[128]);
int ddgi_comm_create(const uint8_t id128[128], int world, int rank, int device, void** nccl_comm_out);
int ddgi_comm_create_all(int ndev, const int* devices, void** nccl_comms_out);
int ddgi_comm_destroy(void* nccl_comm);

/* ---- introspection ------------------------------------------------------------------------------ */

int ddgi_abi_version(void);
const char* ddgi_last_error(void);

/* Geometry helpers shared by every caller (pure host arithmetic, usable without a GPU):
 * reference raster size (rvpt.cpp:873-874) and the tile origin of probe p
 * (probe_pass.comp:139-145). */
int ddgi_texture_size(const ddgi_irradiance_field* field, int* width, int* height);
int ddgi_probe_tile_origin(const ddgi_irradiance_field* field, int probe_index, int* x, int* y);

/* Host-only form of ddgi_generate_probe_rays (no handle, no GPU): seeds a fresh generator with
 * `seed`, discards `skip_calls` whole generate_probe_rays() calls' worth of draws for this
 * configuration (so call k of a process can be reproduced), and writes the full-grid array.
 * n must be probe_count.x*y*z * s*s. */
int ddgi_generate_probe_rays_host(const ddgi_irradiance_field* field, uint32_t seed, int skip_calls,
                                  ddgi_probe_ray* rays, size_t n);

/* Same for a tile_x x tile_y ray tile (ddgi_set_ray_tile); n = probes * tile_x * tile_y. */
int ddgi_generate_probe_rays_host_tile(const ddgi_irradiance_field* field, int tile_x, int tile_y, uint32_t seed,
                                       int skip_calls, ddgi_probe_ray* rays, size_t n);

/* ---- baked scenes on disk, user scenes ------------------------------------ */

#define DDGI_SCENE_USER 3 /* RenderSettings::scene value that selects the loaded user scene */

/* Writes the bake of a built-in scene (0 cave, 1 Cornell, 2 house: getBlockAt evaluated over the
 * scene's box, intersection.glsl:699-826) to a versioned file: "DDGIVOX1", version, source scene,
 * lo[3], dim[3], noise id, then dim.x*dim.y*dim.z block-type bytes (x fastest).  Host only. */
int ddgi_scene_save(int scene, const char* path);

/* Loads such a file, or takes a caller-made grid, as the handle's user scene (scene id 3).  Block
 * types 0..13 keep the reference's meaning (0 empty; albedo per getColorAt, intersection.glsl:872-
 * 1047); outside the box the world is the axis-wise extrusion of the box's outermost layer.  The
 * scene's lights are set with ddgi_set_lights(h, 3, ...) (default: none). */
int ddgi_scene_load(ddgi_handle h, const char* path);
int ddgi_scene_set_grid(ddgi_handle h, const int32_t lo[3], const int32_t dim[3], const uint8_t* block_types);

/* Host evaluation of the baked scene the kernels traverse (block type 0..13 at integer voxel
 * id; replaces getBlockAt, intersection.glsl:699-826, which the reference evaluates per march
 * step on the GPU).  Usable without a GPU; exists so the bake can be checked against the
 * oracle. */
int ddgi_scene_block_at(int scene, int x, int y, int z);

/* The fast march's skip field of a built-in scene (tuning "fast_march"; csrc/ddgi_device.h: fast_march_step), unpacked to one byte
 * per voxel of the bake box (x fastest): 0 = occupied, else 1 + min(r, 2) with r the voxel's free Chebyshev radius — every voxel
 * within r of it, in the world the kernels see (outside the box the border layer repeats), is empty.  lo / dim receive the box;
 * codes == NULL only queries them.  Usable without a GPU; exists so the field's guarantee can be checked by brute force. */
int ddgi_scene_skip_field(int scene, int32_t lo[3], int32_t dim[3], uint8_t* codes, size_t capacity);

/* The pinned elementary functions the engine uses on host and device ("Arithmetic
 * pinning"); exported so tests can compare them with libm and with the oracle's restatement. */
float ddgi_pinned_sinf(float x);
float ddgi_pinned_cosf(float x);
float ddgi_pinned_acosf(float x);
/* P6b: sine and cosine of a small angle in binary32 (the hemisphere sample, probe_pass.comp:153,176) */
int ddgi_pinned_sincos_small(float x, float* sin_out, float* cos_out);

#ifdef __cplusplus
}
#endif
#endif /* DDGI_PROBE_H */
