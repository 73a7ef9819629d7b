// ddgi_trace_wf.hip — the probe update (assets/shaders/probe_pass.comp:main and everything it calls)
// scheduled as a wavefront path tracer inside ONE persistent 1024-lane workgroup per CU:
//   k_probe_trace_aq  slots travel through LDS queues; m waves march, 16 - m waves shade (m per configuration, ddgi_tune);
//                     no workgroup barrier after start-up                                                  (default)
//   k_probe_trace_wf  the same work in synchronous rounds (sort / events / march) with barriers   (cross-check, counters)
// Same per-ray arithmetic, in the same order, as the ray-per-lane k_probe_trace_ref (ddgi_kernels.hip),
// hence bit-identical results; only the schedule differs.  (The opt-in fast march — Cfg::kFast, ddgi_device.h:
// fast_march_step — is the one exception: a tolerance mode, see tests/test_gpu_fast_march.py.)
//
// Why a different schedule: per ray the path alternates ~13 voxel marches (1..125 dependent steps
// each, heavy tailed) with hit shading whose cost depends on the block type hit.  With one ray
// bound to one lane, a wave spends ~3/4 of its march-loop issue slots on parked lanes and every
// shading round pays for every block type present in the wave.
//
// Here rays are NOT bound to lanes.  A pool of P rays lives in LDS as a structure of arrays, 18 dwords per ray
// (WfPool: what a march needs — origin, unit direction, t, t_light, flags — and what an event needs — colour so far,
// RNG, counters, destination, the ray direction as given) next to the scene's occupancy bitmap and the slot rings; only a
// MARCHED light feeler parks its hit normal and albedo in a 32-byte global record (WfColdGlobal).  Two kinds of work
// alternate on a slot:
//   events  (wf_event) 64-lane groups of ONE bucket: hit shading per block-type class (the procedural albedo is a switch
//           over 13 block types) + the single light's feeler decided on the spot where k_light_visibility's classes allow,
//           light hit / miss, light-feeler result, bounce set-up (whose first voxel step the event lane takes itself), texel
//           store and slot release, refill (a new ray into a free slot: 48 B ProbeRay record in, first march set up)
//   marches bursts of 24 predicated, fully unrolled voxel steps (the fast march: 12); a lane takes another march when
//           its own ends
// The round kernel alternates them in phases over compacted slot lists (C sort, D events, B march, stragglers parked for
// the next round); the queue kernel lets every ray run at its own pace.
// ------------------------------------------------------------------------------------------------
#include <algorithm>
#include <cstdlib>

#include "ddgi_device.h"
#include "ddgi_oct.h"

namespace ddgi {

constexpr int kWfStepsPerTrip = 16;  // voxel steps per march-loop trip (burst)
#ifndef DDGI_EXP_ALBEDO_CONST
#define DDGI_EXP_ALBEDO_CONST 0  // timing experiment (WRONG colours): every block is grey — what the albedo's arithmetic and table lookups cost together
#endif
#ifndef DDGI_AQ_STEPS
#define DDGI_AQ_STEPS 24
#endif
constexpr int kAqStepsPerTrip = DDGI_AQ_STEPS;  // the same for k_probe_trace_aq
#ifndef DDGI_LANE_MARCHES
#define DDGI_LANE_MARCHES 1
#endif
constexpr int kLaneMarches = DDGI_LANE_MARCHES;  // marches a lane of k_probe_trace_aq's march waves steps in turn (independent chains interleaved)
constexpr int kWfTailSteps = 1;          // straggler trips (bursts) after the march list is drained
constexpr int kWfDrainTail = 8;      // straggler trips once no new ray can be claimed (8 x 16 steps >= kMarchIters)
constexpr int kWfFetchLanes = 8;     // pull new march tasks once this many lanes are idle
constexpr uint32_t kWfChunk = 4096;  // rays a workgroup claims at a time from the global counter (upper bound, see wf_chunk)
constexpr int kWfBuckets = 8;

enum : uint32_t
{
    kSlotEmpty = 0,
    kSlotMarch = 1,
    kSlotEvPrimary = 2,
    kSlotEvFeeler = 3,
    // flags word: [1:0] state, [2] feeler march, [3] voxel hit, [11:4] march iterations,
    //             [15:12] light id + 1, [19:16] block type of the voxel hit, [20] routing hint: dead feeler expected
    kFlagFeeler = 4,
    kFlagHit = 8,
    kFlagDeadHint = 1u << 20,
};

// REF mode, frames in flight (k_probe_trace_aq): WfCold::dst = texel index | (update's distance from the launch's own) << 30
constexpr uint32_t kDstPairShift = 29;  // (kAqChainMax = 8 pairs: three bits)
constexpr uint32_t kDstTexelMask = (1u << kDstPairShift) - 1u;

struct WfShared  // control block at the start of dynamic LDS (32 dwords)
{
    uint32_t n_march[2];  // fill counts of the two march lists (this round's / next round's)
    uint32_t head_march, cur, end, live, group_head, n_groups;
    uint32_t bucket_count[kWfBuckets + 1];  // entries per event bucket
    uint32_t bucket_base[kWfBuckets + 1];   // first event-list index of each bucket
    uint32_t pad[32 - 8 - 2 * (kWfBuckets + 1)];
};
static_assert(sizeof(WfShared) == 32 * 4, "control block is 32 dwords");

// Shading state of a pool slot — what an event needs besides the march state: touched only by event groups, never
// by the march loop.  During an event it is this struct (registers); between events it is split:
//   LDS     col, rng, cnt, dst, and hc while a PRIMARY march is in flight (then it is the ray direction as given):
//           9 dwords per slot next to the 9 of the march state (WfPool)
//   global  hc (the albedo) and hn while a MARCHED light feeler is in flight: a 32-byte record per slot
//           (WfColdGlobal) — about one feeler per ray still is marched (ddgi_visibility.hip, wf_event)
// (All 48 bytes used to be a global record: 96 B of scratch traffic per event, 25 x the pass's algorithmic bytes.)
struct WfCold
{
    float hn[3];   // hit normal (live while the feelers of a hit are in flight)
    float hc[3];   // hit albedo; during a PRIMARY march: the ray direction exactly as given
    float col[3];  // accumulated radiance
    uint32_t rng;
    uint32_t cnt;  // [7:0] bounce, [11:8] light index, [15:12] visible lights
    uint32_t dst;  // REF: texel index; DDGI: the local ray index pl * n + i (-> ddgi_types.h: rec_dd_index, rec_rgb_index)
};
struct WfColdGlobal
{
    float hc[3], pad0;
    float hn[3], pad1;
};
static_assert(sizeof(WfColdGlobal) == kWfColdBytes, "2 x 16 bytes (ddgi_types.h: kWfColdBytes)");

DDGI_D f3 v3of(const float* a) { return f3{a[0], a[1], a[2]}; }
DDGI_D void set3(float* a, f3 v) { a[0] = v.x, a[1] = v.y, a[2] = v.z; }

struct WfPool
{
    float* ro[3];   // march origin (for a light feeler: the hit position)
    float* dn[3];   // normalize(direction): the direction the march steps along
    float* t;
    float* tl;
    uint32_t* flags;
    float* col[3];  // shading state kept in LDS (see WfCold)
    float* rd[3];   // WfCold::hc while a primary march is in flight
    uint32_t* rng;
    uint32_t* cnt;
    uint32_t* dst;
    struct WfColdGlobal* cold;  // the rest of the shading state, in global memory
    float4* dirbuf;       // accumulated direct light per slot; only when there is more than one light
    uint16_t* march_list[2];  // double buffered: stragglers and new marches are appended for the next round
    uint16_t* event_list;
};

constexpr int kPoolDwords = 18;      // LDS per slot: ro, dn, t, tl, flags + col, rd, rng, cnt, dst
constexpr int kPoolDwordsFast = 16;  // ... the fast build keeps |rd| only
constexpr int wf_dwords_per_ray(bool) { return kPoolDwords; }

DDGI_D f3 ld3(float* const* a, uint32_t i) { return f3{a[0][i], a[1][i], a[2][i]}; }
DDGI_D void st3(float* const* a, uint32_t i, f3 v)
{
    a[0][i] = v.x;
    a[1][i] = v.y;
    a[2][i] = v.z;
}

// A slot's shading state in and out of an event.  with_hn: the hit normal travels too (a marched feeler is in flight).
// kFast: while a primary march is in flight only hc[0] = |ray direction as given| is kept (the direction itself is dn).
template <bool kFast = false>
DDGI_D WfCold load_cold(const WfPool& P, uint32_t slot, bool with_hn)
{
    WfCold c;
    c.hn[0] = c.hn[1] = c.hn[2] = 0.0f;
    // (the LDS words are read whatever with_hn says and the record's loads stand in a branch of their own: written as an
    // if / else the two sides became ONE flat load through a selected pointer — generic addressing for every event's LDS reads)
    if (kFast)
        c.hc[0] = P.rd[0][slot], c.hc[1] = c.hc[2] = 0.0f;
    else
        c.hc[0] = P.rd[0][slot], c.hc[1] = P.rd[1][slot], c.hc[2] = P.rd[2][slot];
    asm volatile("" : "+v"(c.hc[0]), "+v"(c.hc[1]), "+v"(c.hc[2]));  // (pins the three ds_reads here: the optimizer would sink them into the select again)
    if (with_hn)
    {
        const uint4* q = reinterpret_cast<const uint4*>(P.cold + slot);
        const uint4 a = q[0], b = q[1];
        c.hc[0] = __uint_as_float(a.x), c.hc[1] = __uint_as_float(a.y), c.hc[2] = __uint_as_float(a.z);
        c.hn[0] = __uint_as_float(b.x), c.hn[1] = __uint_as_float(b.y), c.hn[2] = __uint_as_float(b.z);
    }
    c.col[0] = P.col[0][slot], c.col[1] = P.col[1][slot], c.col[2] = P.col[2][slot];
    c.rng = P.rng[slot], c.cnt = P.cnt[slot], c.dst = P.dst[slot];
    return c;
}
template <bool kFast = false>
DDGI_D void store_cold(const WfPool& P, uint32_t slot, const WfCold& c, bool with_hn)
{
    if (with_hn)
    {
        uint4* q = reinterpret_cast<uint4*>(P.cold + slot);
        q[0] = uint4{__float_as_uint(c.hc[0]), __float_as_uint(c.hc[1]), __float_as_uint(c.hc[2]), 0u};
        q[1] = uint4{__float_as_uint(c.hn[0]), __float_as_uint(c.hn[1]), __float_as_uint(c.hn[2]), 0u};
    }
    else if (kFast)
        P.rd[0][slot] = c.hc[0];
    else
        P.rd[0][slot] = c.hc[0], P.rd[1][slot] = c.hc[1], P.rd[2][slot] = c.hc[2];
    P.col[0][slot] = c.col[0], P.col[1][slot] = c.col[1], P.col[2][slot] = c.col[2];
    P.rng[slot] = c.rng, P.cnt[slot] = c.cnt, P.dst[slot] = c.dst;
}

// A value of lane `src` (wave-uniform: a constant, or derived from a ballot) for every lane: v_readlane_b32 into a scalar register.
// (__shfl(v, src) is a ds_bpermute_b32 — a trip through the LDS crossbar and a wait for it, on the queue kernel's serial path
// between two events four times per group.)
DDGI_D uint32_t lane_bcast(uint32_t v, int src) { return static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(v), src)); }

// Appends one entry per lane with pred to a list whose fill count is *counter; returns the lane's
// index (valid only where pred).  One LDS atomic per wave.
DDGI_D uint32_t wave_append(bool pred, uint32_t* counter, int lane)
{
    const unsigned long long mask = __ballot(pred);
    if (mask == 0ull) return 0u;
    const uint32_t cnt = static_cast<uint32_t>(__popcll(mask));
    uint32_t base = 0;
    const int leader = __ffsll(static_cast<long long>(mask)) - 1;
    if (lane == leader) base = atomicAdd(counter, cnt);
    base = lane_bcast(base, leader);
    return base + static_cast<uint32_t>(__popcll(mask & ((1ull << lane) - 1ull)));
}

// Which event bucket a block type's hit shading belongs to; buckets are processed in this order,
// most expensive first (ddgi_scene.h: block_albedo).
DDGI_D uint32_t shade_bucket(int type)
{
    // few, well filled buckets: every bucket ends in one partially filled 64-lane group per round
    if (type == 9) return 0;                 // mushroom stem: two fbm's, lattice mostly outside the LUT
    if (type == 10) return 1;                // cave wall (72 % of the cave's hits)
    if (type >= 11 && type <= 13) return 2;  // cave ground, moss, mold: lattice-noise lookups
    return 3;                                // mushroom caps (worley, dots) and the flat colours
}
constexpr uint32_t kBucketNoBlock = 4;  // primary march that ended on a light sphere or missed
constexpr uint32_t kBucketFeeler = 5;
constexpr uint32_t kBucketDead = 6;    // block hit whose light feeler is expected to be dead (see dead_feeler_hint): no albedo, no feeler
constexpr uint32_t kBucketRefill = 7;  // an empty slot that can take a new ray

// ROUTING HINT, never a result: will the single light's feeler of this block hit be dead, i.e. is the Lambert term
// clamp(dot(n, to_light), 0, 1) == 0 (then wf_event skips albedo, feeler march and feeler event — every outcome
// adds exactly +0)?  For an axis normal n = +-e_k the sign of dot(n, normalize(light - hit)) is the sign of
// +-(light_k - hit_k).  Hits expected dead are shaded in groups of their own (kBucketDead), so that a group
// runs EITHER the albedo + feeler set-up OR the bounce set-up, not both at half occupancy.  wf_event decides
// exactly, whatever the bucket; a wrong hint costs time only.  p: the march position at the hit.
DDGI_D bool dead_feeler_hint(f3 p, f3 light)
{
    const f3 cell = cell_id(p);
    const float dx = p.x - (cell.x - 0.5f), dy = p.y - (cell.y - 0.5f), dz = p.z - (cell.z - 0.5f);
    const float ax = fabsf(dx), ay = fabsf(dy), az = fabsf(dz);
    float d = dz, pk = p.z, lk = light.z;
    if (ax >= ay && ax >= az) d = dx, pk = p.x, lk = light.x;
    else if (ay >= az) d = dy, pk = p.y, lk = light.y;
    const float to_light = lk - pk;  // (the hit position is p + 0.001 n: irrelevant at this resolution)
    return d > 0.0f ? to_light <= 0.001f : to_light >= -0.001f;
}

// flags bits [20:16] + [3] of a march that ended in an occupied voxel of block type `type` at position p
template <class Cfg, class Upd>
DDGI_D uint32_t hit_flags(const TraceArgs& A, const Upd& U, uint32_t type, f3 p, bool feeler)
{
    uint32_t f = kFlagHit | (type << 16);
    if (Cfg::nl(A) == 1 && !feeler && type != 12u && type != 13u && dead_feeler_hint(p, U.light_pos(0))) f |= kFlagDeadHint;  // (12, 13: albedo may be NaN, see wf_event)
    return f;
}
DDGI_D uint32_t primary_bucket(uint32_t fl) { return (fl & kFlagDeadHint) ? kBucketDead : shade_bucket(static_cast<int>((fl >> 16) & 15u)); }

// A new voxel march for pool slot `slot` (intersect_scene's set-up, intersection.glsl:1253-1279 +
// grid_march's, 1053-1058): origin, normalised direction and its reciprocal, light spheres.
// What the shared event code may know at compile time.  CfgRuntime reads everything from the arguments;
// CfgPlain<kMode> is the common case — one light, no profiling switches, REF (0) or DDGI (1) output —
// whose light loops and mode branches fold away (fewer instructions, far fewer scalar registers to spill).
#ifndef DDGI_VIS_EARLY
#define DDGI_VIS_EARLY 1  // the feeler class of a hit is asked for before its albedo is evaluated (wf_event)
#endif
#ifndef DDGI_INLINE_FEELER
#define DDGI_INLINE_FEELER 0  // inline steps of a FEELER's march where the bounce rays take more than one (0: as many)
#endif
#ifndef DDGI_INLINE_STEPS
#define DDGI_INLINE_STEPS 2  // voxel steps an event lane takes itself for the march it sets up (one light; see wf_post_march)
#endif
// kFast: the tolerance-mode build (ddgi_device.h: fast_march_step) — the scene in LDS is the 2-bit skip field instead of the
// occupancy bitmap, a slot keeps |rd| instead of rd (16 dwords), the rings hold 1536 entries.
// kRecords: the queue kernel reads the per-update part of its arguments (ddgi_types.h: UpdK) from the ring of per-update records —
// the DDGI instantiations (rotation, key and lights differ from update to update) and the generic one.  The REF instantiations
// read their own arguments: the reference's live path re-submits the same work every frame, a launch goes on with an update
// only if that part is equal byte for byte (ddgi_engine.cpp: plan_hash) — and arguments cost a C3 update 2.5 % less than records
// (they can be re-loaded anywhere; a record's values are held in registers or spilled).
template <bool kFastT>
struct CfgRuntimeT
{
    static constexpr int kNl = 0;
    static constexpr bool kFast = kFastT;
    static constexpr int kInline = 1;
    static constexpr bool kRecords = true;
    static DDGI_D int nl(const TraceArgs& A) { return A.nl; }
#ifdef DDGI_PROFILING  // the ablation / fault-injection switches exist only in the profiling build (make prof)
    static DDGI_D int ablate(const TraceArgs& A) { return A.ablate; }
#else
    static DDGI_D int ablate(const TraceArgs&) { return 0; }
#endif
    static DDGI_D bool ddgi(const TraceArgs& A) { return A.ddgi != 0; }
};
using CfgRuntime = CfgRuntimeT<false>;
// CfgMulti<kMode>: any number of lights (read from the arguments), everything else as CfgPlain — BASELINE's S-Dyn configuration
// (4 animated lights) would otherwise run the fully generic instantiation with its run-time pool and mode.
// kNlT > 0: that many lights, known at compile time (4: the reference's commented cave table, S-Dyn's) — the light loops unroll.
template <int kMode, bool kFastT = false, int kNlT = 0>
struct CfgMulti
{
    static constexpr int kNl = kNlT;
    static constexpr bool kFast = kFastT;
    static constexpr int kInline = 1;
    static constexpr bool kRecords = kMode != 0;
    static DDGI_D int nl(const TraceArgs& A) { return kNlT > 0 ? kNlT : A.nl; }
    static DDGI_D int ablate(const TraceArgs&) { return 0; }
    static DDGI_D bool ddgi(const TraceArgs&) { return kMode != 0; }
};
template <int kMode, bool kFastT = false>
struct CfgPlain
{
    static constexpr int kNl = 1;
    static constexpr bool kFast = kFastT;
    static constexpr int kInline = DDGI_INLINE_STEPS;
    static constexpr bool kRecords = kMode != 0;
    static DDGI_D int nl(const TraceArgs&) { return 1; }
    static DDGI_D int ablate(const TraceArgs&) { return 0; }
    static DDGI_D bool ddgi(const TraceArgs&) { return kMode != 0; }
};

// The first Cfg::kInline voxel steps are taken right here, by the event lane that sets the march up: 57 % of the cave
// workload's marches end with their FIRST step (probes inside rock, rays that start in a corner of the surface's relief),
// and for those a trip through the march queue and a 24-step burst is all overhead.  More than one inline step does not
// pay: only another 4 % of the marches end within steps 2..4, and those steps run at 18 of 64 lanes (ddgi_trace_stats,
// sections "inline step"; 4 steps: 2.134 ms, 2: 2.136, 1: 2.109 on C3) — round 2's finding; with frames in flight (no drain,
// the event and the march waves both near their limits) a second inline step is worth 1.2 % on the one-light instantiations
// (C3 1.615 -> 1.595 ms; 3 steps: 1.615, 4: 1.64) and costs 0.5 % on the four-light one (S-Dyn): Cfg::kInline.  Returns -1 when the march goes on (the slot is
// ready for the march queue, resuming at (t, iterations) like a parked march), else the event bucket of the finished march
// (the slot is in its event state).

// Profiling aid of the counters build of the queue kernel (kStats): how many lanes are active where.  at(id) is called
// at the start of a section of event code: every active lane counts itself, the first active lane counts the visit.
#ifdef DDGI_LAP  // analysis build: the probes are lap timers (cycles per section of a wave's instruction stream)
constexpr int kProbeSections = 16;
#else
constexpr int kProbeSections = 12;
#endif
struct LaneProbe
{
    uint32_t lanes[kProbeSections] = {};
    uint32_t visits[kProbeSections] = {};
#ifdef DDGI_LAP
    uint32_t* lap = nullptr;  // this wave's LDS scratch: [0] section it is in, [1] clock at its start, [2 + s] cycles spent in section s
    DDGI_D void at(int id)
    {
        const unsigned long long m = __ballot(true);
        lanes[id] += 1u;
        if ((__ffsll(static_cast<long long>(m)) - 1) == static_cast<int>(threadIdx.x & 63))
        {
            const uint32_t now = static_cast<uint32_t>(__builtin_readcyclecounter());
            lap[2 + lap[0]] += now - lap[1];
            lap[0] = static_cast<uint32_t>(id);
            lap[1] = static_cast<uint32_t>(__builtin_readcyclecounter());
        }
    }
#else
    DDGI_D void at(int id)
    {
        const unsigned long long m = __ballot(true);
        lanes[id] += 1u;
        if ((__ffsll(static_cast<long long>(m)) - 1) == static_cast<int>(threadIdx.x & 63)) visits[id] += 1u;
    }
#endif
};
#define DDGI_PROBE(lp, id) \
    do                    \
    {                     \
        if (lp) (lp)->at(id); \
    } while (0)

struct InlineEnd  // how a march that ended within its first (inline) steps ended
{
    float t, tl;
    bool occ;
};

// kUnitDir: d is the output of a normalisation or of hemisphere_dir (every march an event posts; not the rays of a refill,
// which are the caller's) — see normalize3_of_unit
template <class Cfg, bool kUnitDir = false, class Upd>
DDGI_D int wf_post_march(const WfPool& P, uint32_t slot, WfCold& c, f3 o, f3 d, bool feeler, const TraceArgs& A, const Upd& U, const uint32_t* s_bits, InlineEnd* end = nullptr,
                         bool have_spheres = false, float tl_in = 0.0f, int lid_in = -1, LaneProbe* lp = nullptr)
{
    DDGI_PROBE(lp, feeler ? 2 : 6);  // sections 2 / 6: a feeler's / a primary march's set-up
    // have_spheres: the caller has already evaluated light_spheres(o, d) -> (tl_in, lid_in)
    float tl = tl_in;
    int lid = lid_in;
    if (!have_spheres) light_spheres<Cfg::kNl>(o, d, A, U, tl, lid);
    const f3 dn = kUnitDir ? normalize3_of_unit(d) : normalize3(d);
    st3(P.ro, slot, o);
    st3(P.dn, slot, dn);
    if (!feeler)  // the hit albedo is dead until this march is shaded
    {
        if (Cfg::kFast) c.hc[0] = kUnitDir ? 1.0f : __builtin_sqrtf(dot3(d, d));  // (hemisphere samples and directions to the light are unit to an ulp)
        else set3(c.hc, d);
    }
    P.tl[slot] = tl;
    const uint32_t base_flags = (feeler ? kFlagFeeler : 0u) | (static_cast<uint32_t>(lid + 1) << 12);
    if (Cfg::kInline > 0)
    {
        // (a feeler's and a bounce ray's marches may take different numbers of inline steps: DDGI_INLINE_FEELER, 0 = the same)
        const int n_inline = (feeler && DDGI_INLINE_FEELER > 0 && Cfg::kInline > 1) ? DDGI_INLINE_FEELER : Cfg::kInline;
        const f3 hi = f3{A.scene.hi_f[0], A.scene.hi_f[1], A.scene.hi_f[2]};
        bool occ = false, fin = false;
        float t_end;
        f3 p_end;
        int cell_end;
        if (Cfg::kFast)
        {
            // (s_bits is the skip field here; the first step is grid_march's own — nothing is known about the start voxel yet)
            FastMarch m;
            fast_march_begin(m, o, dn, 0.0f, tl);
            DDGI_PROBE(lp, 8);
            occ = fast_march_step(m, A.scene, s_bits, hi) == 0u;
            fin = occ | (m.t >= m.tl);
            t_end = m.t, p_end = m.p, cell_end = m.cell;
        }
        else
        {
            March m;
            m.ro = o, m.rd = d, m.dn = dn;
            m.inv = f3{axis_inv(dn.x), axis_inv(dn.y), axis_inv(dn.z)};
            m.cc = f3{dn.x >= 0.0f ? 1.0f : 0.0f, dn.y >= 0.0f ? 1.0f : 0.0f, dn.z >= 0.0f ? 1.0f : 0.0f};
            m.t = 0.0f, m.tl = tl, m.it = 0, m.lid = lid, m.cell = 0;
            m.p = ray_at(o, dn, 0.0f);
#pragma unroll
            for (int k = 0; k < n_inline; ++k)
                if (!fin)
                {
                    DDGI_PROBE(lp, 8 + (k < 3 ? k : 3));  // sections 8..11: inline steps 1, 2, 3, 4+
                    occ = march_step_burst(m, A.scene, s_bits, hi);
                    fin = occ | (m.t >= m.tl);
                }
            DDGI_EXP_RECLAMP(m, A.scene, hi);
            t_end = m.t, p_end = m.p, cell_end = m.cell;
        }
        P.t[slot] = t_end;
        if (fin)
        {
            // (a feeler only asks whether a block was hit, not which)
            const uint32_t hf = !occ ? 0u : (feeler ? static_cast<uint32_t>(kFlagHit) : hit_flags<Cfg>(A, U, static_cast<uint32_t>(hit_block_type(A.scene, A.scene_id, cell_id(p_end), cell_end)), p_end, false));
            P.flags[slot] = base_flags | (feeler ? kSlotEvFeeler : kSlotEvPrimary) | hf;
            if (end) end->t = t_end, end->tl = tl, end->occ = occ;
            const bool block_wins = occ && (t_end < tl);
            return static_cast<int>(feeler ? kBucketFeeler : (block_wins ? primary_bucket(hf) : kBucketNoBlock));
        }
        P.flags[slot] = kSlotMarch | base_flags | (static_cast<uint32_t>(Cfg::kFast ? 1 : n_inline) << 4);
        return -1;
    }
    P.t[slot] = 0.0f;
    P.flags[slot] = kSlotMarch | base_flags;
    return -1;
}

// DDGI mode, bounce 0: the probe ray's hit distance, clamped as the depth blend wants it, and its square
// (dst: a ray of a CHAINED update carries its update in dst[31:29] — A.pair_words != 0 says that launches of this handle chain)
template <class Upd>
DDGI_D void wf_store_distance(const TraceArgs& A, const Upd& U, uint32_t dst, float t)
{
    const float d = gl_min(t, static_cast<float>(A.grid.side) * 1.5f);
    if (A.pair_words) dst &= kDstTexelMask;
    const uint32_t n = static_cast<uint32_t>(A.grid.n), pl = dst / n, i = dst - pl * n;
    const uint32_t n_pad = rec_ray_pad(n);
    float* rad_dd = U.rad_dd();
    rad_dd[rec_dd_index(pl, i, n_pad, 0)] = d, rad_dd[rec_dd_index(pl, i, n_pad, 1)] = d * d;
}

template <class Cfg, class Upd>
DDGI_D void wf_finish_ray(const WfPool& P, uint32_t slot, f3 color, uint32_t dst, const TraceArgs& A, const Upd& U)
{
    const f3 c = div3(color, static_cast<float>(A.max_bounces));  // Q14: always /max_bounces
    if (Cfg::ddgi(A))
    {
        if (A.pair_words) dst &= kDstTexelMask;  // (a chained update's ray: its update's ray records are U's)
        const uint32_t n = static_cast<uint32_t>(A.grid.n), n_pad = rec_ray_pad(n), pl = dst / n, i = dst - pl * n;  // (the distance was written at bounce 0)
        float* rec = U.rad_rgb() + rec_rgb_index(pl, i, n_pad, 0);
        rec[0] = c.x, rec[static_cast<size_t>(n_pad) * 32] = c.y, rec[static_cast<size_t>(n_pad) * 64] = c.z;  // channel stride: n_pad * 32
    }
    else
    {
        // (a ray of a CHAINED update — k_probe_trace_aq, frames in flight — carries in dst[31:29] how many texture pairs after the
        // launch's own its update writes; the pairs of a handle lie pair_words apart)
        const uint32_t at = A.pair_words ? (dst & kDstTexelMask) + (dst >> kDstPairShift) * A.pair_words : dst;
        A.albedo[at] = unorm8(c.x) | (unorm8(c.y) << 8) | (unorm8(c.z) << 16) | (255u << 24);
        A.distance[at] = 0u;  // `distances` is never assigned (probe_pass.comp:276,302)
    }
    P.flags[slot] = kSlotEmpty;
}

// End of get_direct_lighting for one hit: accumulate, then bounce or finish (probe_pass.comp:286-292).
// Returns true when the ray bounces; the caller then posts the march (o, d) — every path of an event
// group shares ONE wf_post_march call site, so divergent lanes do not run its code twice.
template <class Cfg, class Upd>
DDGI_D bool wf_lighting_done(const WfPool& P, uint32_t slot, WfCold& c, f3 contribution, f3 hpos, f3 hnrm, uint32_t cnt, const TraceArgs& A, const Upd& U, f3& o, f3& d, LaneProbe* lp = nullptr)
{
    DDGI_PROBE(lp, 5);  // section 5: accumulate + hemisphere sample
    const f3 color = v3of(c.col) + contribution;
    const uint32_t bounce = (cnt & 255u) + 1u;
    if (static_cast<int>(bounce) < A.max_bounces)
    {
        set3(c.col, color);
        c.cnt = bounce;
        o = hpos + hnrm * 0.0001f;
        d = (Cfg::ablate(A) & 2) ? normalize3(hnrm + mk3(0.3f, 0.2f, 0.1f)) : hemisphere_dir(hnrm, c.rng);
        return true;
    }
    wf_finish_ray<Cfg>(P, slot, color, c.dst, A, U);
    return false;
}

// The feeler class of the (voxel, face) a feeler would start from (k_light_visibility; ddgi_visibility.hip), or kVisUnknown
// when there is no table, the origin is outside the baked box, or it is not where the table's guarantee holds: 5e-4 ..
// 1.5e-3 off the face that was hit (it is put 1e-3 off it) and at least 5e-4 inside the voxel on the two other axes.
// n: the hit's axis normal (points from the block that was hit into the voxel the feeler starts in).
// (light_vis_entry: the table index of that (voxel, face), or -1)
DDGI_D int light_vis_entry(const TraceArgs& A, f3 o, f3 n)
{
    const SceneK& S = A.scene;
    const f3 cell = cell_id(o);
    const float ux = cell.x - o.x, uy = cell.y - o.y, uz = cell.z - o.z;  // in [0, 1): distance to the voxel's upper faces
    // along the normal's axis: distance to the face that was hit; along the others: distance to the nearer face
    const float fx = n.x > 0.0f ? 1.0f - ux : ux, fy = n.y > 0.0f ? 1.0f - uy : uy, fz = n.z > 0.0f ? 1.0f - uz : uz;
    const bool ax = n.x != 0.0f, ay = n.y != 0.0f;
    const float off = ax ? fx : (ay ? fy : fz);
    const float l1 = ax ? uy : ux, l2 = (ax || ay) ? uz : uy;
    const bool inside = off >= 5.0e-4f && off <= 1.5e-3f && fminf(l1, l2) >= 5.0e-4f && fmaxf(l1, l2) <= 1.0f - 5.0e-4f && cell.x >= S.lo_f[0] && cell.x <= S.hi_f[0] &&
                        cell.y >= S.lo_f[1] && cell.y <= S.hi_f[1] && cell.z >= S.lo_f[2] && cell.z <= S.hi_f[2];  // (false for NaN)
    if (!inside) return -1;
    const int idx = static_cast<int>(fmaf(cell.z, S.nxy_f, fmaf(cell.y, S.nx_f, cell.x))) - S.bias;
    const int face = (ax ? 0 : (ay ? 2 : 4)) + ((n.x + n.y + n.z) < 0.0f ? 1 : 0);  // 2 axis + (the block that was hit is on the + side)
    return idx * 8 + face;
}
template <class Upd>
DDGI_D uint32_t light_vis_class(const TraceArgs& A, const Upd& U, f3 o, f3 n, int& entry)
{
    entry = 0;
    const uint8_t* vis = U.vis();
    if (!vis) return kVisUnknown;
    const int en = light_vis_entry(A, o, n);
    if (en < 0) return kVisUnknown;
    entry = en;
    return vis[static_cast<uint32_t>(en)];  // (an unsigned 32-bit offset: the table's base stays in scalar registers)
}


// A feeler that starts in a patch of class kVisListed (ddgi_visibility.hip) and did not end within its first step: its own ray
// (o, unit direction dn, light sphere at t_light) against the patch's list of occupied voxels — everything else its march can look
// up is empty.  True: the march certainly lands in no occupied voxel (it reaches the light); false: undecided, march it.  The
// arithmetic here decides nothing about the result's bits, it only has to be CONSERVATIVE: a march looks up the voxel that holds
// its position, positions lie within 3e-5 of the line o + t dn (|p| < 2^10), so a line that misses a voxel grown by kListGrow on
// every side (ten times that) never has a position inside it; every "misses" needs a comparison that comes out true (a NaN
// decides nothing).
constexpr float kListGrow = 4.0e-4f;
template <class Upd>
DDGI_D bool listed_feeler_clear(const Upd& U, int entry, f3 o, f3 dn, float t_light)
{
    const uint4 packed4 = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(U.vis_occ()) + static_cast<uint32_t>(entry) * static_cast<uint32_t>(kVisListMax * sizeof(uint32_t)));
    const uint32_t packed[kVisListMax] = {packed4.x, packed4.y, packed4.z, packed4.w};
    const f3 cell = cell_id(o);  // the start voxel
    const f3 inv{__builtin_amdgcn_rcpf(dn.x), __builtin_amdgcn_rcpf(dn.y), __builtin_amdgcn_rcpf(dn.z)};
    bool all_clear = true;
#pragma unroll
    for (int k = 0; k < kVisListMax; ++k)
    {
        const uint32_t u = packed[k];
        if (u == kVisListEnd) continue;
        const int dx = static_cast<int>(u & 1023u) - 512, dy = static_cast<int>((u >> 10) & 1023u) - 512, dz = static_cast<int>(u >> 20) - 512;
        const f3 hi{cell.x + static_cast<float>(dx), cell.y + static_cast<float>(dy), cell.z + static_cast<float>(dz)};  // the voxel covers (hi - 1, hi]
        // slab parameters of the faces (lo - o) / dn, (hi - o) / dn; a zero component gives +-inf or NaN, and NaN decides nothing
        const f3 ta{(hi.x - 1.0f - o.x) * inv.x, (hi.y - 1.0f - o.y) * inv.y, (hi.z - 1.0f - o.z) * inv.z};
        const f3 tb{(hi.x - o.x) * inv.x, (hi.y - o.y) * inv.y, (hi.z - o.z) * inv.z};
        // growing / shrinking the box by m moves a face's parameter by m / |dn_axis|
        const f3 ai{fabsf(inv.x), fabsf(inv.y), fabsf(inv.z)};
        const f3 tn{fminf(ta.x, tb.x), fminf(ta.y, tb.y), fminf(ta.z, tb.z)}, tf{fmaxf(ta.x, tb.x), fmaxf(ta.y, tb.y), fmaxf(ta.z, tb.z)};
        // grown box: [tn - g, tf + g] per axis
        const float g_in = fmaxf(fmaxf(tn.x - kListGrow * ai.x, tn.y - kListGrow * ai.y), tn.z - kListGrow * ai.z);
        const float g_out = fminf(fminf(tf.x + kListGrow * ai.x, tf.y + kListGrow * ai.y), tf.z + kListGrow * ai.z);
        const bool clear = (g_in > g_out) || (g_out < 0.0f) || (g_in > t_light);
        all_clear = all_clear && clear;
    }
    return all_clear;
}

// get_direct_lighting's loop body for the light L once its feeler's outcome is known (probe_pass.comp:194-207):
// the feeler reached the light (any_hit, !block_wins): direct += lambert * col * I / dist, one more visible light;
// it hit a block: early return 0.2 * base * lambert (Q10).  hpos: the feeler's origin; nh: the hit normal as
// get_direct_lighting normalises it.  Shared by the feeler event and the table-decided feelers of a primary event.
DDGI_D void feeler_outcome(const LightK& L, f3 hpos, f3 nh, f3 hcol, bool any_hit, bool block_wins, f3& direct, int& nvis, f3& contribution, bool& early)
{
    if (!any_hit) return;
    const f3 lp{L.pos[0], L.pos[1], L.pos[2]};
    const float lambert = gl_clamp(dot3(nh, normalize3(lp - hpos)), 0.0f, 1.0f);
    if (!block_wins)
    {
        const float dist = length3(lp - hpos);
        const f3 lc{L.col[0], L.col[1], L.col[2]};
        direct = direct + div3((lc * lambert) * L.intensity, dist);
        nvis += 1;
    }
    else
    {
        contribution = (hcol * 0.2f) * lambert;  // Q10 early return
        early = true;
    }
}

// Several lights (get_direct_lighting's loop, probe_pass.comp:186-207): the lights from `li` on whose feeler from hpos is decided
// by their table (k_light_visibility, one table per light for the first kVisLights) are dealt with right here — no march, no
// queue trip, no event.  Returns the first light whose feeler has to be marched, or the number of lights when the loop is over
// (all decided, or the early return of a blocked feeler: `early`).
//   kVisLit     no occupied voxel between the hit and light li's layer: the feeler ends on the nearest light sphere its ray meets
//               (its own at the latest — the ray points at its centre), as for a single light.
//   kVisShadow  the feeler lands in an occupied voxel before light li's sphere — unless ANOTHER light's sphere lies on the ray in
//               front of it: then the reference's closest hit may be that sphere.  Used only when the nearest sphere on the ray is
//               li's own (or none); otherwise the feeler is marched.
//   kVisListed (light 0's table has them) counts as unknown here.
// The table's guarantee is about the start patch (voxel, face) only, which all of a hit's feelers share: one entry for all lights.
template <class Cfg, class Upd>
DDGI_D int decided_feelers(const TraceArgs& A, const Upd& U, int li, bool on_axis_face, f3 hpos, f3 hnrm, f3 nh, f3 hcol, f3& direct, int& nvis, f3& contribution, bool& early)
{
    const int nl = Cfg::nl(A);
    if (!on_axis_face || !U.vis()) return li;
    const int en = light_vis_entry(A, hpos, hnrm);
    if (en < 0) return li;
    while (li < nl && li < kVisLights)
    {
        const uint8_t* table = li == 0 ? U.vis() : U.vis_more(li - 1);
        const uint32_t cls = table ? table[en] : kVisUnknown;
        if (cls != kVisLit && cls != kVisShadow) break;
        const LightK L = U.light(li);
        const f3 to_light = normalize3(f3{L.pos[0], L.pos[1], L.pos[2]} - hpos);
        float ftl = __builtin_inff();
        int flid = -1;
        light_spheres<Cfg::kNl>(hpos, to_light, A, U, ftl, flid);
        bool any, block;
        if (cls == kVisLit)
            block = false, any = ftl < __builtin_inff();
        else
        {
            if (flid >= 0 && flid != li) break;  // another light's sphere in front: march it
            block = any = true;
        }
        feeler_outcome(L, hpos, nh, hcol, any, block, direct, nvis, contribution, early);
        if (early) return nl;
        ++li;
    }
    return li;
}

// One event of pool slot `slot` in bucket b (ddgi_trace_wf.hip: shade_bucket): shades a finished march /
// starts local ray r in an empty slot (b == kBucketRefill).  Returns true when the slot has a new march
// posted that goes on (return 1: its state is in the pool arrays, its shading record stored), when the new
// march already ended within its first steps (return 2 + the event bucket it now waits in), or 0 when the ray
// is finished (it has written its output and left the slot empty).
template <class Cfg, class Upd>
DDGI_D int wf_event(const TraceArgs& A, const Upd& U, const WfPool& P, const uint32_t* s_bits, uint32_t b, uint32_t slot, uint32_t r, bool r_valid, LaneProbe* lp = nullptr, uint32_t dst_tag = 0u)
{
    DDGI_PROBE(lp, 0);  // section 0: the event (all lanes of the group)
    const GridK& G = A.grid;
    const int rays_per_probe = G.n;
    const float inf = __builtin_inff();
    const bool multi_light = Cfg::nl(A) > 1;
    bool posted = false;
    if (b == kBucketRefill)
    {
        if (r_valid)
        {
            // local (y, zl, x) probe enumeration -> reference probe index p -> global ray index
            const int pl = static_cast<int>(r / static_cast<uint32_t>(rays_per_probe));
            const int i = static_cast<int>(r) - pl * rays_per_probe;
            const int slab_row = G.czl * G.cx;
            const int y = pl / slab_row;
            const int rem = pl - y * slab_row;
            const int p = y * G.cx * G.cz + G.z0 * G.cx + rem;
            const uint32_t global_ray = static_cast<uint32_t>(p) * static_cast<uint32_t>(rays_per_probe) + static_cast<uint32_t>(i);
            f3 ray_o, ray_d;
            WfCold c;
            if (Cfg::ddgi(A))
            {
                // in-kernel ray generation: probe position + rotated spherical Fibonacci direction
                const int pxz = p - y * G.cx * G.cz;
                ray_o = probe_position(G, pxz % G.cx, y, pxz / G.cx);
                float rot[9];
                U.rot(rot);
                ray_d = fibonacci_dir(i, rays_per_probe, rot);
                c.dst = r | dst_tag;  // pl * n + i
                c.rng = wang_hash(global_ray ^ U.frame_key());
            }
            else
            {
                const float4* rec = U.rays() + 3 * static_cast<size_t>(r);
                const float4 ra = rec[0], rb = rec[1], rc = rec[2];
                ray_o = mk3(ra.x, ra.y, ra.z);
                ray_d = mk3(rb.x, rb.y, rb.z);
                const int dst_probe = static_cast<int>(rc.x);  // int(probe_info.x), probe_pass.comp:269
                c.dst = (static_cast<uint32_t>(slab_slot(G, dst_probe)) * rays_per_probe + static_cast<int>(rc.z) * G.sx + static_cast<int>(rc.y)) | dst_tag;
                c.rng = wang_hash(global_ray);  // p_idx == buffer index (probe_pass.comp:55-57)
            }
            c.cnt = 0u;
            set3(c.col, mk3(0, 0, 0));
            set3(c.hn, mk3(0, 0, 0));
            const int pb = wf_post_march<Cfg>(P, slot, c, ray_o, ray_d, false, A, U, s_bits, nullptr, false, 0.0f, -1, lp);
            store_cold<Cfg::kFast>(P, slot, c, false);
            return pb < 0 ? 1 : 2 + pb;
        }
    }
    else
    {
        WfCold c = load_cold<Cfg::kFast>(P, slot, b == kBucketFeeler);
        f3 mo = mk3(0, 0, 0), md = mk3(0, 0, 0);  // the march this event posts, if any
        // Several lights: get_direct_lighting's loop (probe_pass.comp:186-207) from light li on, as far as it gets without a march —
        // a light whose feeler is decided by its table (decided_feelers) costs nothing more; the others' feelers are set up right
        // here, and most end within their first step (wf_post_march: on the surface's own relief): the loop goes on.  The first
        // feeler that has to be marched takes the slot to the march queue with the loop's state (light index and visible lights in
        // cnt, the sum in dirbuf): returns true, and the loop resumes in that feeler's event.  False: the loop is over, the hit's
        // direct light is in `contribution`.
        auto lights_from = [&](int li, f3 hpos, f3 hnrm, f3 nh, bool on_axis_face, f3 hcol, uint32_t cnt_low, f3 direct, int nvis, f3& contribution) -> bool {
            bool early = false;
            contribution = mk3(0, 0, 0);
            for (;;)
            {
                li = decided_feelers<Cfg>(A, U, li, on_axis_face, hpos, hnrm, nh, hcol, direct, nvis, contribution, early);
                if (early || li >= Cfg::nl(A)) break;
                const LightK L = U.light(li);
                const f3 to_light = normalize3(f3{L.pos[0], L.pos[1], L.pos[2]} - hpos);
                c.cnt = cnt_low | (static_cast<uint32_t>(li) << 8) | (static_cast<uint32_t>(nvis) << 12);
                float ftl = inf;
                int flid = -1;
                light_spheres<Cfg::kNl>(hpos, to_light, A, U, ftl, flid);
                InlineEnd fe;
                const bool inline_end = wf_post_march<Cfg, true>(P, slot, c, hpos, to_light, true, A, U, s_bits, &fe, true, ftl, flid, lp) >= 0;
                if (!inline_end)
                {
                    if (multi_light) P.dirbuf[slot] = float4{direct.x, direct.y, direct.z, 0.0f};
                    store_cold<Cfg::kFast>(P, slot, c, true);  // (hn and the albedo travel with the marched feeler)
                    return true;
                }
                const bool block = fe.occ && (fe.t < fe.tl), any = block || (fe.tl < inf);
                feeler_outcome(L, hpos, nh, hcol, any, block, direct, nvis, contribution, early);
                if (early) break;
                ++li;
            }
            if (!early && nvis != 0) contribution = nvis == 1 ? hcol * direct : div3(hcol * direct, static_cast<float>(nvis));  // x / 1.0f == x
            return false;
        };
        bool as_feeler = false;
        // Every path on which get_direct_lighting has come to its end for this hit sets these and meets at ONE
        // wf_lighting_done below (the bounce set-up — hemisphere sample, new march — is the longest stretch of an event:
        // run once per group by all lanes that need it, not once per path by a few lanes each)
        bool lit_done = false;
        f3 ld_contribution = mk3(0, 0, 0), ld_hpos = mk3(0, 0, 0), ld_hnrm = mk3(0, 0, 0);
        uint32_t ld_cnt = 0u;
        const uint32_t fl = P.flags[slot];
        const float t = P.t[slot], tl = P.tl[slot];
        const f3 ro = ld3(P.ro, slot);
        const bool block_wins = (fl & kFlagHit) && (t < tl);  // temp_isect.t < closest_t
        const bool any_hit = block_wins || (tl < inf);       // closest_t < INF
        if (b != kBucketFeeler)
        {
            const bool first_bounce = Cfg::ddgi(A) && (c.cnt & 255u) == 0u;
            if (!any_hit)
            {
                if (first_bounce) wf_store_distance(A, U, c.dst, kMissDistance);
                wf_finish_ray<Cfg>(P, slot, v3of(c.col), c.dst, A, U);  // probe_pass.comp:288-290 break
            }
            else
            {
                // the ray direction as given (see WfCold::hc); the fast build keeps its length only
                const f3 rd = Cfg::kFast ? ld3(P.dn, slot) * c.hc[0] : v3of(c.hc);
                f3 nraw, hcol = mk3(0, 0, 0);  // (light-sphere hit: Q12, unassigned Material pinned to zero)
                f3 p = mk3(0, 0, 0), nn = mk3(0, 0, 0);  // block hit: march position and block normal, for the albedo
                const int type = static_cast<int>((fl >> 16) & 15u);
                float th;
                bool axis_normal = false;
                if (block_wins)
                {
                    th = t;
                    p = ray_at(ro, ld3(P.dn, slot), t);  // the march position at the hit
                    const f3 cell = cell_id(p);
                    const f3 centre = f3{cell.x - 0.5f, cell.y - 0.5f, cell.z - 0.5f};
                    // The reference picks the axis of the largest |component| of normalize(p - centre).  The
                    // normalisation multiplies all three by the same positive factor, so it can only change
                    // the choice when the two largest are within rounding of each other: unless they are
                    // (a hit on a voxel edge), compare the raw components and skip the square root and division.
                    f3 diff = p - centre;
                    {
                        const float ax = fabsf(diff.x), ay = fabsf(diff.y), az = fabsf(diff.z);
                        const float hi = fmaxf(ax, fmaxf(ay, az));
                        const float mid = fmaxf(fminf(ax, ay), fminf(fmaxf(ax, ay), az));  // the second largest
                        const bool clear = hi > 0x1.0p-60f && hi < 0x1.0p60f && mid < hi * (1.0f - 0x1.0p-21f);  // false for NaN too
                        if (!clear) diff = normalize3(diff);
                    }
                    // axis of the largest |component|, first wins on ties (:1075-1086)
                    f3 n = mk3(0, 0, 0);
                    float best = 0.0f;
                    if (fabsf(diff.x) > best) { best = fabsf(diff.x); n = mk3(gl_sign(diff.x), 0, 0); }
                    if (fabsf(diff.y) > best) { best = fabsf(diff.y); n = mk3(0, gl_sign(diff.y), 0); }
                    if (fabsf(diff.z) > best) { best = fabsf(diff.z); n = mk3(0, 0, gl_sign(diff.z)); }
                    // normalize(n) of a unit axis vector is n itself (1*(1/sqrt(1)) = 1, 0*1 = 0);
                    // n stays (0,0,0) only if diff is NaN, where the reference yields NaN as well
                    nn = best > 0.0f ? n : normalize3(n);
                    nraw = nn;
                    axis_normal = best > 0.0f;
                }
                else
                {
                    th = tl;
                    // (the light whose sphere was hit; one light: it is that one)
                    const f3 lp = U.light_pos(Cfg::kNl == 1 ? 0 : static_cast<int>((fl >> 12) & 15u) - 1);
                    nraw = ray_at((ro - lp) * 10.0f, rd * 10.0f, th);  // sphere-space position
                }
                if (first_bounce) wf_store_distance(A, U, c.dst, th);  // Isect.t of the probe ray
                const f3 hnrm = axis_normal ? nraw : normalize3(nraw);
                const f3 hpos = ray_at(ro, rd, th) + hnrm * 0.001f;
                set3(c.hn, hnrm);
                const uint32_t cnt = c.cnt & 255u;  // light index 0, no visible light yet
                if (Cfg::nl(A) > 0)
                {
                    const LightK L = U.light(0);
                    const f3 to_light = normalize3(f3{L.pos[0], L.pos[1], L.pos[2]} - hpos);
                    // Dead-feeler elimination (single light): whatever the feeler finds, the hit's
                    // direct light is scaled by lambert = clamp(dot(n, to_light), 0, 1)
                    // (probe_pass.comp:194-204), so for lambert == 0 and a finite albedo every
                    // outcome adds exactly +0 to the colour: skip the march and its event — and the albedo itself:
                    // for a hit position of ordinary magnitude (|p| < 2^20: no product in the noise functions overflows)
                    // every block type's albedo is finite, except the moss / mold pattern (types 12, 13), whose
                    // normalize() of a texel-centre offset can be 0/0; those are evaluated and checked.  (Hits expected to be dead come in groups of their own, kBucketDead,
                    // so a group normally runs either this branch or the albedo + feeler set-up.)
                    const f3 nh = axis_normal ? hnrm : normalize3(hnrm);
                    const bool lambert_zero = Cfg::nl(A) == 1 && dot3(nh, to_light) <= 0.0f && !(Cfg::ablate(A) & 4);
                    const bool ordinary = fmaxf(fabsf(p.x), fmaxf(fabsf(p.y), fabsf(p.z))) < 0x1.0p20f;
                    // The feeler class of the hit's (voxel, face) is asked for BEFORE the albedo: its index depends on the hit alone, and the
                    // byte's way from L2 passes under the albedo's arithmetic instead of standing in front of the feeler decision.
                    int vis_entry = 0;
                    const bool vis_early = DDGI_VIS_EARLY && Cfg::nl(A) == 1 && block_wins && axis_normal && !lambert_zero;
                    const uint32_t vis_early_class = vis_early ? light_vis_class(A, U, hpos, hnrm, vis_entry) : kVisUnknown;
                    if (block_wins && (!lambert_zero || type == 12 || type == 13 || !ordinary)) DDGI_PROBE(lp, 1);  // section 1: albedo
                    if (block_wins && (!lambert_zero || type == 12 || type == 13 || !ordinary))
                        hcol = ((Cfg::ablate(A) & 1) || DDGI_EXP_ALBEDO_CONST) ? mk3(0.5f, 0.5f, 0.5f) : block_albedo(p, type, nn, A.noise);
#ifdef DDGI_LAP
                    DDGI_PROBE(lp, 7);  // after the albedo: visibility class, feeler decision
#endif
                    const bool finite_albedo = fabsf(hcol.x) < inf && fabsf(hcol.y) < inf && fabsf(hcol.z) < inf;
                    set3(c.hc, hcol);
                    // Is the feeler's outcome certain (k_light_visibility)?  Then its march, queue trip and event are
                    // skipped: the light-sphere test it would start with (does the ray reach the sphere at all) and
                    // get_direct_lighting's arithmetic are evaluated right here, on the same values.
                    // (a hit with lambert == 0 whose albedo is not finite — the moss / mold pattern's 0/0 — needs the class after all)
                    const uint32_t vis = vis_early ? vis_early_class
                                                   : ((Cfg::nl(A) == 1 && block_wins && axis_normal && !(lambert_zero && finite_albedo)) ? light_vis_class(A, U, hpos, hnrm, vis_entry) : kVisUnknown);
                    if ((Cfg::ablate(A) & 16) && A.stats && block_wins) atomicAdd(&A.stats[(lambert_zero && finite_albedo) ? 59 : (vis == kVisListed ? 63 : 56 + vis)], 1ull);  // profiling build: feeler classes
                    ld_hpos = hpos, ld_hnrm = hnrm, ld_cnt = cnt;
                    if (lambert_zero && finite_albedo)
                        lit_done = true;  // contributes +0
                    else if (Cfg::nl(A) == 1 && Cfg::kInline > 0)
                    {
                        // The single light's feeler.  intersect_scene's sphere half for it (does the ray reach the light's
                        // sphere at all, and where) is evaluated once, for the table's "lit" class and for the feeler's march.
                        float ftl = inf;
                        int flid = -1;
                        if (vis != kVisShadow) DDGI_PROBE(lp, 3);  // section 3: sphere test of the feeler
                        if (vis != kVisShadow) light_spheres<Cfg::kNl>(hpos, to_light, A, U, ftl, flid);
                        bool feeler_any = true, feeler_block = true;  // kVisShadow: a block before the light (t_block < t_light, or no sphere hit at all)
                        if (vis == kVisLit) feeler_block = false, feeler_any = ftl < inf;
                        else if (vis == kVisUnknown || vis == kVisListed)
                        {
                            // The feeler is set up right here.  Most shadowed feelers end within their first steps (on the
                            // surface's own relief); then get_direct_lighting goes on in this event as well — no record round
                            // trip, no second event.  A feeler that has to be marched takes the slot to the march queue.
                            c.cnt = cnt;
                            InlineEnd fe;
                            // (a listed patch: the feeler's own ray against the few occupied voxels of its bundle, before it is queued)
                            bool clear = false;
                            const bool inline_end = wf_post_march<Cfg, true>(P, slot, c, hpos, to_light, true, A, U, s_bits, &fe, true, ftl, flid, lp) >= 0;
                            if (!inline_end && vis == kVisListed) clear = listed_feeler_clear(U, vis_entry, hpos, normalize3_of_unit(to_light), ftl);
                            if (!inline_end && !clear)
                            {
                                store_cold<Cfg::kFast>(P, slot, c, true);  // (hn and the albedo travel with the marched feeler)
                                return 1;
                            }
                            if (inline_end)
                            {
                                feeler_block = fe.occ && (fe.t < fe.tl);
                                feeler_any = feeler_block || (fe.tl < inf);
                            }
                            else
                                feeler_block = false, feeler_any = ftl < inf;  // (clear of every listed voxel: as for the class kVisLit)
                        }
                        f3 direct = mk3(0, 0, 0), contribution = mk3(0, 0, 0);
                        int nvis = 0;
                        bool early = false;
                        DDGI_PROBE(lp, 4);  // section 4: the light's contribution for a decided feeler
                        feeler_outcome(L, hpos, nh, hcol, feeler_any, feeler_block, direct, nvis, contribution, early);
                        if (!early && nvis != 0) contribution = hcol * direct;  // one visible light: x / 1.0f == x
                        ld_contribution = contribution, lit_done = true;
                    }
                    else
                    {
                        // several lights: the loop over the lights runs right here as far as it can (lights_from)
                        f3 contribution = mk3(0, 0, 0);
                        c.cnt = cnt;
                        if (lights_from(0, hpos, hnrm, nh, block_wins && axis_normal, hcol, cnt, mk3(0, 0, 0), 0, contribution)) return 1;
                        ld_contribution = contribution, lit_done = true;
                    }
                }
                else
                {
                    set3(c.hc, hcol);  // no light: the albedo is never used
                    ld_hpos = hpos, ld_hnrm = hnrm, ld_cnt = cnt, lit_done = true;
                }
            }
        }
        else  // a light feeler came back: get_direct_lighting's loop body (probe_pass.comp:186-207)
        {
            const f3 hpos = ro;  // a feeler starts at the hit position
            const f3 hnrm = v3of(c.hn), hcol = v3of(c.hc);
            const uint32_t cnt = c.cnt;
            int li = static_cast<int>((cnt >> 8) & 15u);
            int nvis = static_cast<int>((cnt >> 12) & 15u);
            f3 direct = mk3(0, 0, 0);
            if (multi_light)
            {
                const float4 dv = P.dirbuf[slot];
                direct = mk3(dv.x, dv.y, dv.z);
            }
            f3 contribution = mk3(0, 0, 0);
            bool early = false;
            if ((Cfg::ablate(A) & 16) && A.stats) atomicAdd(&A.stats[block_wins ? 61 : (any_hit ? 60 : 62)], 1ull);  // profiling build: outcomes of marched feelers
            {
                const bool is_axis = (fabsf(hnrm.x) + fabsf(hnrm.y) + fabsf(hnrm.z) == 1.0f) &&
                                     (fabsf(hnrm.x) == 1.0f || fabsf(hnrm.y) == 1.0f || fabsf(hnrm.z) == 1.0f);
                const f3 nh = is_axis ? hnrm : normalize3(hnrm);  // identity for a unit axis vector
                feeler_outcome(U.light(Cfg::kNl == 1 ? 0 : li), hpos, nh, hcol, any_hit, block_wins, direct, nvis, contribution, early);
            }
            li += 1;
            if (!early && li < Cfg::nl(A))
            {
                const bool is_axis = (fabsf(hnrm.x) + fabsf(hnrm.y) + fabsf(hnrm.z) == 1.0f) &&
                                     (fabsf(hnrm.x) == 1.0f || fabsf(hnrm.y) == 1.0f || fabsf(hnrm.z) == 1.0f);
                const f3 nh = is_axis ? hnrm : normalize3(hnrm);
                if (lights_from(li, hpos, hnrm, nh, is_axis, hcol, cnt & 255u, direct, nvis, contribution)) return 1;
            }
            else if (!early && nvis != 0)
                contribution = nvis == 1 ? hcol * direct : div3(hcol * direct, static_cast<float>(nvis));  // x / 1.0f == x
            ld_contribution = contribution, ld_hpos = hpos, ld_hnrm = hnrm, ld_cnt = cnt, lit_done = true;
        }
        if (lit_done) posted = wf_lighting_done<Cfg>(P, slot, c, ld_contribution, ld_hpos, ld_hnrm, ld_cnt, A, U, mo, md, lp);
        if (posted)
        {
            const int pb = wf_post_march<Cfg, true>(P, slot, c, mo, md, as_feeler, A, U, s_bits, nullptr, false, 0.0f, -1, lp);  // md: to_light or a hemisphere sample
#ifdef DDGI_LAP
            DDGI_PROBE(lp, 12);  // write-back
#endif
            store_cold<Cfg::kFast>(P, slot, c, as_feeler);  // the slot lives on: write its shading state back
            return pb < 0 ? 1 : 2 + pb;
        }
    }
    return 0;
}

// T lanes per workgroup, kBlocksPerCU workgroups resident per CU (T * kBlocksPerCU = 1024 lanes = 4 waves/SIMD)
template <int T, int kBlocksPerCU, bool kStats>
__global__ __launch_bounds__(T, T * kBlocksPerCU / 256) void k_probe_trace_wf(const TraceArgs A, const int pool_size, uint32_t* __restrict__ work_counter)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t wf_lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const uint32_t PS = static_cast<uint32_t>(pool_size);
    const bool multi_light = A.nl > 1;
    const int tail_steps = A.wf_tail > 0 ? A.wf_tail : kWfTailSteps;
    const int drain_tail = A.wf_drain > 0 ? A.wf_drain : kWfDrainTail;
    const int fetch_lanes = A.wf_fetch > 0 ? A.wf_fetch : kWfFetchLanes;

    // ---- carve LDS: control block | occupancy bitmap | pool arrays | slot lists ----
    WfShared* sh = reinterpret_cast<WfShared*>(wf_lds);
    uint32_t* s_bits = wf_lds + 32;
    uint32_t* cursor = s_bits + ((A.scene.nwords + 3) & ~3);
    WfPool P;
    auto takef = [&]() { float* p = reinterpret_cast<float*>(cursor); cursor += PS; return p; };
    auto takeu = [&]() { uint32_t* p = cursor; cursor += PS; return p; };
    for (int a = 0; a < 3; ++a) P.ro[a] = takef();
    for (int a = 0; a < 3; ++a) P.dn[a] = takef();
    P.t = takef();
    P.tl = takef();
    P.flags = takeu();
    for (int a = 0; a < 3; ++a) P.col[a] = takef();
    for (int a = 0; a < 3; ++a) P.rd[a] = takef();
    P.rng = takeu(), P.cnt = takeu(), P.dst = takeu();
    P.cold = static_cast<WfColdGlobal*>(A.wf_cold) + static_cast<size_t>(blockIdx.x) * PS;
    P.dirbuf = multi_light ? A.wf_dir + static_cast<size_t>(blockIdx.x) * PS : nullptr;
    P.march_list[0] = reinterpret_cast<uint16_t*>(cursor);
    P.march_list[1] = P.march_list[0] + PS;
    P.event_list = P.march_list[1] + PS;

    for (int i = tid; i < A.scene.nwords; i += T) s_bits[i] = A.scene.bits[i];
    for (uint32_t i = tid; i < PS; i += T) P.flags[i] = kSlotEmpty;
    const uint32_t chunk = static_cast<uint32_t>(A.wf_chunk);
    const uint32_t n_chunks = (A.n_rays + chunk - 1) / chunk;
    if (tid == 0)
    {
        sh->n_march[0] = sh->n_march[1] = sh->head_march = 0;
        sh->live = 0;
        sh->group_head = sh->n_groups = 0;
        const uint32_t c = atomicAdd(work_counter, 1u);
        sh->cur = c < n_chunks ? c * chunk : 0u;
        sh->end = c < n_chunks ? min(c * chunk + chunk, A.n_rays) : 0u;
    }
    if (tid <= kWfBuckets) sh->bucket_count[tid] = 0u;
    __syncthreads();

    const float inf = __builtin_inff();
    unsigned long long st_trips = 0, st_lane_steps = 0, st_groups = 0, st_lane_events = 0, st_iters = 0, st_fetches = 0;
    long long cy[6] = {0, 0, 0, 0, 0, 0};
    long long cy_bucket[kWfBuckets] = {};
    unsigned long long n_bucket[kWfBuckets] = {};

    for (uint32_t round = 0;; ++round)
    {
        const uint32_t cur_list = round & 1u, next_list = cur_list ^ 1u;
        // ================= C: bucket every slot by its next action =================
        const long long c0 = kStats ? clock64() : 0;
        const uint32_t ray_cur = sh->cur, ray_end = sh->end;
        const bool have_rays = ray_cur < ray_end;
        uint32_t my_bucket[3], my_rank[3];  // a lane looks after at most 3 pool slots
        {
            bool any_live = false;
            int k = 0;
            for (uint32_t slot = tid; slot < PS; slot += T, ++k)
            {
                const uint32_t fl = P.flags[slot];
                const uint32_t st = fl & 3u;
                uint32_t b = kWfBuckets;  // none (a march in flight)
                if (st == kSlotEvFeeler) b = kBucketFeeler;
                else if (st == kSlotEvPrimary)
                {
                    const bool block_wins = (fl & kFlagHit) && (P.t[slot] < P.tl[slot]);
                    b = block_wins ? primary_bucket(fl) : kBucketNoBlock;
                }
                else if (st == kSlotEmpty && have_rays) b = kBucketRefill;
                my_bucket[k] = b;
                my_rank[k] = (b < kWfBuckets) ? atomicAdd(&sh->bucket_count[b], 1u) : 0u;
                any_live |= (st != kSlotEmpty) || have_rays;
            }
            if (__ballot(any_live) != 0ull && lane == 0) sh->live = 1u;
        }
        __syncthreads();
        if (sh->live == 0u) break;  // nothing in flight and no ray left to claim
        st_iters += 1;
        if (st_iters > (1ull << 20)) break;  // safety net: never spin forever on the GPU
        if (tid == 0)
        {
            uint32_t base = 0, groups = 0;
            for (int b = 0; b < kWfBuckets; ++b)
            {
                sh->bucket_base[b] = base;
                base += sh->bucket_count[b];
                groups += (sh->bucket_count[b] + 63u) / 64u;
            }
            sh->n_groups = groups;
            sh->group_head = 0;
            sh->n_march[next_list] = 0;  // last round's list is consumed; it now collects next round's marches
            sh->head_march = 0;
        }
        __syncthreads();
        {
            int k = 0;
            for (uint32_t slot = tid; slot < PS; slot += T, ++k)
                if (my_bucket[k] < kWfBuckets) P.event_list[sh->bucket_base[my_bucket[k]] + my_rank[k]] = static_cast<uint16_t>(slot);
        }
        __syncthreads();

        // ================= D: events, in 64-lane groups of one bucket =================
        const long long c1 = kStats ? clock64() : 0;
        for (;;)
        {
            uint32_t g = 0;
            if (lane == 0) g = atomicAdd(&sh->group_head, 1u);
            g = __shfl(g, 0);
            if (g >= sh->n_groups) break;
            // locate the group: bucket b, 64-entry window inside it
            uint32_t b = 0, first = 0;
            for (; b < kWfBuckets; ++b)
            {
                const uint32_t gb = (sh->bucket_count[b] + 63u) / 64u;
                if (g < first + gb) break;
                first += gb;
            }
            const uint32_t e = (g - first) * 64u + lane;
            const bool valid = e < sh->bucket_count[b];
            const long long cg0 = kStats ? clock64() : 0;
            if (kStats)
            {
                st_groups += 1;
                st_lane_events += __popcll(__ballot(valid));
            }
            uint32_t slot = 0;
            bool posted = false;  // this lane set up a new voxel march for its slot
            if (valid)
            {
                slot = P.event_list[sh->bucket_base[b] + e];
                posted = wf_event<CfgRuntime>(A, UpdOfArgs{A}, P, s_bits, b, slot, ray_cur + e, ray_cur + e < ray_end) == 1;  // (a march that ended at once waits in its event state for the next round)
            }
            const uint32_t at = wave_append(posted, &sh->n_march[cur_list], lane);
            if (posted) (P.march_list[0] + cur_list * PS)[at] = static_cast<uint16_t>(slot);
            if (kStats)
            {
                const long long dt = clock64() - cg0;
#pragma unroll
                for (int k = 0; k < kWfBuckets; ++k)
                    if (static_cast<uint32_t>(k) == b) cy_bucket[k] += dt, n_bucket[k] += 1;
            }
        }
        const long long c2 = kStats ? clock64() : 0;
        __syncthreads();

        // ================= B: march =================
        const long long c3 = kStats ? clock64() : 0;
        if (tid == 0)
        {
            sh->live = 0;
            const uint32_t taken = sh->cur + sh->bucket_count[kBucketRefill];
            sh->cur = taken;
            if (taken >= sh->end)  // range used up: claim the next chunk
            {
                const uint32_t c = atomicAdd(work_counter, 1u);
                sh->cur = c < n_chunks ? c * chunk : 0u;
                sh->end = c < n_chunks ? min(c * chunk + chunk, A.n_rays) : 0u;
            }
            for (int b = 0; b <= kWfBuckets; ++b) sh->bucket_count[b] = 0u;  // next read after this phase's barrier
        }
        {
            const uint32_t n_list = sh->n_march[cur_list];
            const uint16_t* list = P.march_list[0] + cur_list * PS;
            March m;
            m.ro = m.rd = m.dn = m.inv = m.cc = m.p = mk3(0, 0, 0);
            m.t = 0.0f, m.tl = inf, m.it = 0, m.lid = -1, m.cell = 0;
            uint32_t slot = 0, fl = 0;
            f3 hi_v = f3{A.scene.hi_f[0], A.scene.hi_f[1], A.scene.hi_f[2]};
            asm volatile("" : "+v"(hi_v.x), "+v"(hi_v.y), "+v"(hi_v.z));  // keep the clamp bound in VGPRs (see march_step_burst)
            bool have = false, exhausted = false;
            int tail = 0, trips = 0;
            for (;;)
            {
                const unsigned long long idle_mask = __ballot(!have);
                if (!exhausted && (__popcll(idle_mask) >= fetch_lanes))
                {
                    if (kStats) st_fetches += 1;
                    const uint32_t idx = wave_append(!have, &sh->head_march, lane);
                    const uint32_t top = __shfl(idx, 63 - __builtin_clzll(idle_mask)) + 1u;  // one past the wave's last claim
                    if (!have && idx < n_list)
                    {
                        slot = list[idx];
                        fl = P.flags[slot];
                        m.ro = ld3(P.ro, slot);
                        m.dn = ld3(P.dn, slot);
                        m.inv = f3{axis_inv(m.dn.x), axis_inv(m.dn.y), axis_inv(m.dn.z)};  // P5; recomputed, not stored
                        m.t = P.t[slot];
                        m.tl = P.tl[slot];
                        m.it = static_cast<int>((fl >> 4) & 255u);
                        m.cc = f3{m.dn.x >= 0.0f ? 1.0f : 0.0f, m.dn.y >= 0.0f ? 1.0f : 0.0f, m.dn.z >= 0.0f ? 1.0f : 0.0f};
                        m.p = ray_at(m.ro, m.dn, m.t);
                        have = true;
                    }
                    if (top >= n_list) exhausted = true;
                }
                const unsigned long long hb = __ballot(have);
                if (hb == 0ull)
                {
                    if (exhausted) break;
                    continue;
                }
                if (kStats)
                {
                    st_trips += 1;
                    st_lane_steps += __popcll(hb);
                }
                if (have)
                {
                    // kWfStepsPerTrip voxel steps per trip, fully unrolled and predicated: the loop's scalar
                    // bookkeeping (ballots, fetch / park decisions, ~28 SALU) is paid once per burst instead
                    // of once per step, and no branch breaks the burst's instruction stream; a march that
                    // ends early sits out the rest of the burst (measured: 1 step/trip 5.15 ms, 2: 4.73,
                    // 4: 4.37, 8: 4.08, 16: 4.00 on C3; a per-step wave-wide early exit costs more than it saves)
                    const int left = kMarchIters - m.it;  // >= 1: a march that used up its iterations is not in flight
                    bool occ = march_step_burst(m, A.scene, s_bits, hi_v);
                    bool fin = occ | (m.t >= m.tl) | (left <= 1);
#pragma unroll
                    for (int sub = 1; sub < kWfStepsPerTrip; ++sub)
                        if (!fin)
                        {
                            occ = march_step_burst(m, A.scene, s_bits, hi_v);
                            fin = occ | (m.t >= m.tl) | (left <= sub + 1);
                        }
                    m.it += kWfStepsPerTrip;  // only read again for a march that is still going (then all steps ran)
                    if (!fin && ((trips & 3) == 3)) fin = march_escaped(m, A.scene);
                    if (fin)
                    {
                        const uint32_t hf = !occ ? 0u : ((fl & kFlagFeeler) ? static_cast<uint32_t>(kFlagHit) : hit_flags<CfgRuntime>(A, UpdOfArgs{A}, static_cast<uint32_t>(hit_block_type(A.scene, A.scene_id, cell_id(m.p), m.cell)), m.p, false));
                        P.t[slot] = m.t;
                        P.flags[slot] = (fl & 0xf000u) | ((fl & kFlagFeeler) ? kSlotEvFeeler : kSlotEvPrimary) | (fl & kFlagFeeler) | hf;
                        have = false;
                    }
                }
                ++trips;
                // While new rays can still be claimed a straggler is parked after tail_steps trips (the
                // other waves have events to run).  Once the workgroup is draining its pool there is
                // nothing to overlap with: marches run to their end, so that a ray's remaining bounces
                // cost one round each instead of one round per parked trip.
                if (exhausted && ++tail >= (have_rays ? tail_steps : drain_tail))
                {
                    if (have)  // park the straggler: it resumes from (t, it) next round
                    {
                        P.t[slot] = m.t;
                        P.flags[slot] = (fl & ~0xff0u) | (static_cast<uint32_t>(m.it) << 4);
                    }
                    const uint32_t at = wave_append(have, &sh->n_march[next_list], lane);
                    if (have) (P.march_list[0] + next_list * PS)[at] = static_cast<uint16_t>(slot);
                    break;
                }
            }
        }
        const long long c4 = kStats ? clock64() : 0;
        __syncthreads();
        if (kStats)
        {
            const long long c5 = clock64();
            cy[0] += c1 - c0;  // C (sort) incl. its barriers
            cy[1] += c4 - c3;  // B work
            cy[2] += c5 - c4;  // B barrier wait
            cy[3] += 0;
            cy[4] += c2 - c1;  // D work
            cy[5] += c3 - c2;  // D barrier wait
        }
    }

    if (kStats && A.stats && lane == 0)
    {
        atomicAdd(&A.stats[0], st_trips);
        atomicAdd(&A.stats[1], st_lane_steps);
        atomicAdd(&A.stats[2], st_groups);
        atomicAdd(&A.stats[3], st_lane_events);
        atomicAdd(&A.stats[4], 1ull);
        if (wave == 0) atomicAdd(&A.stats[5], st_iters);
        atomicAdd(&A.stats[6], st_fetches);
        for (int k = 0; k < 6; ++k) atomicAdd(&A.stats[8 + k], static_cast<unsigned long long>(cy[k]));
        for (int k = 0; k < kWfBuckets; ++k)
        {
            atomicAdd(&A.stats[16 + k], static_cast<unsigned long long>(cy_bucket[k]));
            atomicAdd(&A.stats[24 + k], n_bucket[k]);
        }
    }
}

// ---- launchers (called from ddgi_engine.cpp) -----------------------------------------------------

// =================================================================================================
// k_probe_trace_aq — the same pool, marches and events WITHOUT rounds: no workgroup barrier after
// start-up.  Slots travel through LDS queues (rings of 16-bit slot ids; a slot is in at most one
// queue, every ring holds kAqCap >= pool entries, so a ring never overflows):
//     FQ free slots -> (refill event) -> MQ marches -> (march waves) -> EQ[bucket] -> (event waves) -> MQ | FQ
// The workgroup's first `march_waves` waves only march: a lane takes a march from MQ whenever it is
// idle, runs it in 16-step bursts to its end (nothing is ever parked) and pushes the slot to the event
// queue of its block-type bucket.  The other waves only run events: a full 64-lane group of one bucket
// if any queue holds 64, else a refill (64 new rays into free slots), else the fullest partial group.
// Every ray proceeds at its own pace: a round no longer lasts as long as its slowest member, and the
// pool drains without barrier-bound near-empty rounds.
// Ring protocol: producers reserve indices with one wave-aggregated atomic add on `tail` and then
// write the entries; consumers claim indices below `tail` with a compare-and-swap on `head`, wait for
// the entry to become valid (!= 0xffff) and invalidate it.
// =================================================================================================
#ifndef DDGI_AQ_T
#define DDGI_AQ_T 1024  // lanes per workgroup of k_probe_trace_aq
#endif
#ifndef DDGI_AQ_WGS
#define DDGI_AQ_WGS 1   // workgroups resident per CU (each with 160 KB / DDGI_AQ_WGS of LDS)
#endif
#ifndef DDGI_AQ_CAP
#define DDGI_AQ_CAP 2048
#endif
#ifndef DDGI_AQ_POOL_CT
#define DDGI_AQ_POOL_CT 1344
#endif
constexpr int kAqThreads = DDGI_AQ_T;
constexpr int kAqWgsPerCU = DDGI_AQ_WGS;
constexpr int kAqMinPool = kAqWgsPerCU == 1 ? 1024 : 320;
constexpr uint32_t kAqCap = DDGI_AQ_CAP;
constexpr uint32_t kAqCapFast = 1536;  // the fast build's compile-time pool (1280) + slack: a ring index is reused only after 256 later claims
#ifndef DDGI_AQ_FAST_STEPS
#define DDGI_AQ_FAST_STEPS 12
#endif
constexpr int kAqFastSteps = DDGI_AQ_FAST_STEPS;  // steps per burst of the fast build's march waves
#ifndef DDGI_AQ_SLEEP
#define DDGI_AQ_SLEEP 2  // an idle wave's nap between two looks at the queues, in units of 64 cycles
#endif
#ifndef DDGI_AQ_QUICK
#define DDGI_AQ_QUICK 1
#endif
#ifndef DDGI_EVENT_PRIO
#define DDGI_EVENT_PRIO 1
#endif
#ifndef DDGI_AQ_PARTIAL_MIN
#define DDGI_AQ_PARTIAL_MIN 1  // a partial event group is taken only with at least this many entries (while new rays can still come)
#endif
#ifndef DDGI_AQ_PICK
#define DDGI_AQ_PICK 1  // which full event queue an event wave takes: 0 the lowest bucket, 1 the highest (-0.4 %: kept), 2 round robin (+0.8 %)
#endif
#ifndef DDGI_AQ_SPEC_BURST
#define DDGI_AQ_SPEC_BURST 0  // march waves: the occupancy lookup off the dependent chain (experiment: +2.4 %, off)
#endif
#ifndef DDGI_MARCH_PRIO
#define DDGI_MARCH_PRIO 0  // s_setprio of the (exact) march waves; the event waves run at DDGI_EVENT_PRIO
#endif
// tools/valu_attribution.py: comment lines in the kernel's assembly that delimit the march waves' loop (-DDDGI_MARKS=1: analysis builds only)
#if defined(DDGI_MARKS) && DDGI_MARKS
#define DDGI_MARK(name) asm volatile("; DDGI_MARK " name)
#else
#define DDGI_MARK(name) \
    do                  \
    {                   \
    } while (0)
#endif
#ifndef DDGI_AQ_MIX_PICK
#define DDGI_AQ_MIX_PICK 0  // which update's rays an event group that holds rays of two updates keeps (k_probe_trace_aq, records instantiations)
#endif
#ifndef DDGI_AQ_ROLE_PERM
#define DDGI_AQ_ROLE_PERM 0xFEDCBA9876543210ull  // rank of every wave in the order in which waves become march waves (k_probe_trace_aq: `role`)
#endif
#ifndef DDGI_AQ_REQUEUE
#define DDGI_AQ_REQUEUE 0  // (experiment, round 5: measured slower, off) march waves: every burst starts from the queue, unfinished marches are queued
                           // again.  Lanes per burst 30.6 -> 40.2 of 64, bursts per ray 0.23 -> 0.18 — and C3 1.578 -> 1.647 ms at every wave split
                           // (profiles/r05_requeue_split_sweep.txt): a march wave's throughput is its bursts' latency, and a burst that fetches all
                           // 64 lanes and writes the unfinished ones back is longer than one that tops a few lanes up; the fill threshold never waits
                           // (1 / 32 / 48 / 60 entries: the same times — the queue is deep whenever the march side is the limit)
#endif
#ifndef DDGI_AQ_TL_AT_END
#define DDGI_AQ_TL_AT_END 1  // march waves: a march's t is compared with its light sphere's once per burst, not once per step (below)
#endif
#ifndef DDGI_AQ_MASKED
#define DDGI_AQ_MASKED 1  // march waves: the "march has ended" state of a burst as a mask in a VGPR (march_step_masked)
#endif
#ifndef DDGI_AQ_GIVE_BACK
#define DDGI_AQ_GIVE_BACK 0  // march waves: a wave left with fewer marches than this hands them back to the queue and naps (0: off)
#endif
#ifndef DDGI_AQ_RESTART
#define DDGI_AQ_RESTART 40   // ... until the queue holds this many marches, or DDGI_AQ_RESTART_WAITS naps of DDGI_AQ_RESTART_NAP x 64 cycles have passed
#endif
#ifndef DDGI_AQ_RESTART_WAITS
#define DDGI_AQ_RESTART_WAITS 8
#endif
#ifndef DDGI_AQ_RESTART_NAP
#define DDGI_AQ_RESTART_NAP 4
#endif
#ifndef DDGI_AQ_EARLY_OUT
#define DDGI_AQ_EARLY_OUT 0  // march waves: leave a burst at step 6 / 12 / 18 when every march of the wave has ended (0: off)
#endif
#ifndef DDGI_AQ_UNIFIED
#define DDGI_AQ_UNIFIED 0  // every wave of the workgroup serves every queue (below: "unified waves"); the exact march only
#endif
#ifndef DDGI_AQ_UNI_PICK
#define DDGI_AQ_UNI_PICK 2  // among full queues: 0 = new rays, marches, events; 1 = events, marches, new rays; 2 = marches, events, new rays
#endif
#ifndef DDGI_AQ_UNI_IDLE
#define DDGI_AQ_UNI_IDLE 8   // an idle wave's nap when no queue holds anything, in units of 64 cycles
#endif
#ifndef DDGI_AQ_UNI_MPRIO
#define DDGI_AQ_UNI_MPRIO DDGI_EVENT_PRIO  // s_setprio during a march burst (the events run at DDGI_EVENT_PRIO)
#endif
#ifndef DDGI_AQ_UNI_T2
#define DDGI_AQ_UNI_T2 32   // the smaller group a wave settles for first
#endif
#ifndef DDGI_AQ_UNI_W32
#define DDGI_AQ_UNI_W32 1   // naps before a wave settles for a group of DDGI_AQ_UNI_T2..63
#endif
#ifndef DDGI_AQ_UNI_W1
#define DDGI_AQ_UNI_W1 4    // naps before it settles for any group
#endif
#ifndef DDGI_AQ_FILL
#define DDGI_AQ_FILL 48  // ... and starts only when the queue holds this many marches (or after DDGI_AQ_FILL_WAITS naps of DDGI_AQ_FILL_NAP x 64 cycles)
#endif
#ifndef DDGI_AQ_FILL_WAITS
#define DDGI_AQ_FILL_WAITS 6
#endif
#ifndef DDGI_AQ_FILL_NAP
#define DDGI_AQ_FILL_NAP 4
#endif
[[maybe_unused]] constexpr uint32_t kAqFill = DDGI_AQ_FILL;
[[maybe_unused]] constexpr int kAqFillWaits = DDGI_AQ_FILL_WAITS;
#ifndef DDGI_AQ_TRIP_ARGS
#define DDGI_AQ_TRIP_ARGS 1  // the event waves read the kernel's arguments afresh in every trip (args_of_this_trip)
#endif
#ifndef DDGI_AQ_THIN_WAITS
#define DDGI_AQ_THIN_WAITS 4
#endif
#ifndef DDGI_AQ_THIN_NAP
#define DDGI_AQ_THIN_NAP 4
#endif
#ifndef DDGI_AQ_REFILL_FIRST
#define DDGI_AQ_REFILL_FIRST 0
#endif
#ifndef DDGI_AQ_THIN
#define DDGI_AQ_THIN 0
#endif
constexpr int kAqThinTrip = DDGI_AQ_THIN;    // a march wave with fewer lanes in flight (and nothing queued) yields for a moment
#ifndef DDGI_AQ_PARTIAL_BELOW
#define DDGI_AQ_PARTIAL_BELOW 64
#endif
constexpr uint32_t kAqPartialBelow = DDGI_AQ_PARTIAL_BELOW;  // an event wave takes a partial group only while fewer marches than this are queued
constexpr int kAqEventQueues = 7;  // buckets 0..6 (kBucketRefill is served from FQ)

// STAGING RINGS (round 6, experiment: -DDDGI_AQ_STAGING=1; records instantiations with frames in flight only).  An event group is shaded with ONE update's
// record; where the pool holds rays of two updates (two thirds of a slab's update) a group drawn from a bucket's ring is mixed, the others go round again.
// In front of every bucket's ring stand two small rings, one per PARITY of the update a ray belongs to: a push goes there while there is room (a counting
// semaphore per ring: a reservation that fails must not consume a ring index), else into the bucket's ring as before; event waves serve the staging rings first.
// A group from a staging ring is pure unless three updates meet (k and k + 2 share a parity): the mixed-group handling stays for every queue.
// MEASURED (profiles/r06_staging_ab.txt, same box, bit-exact: 69 tests): C3 DDGI at eight frames in flight 1.656 -> 1.960 ms per update, a G = 8 slab 0.328 -> 0.369 - 0.410 ms,
// unchanged where no launch goes on with a later update.  Every event entry pays three more trips to the LDS (the ray's update, the semaphore, the ring's place) on the waves that
// bound the kernel, and what does not fit a 128-entry ring waits in the bucket's own ring behind it.  Serving a bucket's own ring first was no better (1.93 ms).  OFF.
#ifndef DDGI_AQ_STAGING
#define DDGI_AQ_STAGING 0
#endif
#if DDGI_AQ_STAGING && (DDGI_AQ_UNIFIED || DDGI_AQ_REFILL_FIRST || DDGI_AQ_REQUEUE || (DDGI_AQ_PICK != 1) || !DDGI_AQ_QUICK)
#error "DDGI_AQ_STAGING is written for the default event-wave loop (DDGI_AQ_PICK 1, DDGI_AQ_QUICK 1, no unified waves / refill-first / requeue experiments)"
#endif
// VARIANT 3 (-DDDGI_AQ_STAGING=3 -DDDGI_AQ_POOL_CT=1216): no staging in front of anything — the event rings themselves are one per (bucket, parity), each deep enough
// for the whole pool (kAqEvCap3 entries: pool + 256), paid for with 128 of the pool's slots; where no launch goes on with a later update the first seven serve as the
// buckets' rings.  One ring per entry, no semaphore: an entry costs what it costs today plus the read of its ray's update.
// MEASURED (profiles/r06_staging_ab.txt, bit-exact): a G = 8 DDGI slab 0.325 - 0.340 -> 0.329 ms (nothing), the whole grid at eight frames in flight 1.654 -> 1.819 ms, REF 1.487 -> 1.537:
// rays split over twice as many rings fill a 64-lane group half as often — the lanes a mixed group loses become groups that wait or go partial.  OFF.
constexpr uint32_t kAqSq = 128;                                  // entries per staging ring (a power of two)
constexpr uint32_t kAqEvCap3 = 1472;                             // variant 3: entries per event ring
constexpr int kAqStagingQueues = 2 * kAqEventQueues;             // (bucket, parity)
constexpr int kAqAllQueues = DDGI_AQ_STAGING == 3 ? kAqStagingQueues : (DDGI_AQ_STAGING ? kAqEventQueues + kAqStagingQueues : kAqEventQueues);
constexpr int kAqCtrlDwords = DDGI_AQ_STAGING ? 64 : 32;

struct AqShared  // control block at the start of dynamic LDS (kAqCtrlDwords dwords)
{
    uint32_t mq_head, mq_tail;
    uint32_t fq_head, fq_tail;
    uint32_t eq_head[kAqAllQueues], eq_tail[kAqAllQueues];  // [0, kAqEventQueues): the buckets' rings; then the staging rings, index kAqEventQueues + 2 bucket + parity
    uint32_t live;     // rays in flight (claimed and not yet finished)
    uint32_t no_more;  // the launch's ray counter is used up
    uint32_t abort;    // safety net tripped: every wave leaves
    uint32_t cur_seq;  // the update whose rays the workgroup claims (frames in flight: the launch's own, then the ones chained to it)
#if DDGI_AQ_STAGING == 3
    uint32_t pad[64 - 4 - 2 * kAqAllQueues - 4];
#elif DDGI_AQ_STAGING
    uint32_t sq_room[kAqStagingQueues];  // free entries of a staging ring (taken by a producer before it reserves an index, given back by the consumer)
#else
    uint32_t pad[32 - 4 - 2 * kAqEventQueues - 4];
#endif
};
static_assert(sizeof(AqShared) == kAqCtrlDwords * 4, "the control block's size");

// The queue kernel's arguments as an event group reads them.  The kernel is one persistent loop around a very large body; its
// arguments are invariant loads, which the optimizer hoists out of that loop — a hundred scalars live across everything, most
// of them spilled into VGPR lanes and read back with a v_readlane (a VALU slot, on the unit that bounds the kernel) at every
// use.  Read through a pointer the optimizer cannot see through, once per trip of the loop, they are loaded where a trip
// uses them (s_load: the scalar unit's time) and are dead at its end.  kFirstArg: TraceArgs is the kernel's first argument.
DDGI_D const TraceArgs& args_of_this_trip()
{
    typedef const __attribute__((address_space(4))) TraceArgs* KernArg;
    KernArg kp = (KernArg)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kp));
    return *(const TraceArgs*)kp;
}

DDGI_D uint32_t aq_load(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// One lane: claims up to `want` entries of a ring; returns the count, base = first claimed index.
DDGI_D uint32_t aq_claim(uint32_t* head, const uint32_t* tail, uint32_t want, uint32_t& base)
{
    for (int tries = 0; tries < 1024; ++tries)
    {
        const uint32_t h = aq_load(head), t = aq_load(tail);
        const uint32_t avail = t - h;
        const uint32_t k = avail < want ? avail : want;
        base = h;
        if (k == 0u || static_cast<int32_t>(avail) < 0) return 0u;
        if (atomicCAS(head, h, h + k) == h) return k;
    }
    return 0u;
}

// Reads (and invalidates) ring entry idx; waits until its producer has written it.  kCap: the ring's capacity (a power of
// two folds the modulo into a mask; the fast build's rings hold kAqCapFast entries).
template <uint32_t kCap>
DDGI_D uint32_t aq_take(uint16_t* ring, uint32_t idx, uint32_t* abort)
{
    // (relaxed atomics, not `volatile`: address-space inference leaves volatile accesses generic — flat_load / flat_store with
    // system scope and a wait for EVERY outstanding global store of the wave in front of each ring entry; these are ds_read_u16 / ds_write_b16)
    uint16_t* p = ring + (idx % kCap);
    uint32_t v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    for (int spins = 0; v == 0xffffu; ++spins)
    {
        if (spins > (1 << 22))
        {
            *abort = 1u;
            return 0u;
        }
        v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __hip_atomic_store(p, static_cast<uint16_t>(0xffffu), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return v;
}

// Every lane with pred appends `value` to a ring.
template <uint32_t kCap>
DDGI_D void aq_push(uint16_t* ring, uint32_t* tail, bool pred, uint32_t value, int lane)
{
    const uint32_t at = wave_append(pred, tail, lane);
    if (pred) ring[at % kCap] = static_cast<uint16_t>(value);
}

template <bool kRecords>
struct UpdSource;
template <>
struct UpdSource<true>
{
    typedef UpdOfRing Type;
    static DDGI_D UpdOfRing make(const TraceArgs&, const uint32_t* record) { return UpdOfRing(record); }
};
template <>
struct UpdSource<false>
{
    typedef UpdOfArgs Type;
    static DDGI_D UpdOfArgs make(const TraceArgs& A, const uint32_t*) { return UpdOfArgs{A}; }
};

// kPool > 0: the pool size is a compile-time constant, so every pool array is the LDS base plus a constant
// offset (folded into the ds instructions: no address arithmetic, one SGPR instead of eleven).
template <bool kStats, int kPool, class Cfg>
__global__ __launch_bounds__(kAqThreads, kAqThreads * kAqWgsPerCU / 256) void k_probe_trace_aq(const TraceArgs A, const int pool_size, const int march_waves, const AqChain C, uint32_t* __restrict__ status)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t wf_lds[];
    constexpr int T = kAqThreads;
    const int tid = threadIdx.x;
    // Every launch on a handle has a sequence number; launch s claims its rays from counters[s % 16].  The counter launch s + 8 will
    // use is zeroed here instead of by a fill kernel in front of every launch (5 us of kernel and a dependency of its own per
    // update): its last user, launch s - 8, ended before this one started (stream order), and its next user starts after this
    // one has ended — a launch claims rays of at most kAqChainMax - 1 = 7 launches after itself (below).
    if (blockIdx.x == 0 && tid == 0) C.counters[(C.seq + kAqChainMax) & (kAqCounters - 1u)] = 0u;
    // FRAMES IN FLIGHT.  When the rays of this launch's update are used up, the workgroups would drain — all of them at once, for
    // the life of their last rays (a ray's 8 bounces are a dependent chain: 0.25 ms of thinning pool per launch).  If the host has
    // already submitted the NEXT update, and that update is the same work into the next texture pair (C.pub, written by
    // ddgi_probe_update before it launches that update's own kernel), the workgroups go on with ITS rays instead, up to
    // C.chain_max updates ahead.  That update's own launch then finds its counter used up and leaves at once (here).
    if (tid == 0) wf_lds[0] = __hip_atomic_load(C.counters + (C.seq & (kAqCounters - 1u)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (wf_lds[0] >= A.n_rays) return;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    // Which waves march: the `march_waves` waves of lowest RANK (DDGI_AQ_ROLE_PERM: nibble w = rank of wave w; the identity: the first
    // `march_waves` waves).  A workgroup's waves go round the CU's four SIMDs (wave w on SIMD w & 3), so the ranks decide how march and
    // event waves are mixed on each SIMD: the identity at 7 gives 2 + 2, 2 + 2, 2 + 2, 1 + 3 (march + event waves).
    const int role = DDGI_AQ_ROLE_PERM == 0xFEDCBA9876543210ull ? wave : static_cast<int>((static_cast<unsigned long long>(DDGI_AQ_ROLE_PERM) >> (4 * wave)) & 15ull);
    // THE PER-UPDATE RECORDS (ddgi_types.h: UpdK).  What differs from one update to the next — lights, DDGI mode's rotation and key, ray
    // and record buffers, feeler classes — is not read from the kernel's arguments but from a ring of records: a ray of update
    // C.seq + t (t in dst[31:29]) is shaded with record t.  The host has written the launch's own record into pinned memory before
    // the launch; every workgroup copies it into the device ring here (the same bytes from every workgroup), and the record of an
    // update it goes on with when it gets there (event waves, below).  The events then read it with scalar loads (UpdOfRing).
    auto upd_record = [&](uint32_t t) -> const uint32_t* {
        return C.upd_dev + ((C.seq + t) & (kAqCounters - 1u)) * kUpdWords;
    };
    auto copy_record = [&](uint32_t seq_of) {  // one wave: pinned host [seq_of] -> the device ring
        const uint32_t* src = C.upd_host + (seq_of & (kAqPubRing - 1u)) * kUpdWords;
        uint32_t* dst = C.upd_dev + (seq_of & (kAqCounters - 1u)) * kUpdWords;
        const uint32_t v0 = __hip_atomic_load(src + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const uint32_t v1 = __hip_atomic_load(src + 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        dst[lane] = v0, dst[64 + lane] = v1;
        // In this XCD's L2 before anything of this workgroup reads it (the waves of a workgroup read through the scalar cache of
        // their own CU, which has never held these lines in this launch): the stores' completion is all that takes — a
        // workgroup-scope release, i.e. a wait.  (An agent-scope fence here writes back and invalidates the XCD's L2, once per
        // workgroup and update: the noise tables and block types every event reads went with it — 3.5 % of a C3 update.)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    };
    if (Cfg::kRecords && wave == 0) copy_record(C.seq);
    typedef UpdSource<Cfg::kRecords> Upd;
    const uint32_t PS = kPool > 0 ? static_cast<uint32_t>(kPool) : static_cast<uint32_t>(pool_size);
    const int fetch_lanes = A.wf_fetch > 0 ? A.wf_fetch : kWfFetchLanes;

    // ---- carve LDS: control block | occupancy bitmap (fast build: the skip field) | pool arrays | rings ----
    constexpr uint32_t kCap = (Cfg::kFast && kPool > 0) ? kAqCapFast : kAqCap;  // ring capacity > pool
    const int scene_words = Cfg::kFast ? A.scene.nwords_skip : A.scene.nwords;
    const uint32_t* __restrict__ scene_src = Cfg::kFast ? A.scene.skip : A.scene.bits;
    AqShared* sh = reinterpret_cast<AqShared*>(wf_lds);
    uint32_t* s_bits = wf_lds + kAqCtrlDwords;
    uint32_t* cursor = s_bits + ((scene_words + 3) & ~3);
    WfPool P;
    auto takef = [&]() { float* p = reinterpret_cast<float*>(cursor); cursor += PS; return p; };
    auto takeu = [&]() { uint32_t* p = cursor; cursor += PS; return p; };
    for (int a = 0; a < 3; ++a) P.ro[a] = takef();
    for (int a = 0; a < 3; ++a) P.dn[a] = takef();
    P.t = takef();
    P.tl = takef();
    P.flags = takeu();
    for (int a = 0; a < 3; ++a) P.col[a] = takef();
    P.rd[0] = takef();
    P.rd[1] = Cfg::kFast ? P.rd[0] : takef();  // (the fast build keeps |rd| in rd[0] and never touches the others)
    P.rd[2] = Cfg::kFast ? P.rd[0] : takef();
    P.rng = takeu(), P.cnt = takeu(), P.dst = takeu();
    P.cold = static_cast<WfColdGlobal*>(A.wf_cold) + static_cast<size_t>(blockIdx.x) * PS;
    P.dirbuf = Cfg::nl(A) > 1 ? A.wf_dir + static_cast<size_t>(blockIdx.x) * PS : nullptr;
    P.march_list[0] = P.march_list[1] = P.event_list = nullptr;
    uint16_t* ring_mq = reinterpret_cast<uint16_t*>(cursor);
    uint16_t* ring_fq = ring_mq + kCap;
    uint16_t* ring_eq = ring_fq + kCap;  // kAqEventQueues rings
    [[maybe_unused]] uint16_t* ring_sq = ring_eq + kCap * kAqEventQueues;  // DDGI_AQ_STAGING: kAqStagingQueues rings of kAqSq entries
    // (staging is used by the records instantiations while a launch may go on with later updates; the rings exist in every instantiation of a staging build)
    [[maybe_unused]] const bool staging = DDGI_AQ_STAGING && Cfg::kRecords && C.chain_max != 0u;

    for (int i = tid; i < scene_words; i += T) s_bits[i] = scene_src[i];
    for (uint32_t i = tid; i < PS; i += T) P.flags[i] = kSlotEmpty;
    constexpr uint32_t kEvCap = DDGI_AQ_STAGING == 3 ? kAqEvCap3 : kCap;  // entries per event ring
    for (uint32_t i = tid; i < (DDGI_AQ_STAGING == 3 ? kCap * 2 + kAqEvCap3 * kAqStagingQueues : kCap * (2 + kAqEventQueues) + (DDGI_AQ_STAGING ? kAqSq * kAqStagingQueues : 0u)); i += T) ring_mq[i] = 0xffffu;
    __syncthreads();
    for (uint32_t i = tid; i < PS; i += T) ring_fq[i] = static_cast<uint16_t>(i);  // every slot starts free
    if (tid < kAqCtrlDwords) wf_lds[tid] = 0u;
    __syncthreads();
    if (tid == 0) sh->fq_tail = PS, sh->cur_seq = C.seq;
#if DDGI_AQ_STAGING && DDGI_AQ_STAGING != 3
    if (tid < kAqStagingQueues) sh->sq_room[tid] = kAqSq;
#endif
    __syncthreads();

    // One lane: an event-queue entry for bucket `bk`.  Staging build: into the staging ring of the ray's update's parity while that ring has room.
    auto push_event = [&](uint32_t bk, uint32_t slot_) {
#if DDGI_AQ_STAGING == 3
        {
            const uint32_t r = staging ? bk * 2u + ((P.dst[slot_] >> kDstPairShift) & 1u) : bk;
            const uint32_t at = atomicAdd(&sh->eq_tail[r], 1u);
            (ring_eq + r * kEvCap)[at % kEvCap] = static_cast<uint16_t>(slot_);
            return;
        }
#elif DDGI_AQ_STAGING
        if (staging)
        {
            const uint32_t si = bk * 2u + ((P.dst[slot_] >> kDstPairShift) & 1u);
            if (static_cast<int32_t>(atomicSub(&sh->sq_room[si], 1u)) > 0)
            {
                const uint32_t at = atomicAdd(&sh->eq_tail[kAqEventQueues + si], 1u);
                uint16_t* const q = ring_sq + si * kAqSq + (at & (kAqSq - 1u));
                // (a 128-entry ring wraps onto entries a slow wave has claimed and not yet read: wait for the place to be empty.  The claimer only has to
                //  read it — it waits for nobody)
                for (int spins = 0; __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0xffffu; ++spins)
                    if (spins > (1 << 22))
                    {
                        sh->abort = 1u;
                        break;
                    }
                __hip_atomic_store(q, static_cast<uint16_t>(slot_), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                return;
            }
            atomicAdd(&sh->sq_room[si], 1u);
        }
#endif
        const uint32_t at = atomicAdd(&sh->eq_tail[bk], 1u);
        (ring_eq + bk * kEvCap)[at % kEvCap] = static_cast<uint16_t>(slot_);
    };

    // (the march waves' only use of a record is the routing HINT of a hit — dead_feeler_hint, never a result: the launch's own)
    const typename Upd::Type U0 = Upd::make(A, upd_record(0u));
    const float inf = __builtin_inff();
    unsigned int guard = 0;  // safety net: consecutive polls without work (about 1 s of them trips it); never spin forever on the GPU
    unsigned long long st_a = 0, st_b = 0;  // utilisation counters: trips / groups, and the lanes that had work in them
    [[maybe_unused]] unsigned long long st_ma = 0, st_mb = 0;  // unified waves: the march bursts' (st_a / st_b are the event groups')
    unsigned long long st_useful = 0;       // counters build: lane-steps of the march bursts that moved a march (the others stood still: frozen or no march)
    LaneProbe probe;                        // (counters build only)
#ifdef DDGI_LAP
    if (kStats)
    {
        probe.lap = reinterpret_cast<uint32_t*>(ring_eq + kCap * kAqEventQueues) + wave * 32;
        if (lane < 32) probe.lap[lane] = lane == 0 ? 15u : (lane == 1 ? static_cast<uint32_t>(__builtin_readcyclecounter()) : 0u);
    }
#endif
    unsigned long long st_q[8] = {};        // counters build: [0] samples [1..3] sum of MQ / FQ / EQ depths at an event wave's poll, [4] idle polls,
                                            // [5] march bursts, [6] fetches that found MQ short of the idle lanes, [7] lanes in flight at burst start

    if (Cfg::kFast && role < march_waves)
    {
        // ================= march waves, fast build =================
        // The same loop as below around fast_march_step: a lane's march skips through voxels the skip field calls empty.
        // Marches are a third as many steps long, so a burst is kAqFastSteps steps.
        FastMarch m;
        m.ro = m.dn = m.ainv = m.nsgn = m.c1 = m.p = mk3(0, 0, 0);
        m.t = 0.0f, m.tl = inf, m.code = 1.0f, m.cell = 0;
        uint32_t slot = 0, fl = 0;
        f3 hi_v = f3{A.scene.hi_f[0], A.scene.hi_f[1], A.scene.hi_f[2]};
        asm volatile("" : "+v"(hi_v.x), "+v"(hi_v.y), "+v"(hi_v.z));
        bool have = false;
        int trips = 0, thin_waits = 0;
        for (;;)
        {
            if (++guard > (1u << 23)) sh->abort = 1u;
            const unsigned long long idle_mask = __ballot(!have);
            const int n_idle = __popcll(idle_mask);
            if (n_idle >= fetch_lanes)
            {
                uint32_t base = 0, k = 0;
                if (lane == 0) k = aq_claim(&sh->mq_head, &sh->mq_tail, static_cast<uint32_t>(n_idle), base);
                k = lane_bcast(k, 0), base = lane_bcast(base, 0);
                const uint32_t rank = static_cast<uint32_t>(__popcll(idle_mask & ((1ull << lane) - 1ull)));
                if (!have && rank < k)
                {
                    slot = aq_take<kCap>(ring_mq, base + rank, &sh->abort);
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    fl = P.flags[slot];
                    fast_march_begin(m, ld3(P.ro, slot), ld3(P.dn, slot), P.t[slot], P.tl[slot]);
                    have = true;
                }
            }
            if (__ballot(have) == 0ull)
            {
                if ((aq_load(&sh->no_more) != 0u && aq_load(&sh->live) == 0u) || aq_load(&sh->abort) != 0u) break;
                __builtin_amdgcn_s_sleep(DDGI_AQ_SLEEP);
                continue;
            }
            if (__popcll(__ballot(have)) < kAqThinTrip && thin_waits < 4 && aq_load(&sh->no_more) == 0u)
            {
                ++thin_waits;
                __builtin_amdgcn_s_sleep(4);
                continue;
            }
            thin_waits = 0;
            guard = 0;
            bool finished = false;
            uint32_t bucket = 0;
            if (have)
            {
                bool fin = (fast_march_step(m, A.scene, s_bits, hi_v) == 0u) | (m.t >= m.tl);
#pragma unroll
                for (int sub = 1; sub < kAqFastSteps; ++sub)
                    if (!fin) fin = (fast_march_step(m, A.scene, s_bits, hi_v) == 0u) | (m.t >= m.tl);
                bool occ = m.code == 0.0f;
                // grid_march's 125 iterations, as planes crossed: a voxel reached after more of them is never looked at, and a
                // march that has used them up is over
                const float planes = fast_march_planes(m);
                if (planes >= static_cast<float>(kMarchIters))
                {
                    fin = true;
                    if (planes > static_cast<float>(kMarchIters)) occ = false;
                }
                if (!fin && (trips & 1)) fin = march_escaped(m, A.scene);
                if (fin)
                {
                    const uint32_t hf = !occ ? 0u : ((fl & kFlagFeeler) ? static_cast<uint32_t>(kFlagHit) : hit_flags<Cfg>(A, U0, static_cast<uint32_t>(hit_block_type(A.scene, A.scene_id, cell_id(m.p), m.cell)), m.p, false));
                    P.t[slot] = m.t;
                    P.flags[slot] = (fl & 0xf000u) | ((fl & kFlagFeeler) ? kSlotEvFeeler : kSlotEvPrimary) | (fl & kFlagFeeler) | hf;
                    const bool block_wins = occ && (m.t < m.tl);
                    bucket = (fl & kFlagFeeler) ? kBucketFeeler : (block_wins ? primary_bucket(hf) : kBucketNoBlock);
                    have = false;
                    finished = true;
                }
            }
            ++trips;
            if (__ballot(finished) != 0ull)
            {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                if (finished) push_event(bucket, slot);
            }
        }
    }
#if DDGI_AQ_REQUEUE
    else if (role < march_waves)
    {
        // ================= march waves (round 5): every burst starts from the queue =================
        // Per C3 update the march waves used to issue 0.7 G of the kernel's 1.1 G VALU wave-instructions at 26 of 64 lanes: a wave
        // kept its unfinished marches from burst to burst and topped its idle lanes up with whatever the queue held at that moment —
        // and the queue held little, because seven waves were emptying it as fast as the events filled it.  A march in the middle of
        // its steps is nothing but (t, iterations) next to the slot's origin and direction — what an event leaves for a march it has
        // taken the first steps of — so a wave that has finished a burst writes that back and queues the unfinished slots again, like
        // the finished ones go to their event queues.  Then every burst starts with an empty wave, and a wave starts one only when
        // the queue can FILL it (kAqFill entries, or what there is after a bounded wait): the marches in flight share few, full
        // bursts, the other march waves sleep — which is issue time for the event waves on their SIMDs.
        if (DDGI_MARCH_PRIO > 0) __builtin_amdgcn_s_setprio(DDGI_MARCH_PRIO);
        March m;
        m.ro = m.rd = m.dn = m.inv = m.cc = m.p = mk3(0, 0, 0);  // (a lane without a march takes zero-length steps from this state: every index it computes is inside the bitmap)
        m.t = 0.0f, m.tl = inf, m.it = 0, m.lid = -1, m.cell = 0;
        f3 hi_v = f3{A.scene.hi_f[0], A.scene.hi_f[1], A.scene.hi_f[2]};
        asm volatile("" : "+v"(hi_v.x), "+v"(hi_v.y), "+v"(hi_v.z));
        int waited = 0, bursts = 0;
        for (;;)
        {
            if (++guard > (1u << 23)) sh->abort = 1u;
            uint32_t avail = 0;
            if (lane == 0)
            {
                avail = aq_load(&sh->mq_tail) - aq_load(&sh->mq_head);
                if (avail > kCap) avail = 0;  // a claim in flight can make tail - head wrap for an instant
            }
            avail = lane_bcast(avail, 0);
            const bool draining = aq_load(&sh->no_more) != 0u;
            if (avail == 0u || (avail < kAqFill && waited < kAqFillWaits && !draining))
            {
                if ((avail == 0u && draining && aq_load(&sh->live) == 0u) || aq_load(&sh->abort) != 0u) break;
                if (avail != 0u) ++waited;
                if (avail != 0u) __builtin_amdgcn_s_sleep(DDGI_AQ_FILL_NAP);
                else __builtin_amdgcn_s_sleep(DDGI_AQ_SLEEP);
                continue;
            }
            uint32_t base = 0, k = 0;
            if (lane == 0) k = aq_claim(&sh->mq_head, &sh->mq_tail, 64u, base);
            k = lane_bcast(k, 0), base = lane_bcast(base, 0);
            if (k == 0u) continue;  // (another wave was quicker)
            waited = 0;
            guard = 0;
            const bool have = static_cast<uint32_t>(lane) < k;
            uint32_t slot = 0, fl = 0;
            if (have)
            {
                slot = aq_take<kCap>(ring_mq, base + lane, &sh->abort);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                fl = P.flags[slot];
                m.ro = ld3(P.ro, slot);
                m.dn = ld3(P.dn, slot);
                m.inv = f3{axis_inv(m.dn.x), axis_inv(m.dn.y), axis_inv(m.dn.z)};  // P5; recomputed, not stored
                m.t = P.t[slot];
                m.tl = P.tl[slot];
                m.it = static_cast<int>((fl >> 4) & 255u);
                m.cc = f3{m.dn.x >= 0.0f ? 1.0f : 0.0f, m.dn.y >= 0.0f ? 1.0f : 0.0f, m.dn.z >= 0.0f ? 1.0f : 0.0f};
                m.p = ray_at(m.ro, m.dn, m.t);
            }
            if (kStats) st_a += 1, st_b += k, st_q[5] += 1, st_q[7] += k;
            // The burst: kAqStepsPerTrip unrolled steps without exec-mask predication — a march that has ended, or a lane without
            // one, takes steps of length 0 (march_step_frozen); whether the cell reached is occupied is read once after the burst
            // (m.cell); the test against grid_march's iteration limit is in the steps only when a lane can reach it in this burst.
            const bool near_limit = __ballot(have && kMarchIters - m.it < kAqStepsPerTrip) != 0ull;
            bool fin = !have;
            if (near_limit)
            {
                fin = fin || (kMarchIters - m.it <= 0);
#pragma unroll
                for (int sub = 0; sub < kAqStepsPerTrip; ++sub)
                {
                    if (kStats) st_useful += static_cast<unsigned long long>(__popcll(__ballot(!fin)));
                    fin = fin | march_step_frozen(m, A.scene, s_bits, hi_v, fin) | (m.t >= m.tl) | (kMarchIters - m.it <= sub + 1);
                }
            }
            else
            {
#pragma unroll
                for (int sub = 0; sub < kAqStepsPerTrip; ++sub)
                {
                    if (kStats) st_useful += static_cast<unsigned long long>(__popcll(__ballot(!fin)));
                    fin = fin | march_step_frozen(m, A.scene, s_bits, hi_v, fin) | (m.t >= m.tl);
                }
            }
            bool finished = false, again = false;
            uint32_t bucket = 0;
            if (have)
            {
                const uint32_t* __restrict__ bits_base = s_bits - (A.scene.bias32 >> 5);
                DDGI_EXP_RECLAMP(m, A.scene, hi_v);
                const bool occ = __builtin_amdgcn_ubfe(bits_base[m.cell >> 5], static_cast<uint32_t>(m.cell), 1u) != 0u;  // (march_step_frozen's own test)
                m.it += kAqStepsPerTrip;
                bool f = fin;
                // (a march that has left the box for good can only miss: looked for in every fourth burst of the wave)
                if (!f && (bursts & 3) == 3) f = march_escaped(m, A.scene);
                P.t[slot] = m.t;
                if (f)
                {
                    const uint32_t hf = !occ ? 0u : ((fl & kFlagFeeler) ? static_cast<uint32_t>(kFlagHit) : hit_flags<Cfg>(A, U0, static_cast<uint32_t>(hit_block_type(A.scene, A.scene_id, cell_id(m.p), m.cell)), m.p, false));
                    P.flags[slot] = (fl & 0xf000u) | ((fl & kFlagFeeler) ? kSlotEvFeeler : kSlotEvPrimary) | (fl & kFlagFeeler) | hf;
                    const bool block_wins = occ && (m.t < m.tl);
                    bucket = (fl & kFlagFeeler) ? kBucketFeeler : (block_wins ? primary_bucket(hf) : kBucketNoBlock);
                    finished = true;
                }
                else
                {
                    P.flags[slot] = (fl & ~0xff0u) | (static_cast<uint32_t>(m.it) << 4);  // goes on at (t, iterations), like a march an event has set up
                    again = true;
                }
            }
            ++bursts;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (__ballot(finished) != 0ull)
            {
                // a handful of lanes per queue: one LDS atomic per lane is cheaper than six wave-aggregated appends
                if (finished)
                {
                    const uint32_t at = atomicAdd(&sh->eq_tail[bucket], 1u);
                    (ring_eq + bucket * kCap)[at % kCap] = static_cast<uint16_t>(slot);
                }
            }
            aq_push<kCap>(ring_mq, &sh->mq_tail, again, slot, lane);
        }
    }
#endif
    else if (!DDGI_AQ_REQUEUE && role < march_waves)
    {
        // ================= march waves =================
        // A lane holds kLaneMarches marches at a time and steps them in turn inside one burst.  CDNA4's SIMDs are 32 lanes wide:
        // a wave64 VALU instruction issues over 2 cycles, a wave can issue one every 4, and a DEPENDENT one only every ~7.6
        // (tools/microbench/valu_latency.hip) — and a voxel step is a dependent chain from end to end, with a trip to LDS in it.
        // A march wave is therefore bound by its own latency, not by the SIMD's issue rate (the kernel keeps the VALU busy about
        // half the cycles): two independent chains interleaved in one wave march nearly twice as many rays per wave, so that
        // fewer of the workgroup's 16 waves have to march and more of them run events — which is what limits the kernel.
        constexpr int kM = kLaneMarches;
        if (DDGI_MARCH_PRIO > 0) __builtin_amdgcn_s_setprio(DDGI_MARCH_PRIO);
        March m[kM];
        uint32_t slot[kM], fl[kM];
        bool have[kM];
#pragma unroll
        for (int q = 0; q < kM; ++q)
        {
            // (a lane without a march takes zero-length steps from this state: every index it computes is inside the bitmap)
            m[q].ro = m[q].rd = m[q].dn = m[q].inv = m[q].cc = m[q].p = mk3(0, 0, 0);
            m[q].t = 0.0f, m[q].tl = inf, m[q].it = 0, m[q].lid = -1, m[q].cell = 0;
            slot[q] = 0, fl[q] = 0, have[q] = false;
        }
        f3 hi_v = f3{A.scene.hi_f[0], A.scene.hi_f[1], A.scene.hi_f[2]};
        asm volatile("" : "+v"(hi_v.x), "+v"(hi_v.y), "+v"(hi_v.z));
        int trips = 0, thin_waits = 0;
        [[maybe_unused]] bool forced = false;
        for (;;)
        {
            DDGI_MARK("march_trip_begin");
            if (++guard > (1u << 23)) sh->abort = 1u;
            unsigned long long idle_mask[kM];
            int n_idle = 0;
#pragma unroll
            for (int q = 0; q < kM; ++q) idle_mask[q] = __ballot(!have[q]), n_idle += __popcll(idle_mask[q]);
            if (n_idle >= fetch_lanes)
            {
                uint32_t base = 0, k = 0;
                if (lane == 0) k = aq_claim(&sh->mq_head, &sh->mq_tail, static_cast<uint32_t>(n_idle), base);
                k = lane_bcast(k, 0), base = lane_bcast(base, 0);
                if (kStats && static_cast<int>(k) < n_idle) st_q[6] += 1;
                uint32_t before = 0;  // idle places of the marches in front of march q
#pragma unroll
                for (int q = 0; q < kM; ++q)
                {
                    const uint32_t rank = before + static_cast<uint32_t>(__popcll(idle_mask[q] & ((1ull << lane) - 1ull)));
                    before += static_cast<uint32_t>(__popcll(idle_mask[q]));
                    if (!have[q] && rank < k)
                    {
                        slot[q] = aq_take<kCap>(ring_mq, base + rank, &sh->abort);
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                        fl[q] = P.flags[slot[q]];
                        m[q].ro = ld3(P.ro, slot[q]);
                        m[q].dn = ld3(P.dn, slot[q]);
                        m[q].inv = f3{axis_inv(m[q].dn.x), axis_inv(m[q].dn.y), axis_inv(m[q].dn.z)};  // P5; recomputed, not stored
                        m[q].t = P.t[slot[q]];
                        m[q].tl = P.tl[slot[q]];
                        m[q].it = static_cast<int>((fl[q] >> 4) & 255u);
                        m[q].cc = f3{m[q].dn.x >= 0.0f ? 1.0f : 0.0f, m[q].dn.y >= 0.0f ? 1.0f : 0.0f, m[q].dn.z >= 0.0f ? 1.0f : 0.0f};
                        m[q].p = ray_at(m[q].ro, m[q].dn, m[q].t);
                        have[q] = true;
                    }
                }
            }
            bool any = false;
#pragma unroll
            for (int q = 0; q < kM; ++q) any = any || have[q];
            if (__ballot(any) == 0ull)
            {
                if ((aq_load(&sh->no_more) != 0u && aq_load(&sh->live) == 0u) || aq_load(&sh->abort) != 0u) break;
                __builtin_amdgcn_s_sleep(DDGI_AQ_SLEEP);
                continue;
            }
            if (DDGI_AQ_GIVE_BACK > 0 && kM == 1)
            {
                // A THIN WAVE GIVES ITS MARCHES BACK.  A burst costs the same instructions for 10 marches as for 64, and the seven march
                // waves share ~200 marches in flight: each tops its idle lanes up from a queue the others keep short, and all of them
                // run half empty.  A wave left with few marches writes them back as (t, iterations) — what an event leaves for a march
                // it has taken the first steps of — queues them again and naps; the other waves take them into THEIR idle lanes (a fetch
                // they run anyway).  It starts again when the queue can fill a burst, or after a bounded wait (then with whatever
                // there is: progress is never held up).
                const int n_have = __popcll(__ballot(have[0]));
                if (n_have < DDGI_AQ_GIVE_BACK && !forced && aq_load(&sh->no_more) == 0u)
                {
                    if (have[0])
                    {
                        P.t[slot[0]] = m[0].t;
                        P.flags[slot[0]] = (fl[0] & ~0xff0u) | (static_cast<uint32_t>(m[0].it) << 4);
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    aq_push<kCap>(ring_mq, &sh->mq_tail, have[0], slot[0], lane);
                    have[0] = false;
                    for (int w = 0; w < DDGI_AQ_RESTART_WAITS; ++w)
                    {
                        __builtin_amdgcn_s_sleep(DDGI_AQ_RESTART_NAP);
                        const uint32_t avail = aq_load(&sh->mq_tail) - aq_load(&sh->mq_head);
                        if (avail >= static_cast<uint32_t>(DDGI_AQ_RESTART) && avail <= kCap) break;
                    }
                    forced = true;  // (the next trip runs with what it gets)
                    continue;
                }
                forced = false;
            }
            if (kAqThinTrip > 0)
            {
                // a thin wave waits a moment for more marches instead of spending a whole burst's instructions on a few lanes
                if (__popcll(__ballot(any)) < kAqThinTrip && thin_waits < DDGI_AQ_THIN_WAITS && aq_load(&sh->no_more) == 0u)
                {
                    ++thin_waits;
                    __builtin_amdgcn_s_sleep(DDGI_AQ_THIN_NAP);
                    continue;
                }
                thin_waits = 0;
            }
            guard = 0;
            if (kStats)
            {
                unsigned long long busy = 0;
#pragma unroll
                for (int q = 0; q < kM; ++q) busy += static_cast<unsigned long long>(__popcll(__ballot(have[q])));
                st_a += 1, st_b += busy, st_q[5] += 1, st_q[7] += busy;
            }
            // The burst.  Per step the wave pays 26 VALU and a dozen scalar instructions of lane-mask bookkeeping, and the kernel's
            // time follows the total: (1) whether the cell reached is occupied is NOT carried from step to step as a lane mask
            // — it is the bit of the last cell reached (m.cell), read once after the burst; (2) the test against grid_march's
            // iteration limit is dropped from the steps when no lane of the wave can reach the limit within this burst (all but
            // the last burst of the few marches longer than 101 steps); (3) no exec-mask predication inside the burst: a march that
            // has ended — or a place without a march — takes steps of length 0 (march_step_frozen).
            DDGI_MARK("march_burst_begin");
            bool near = false;
#pragma unroll
            for (int q = 0; q < kM; ++q) near = near || (have[q] && kMarchIters - m[q].it < kAqStepsPerTrip);
            const bool near_limit = __ballot(near) != 0ull;
            bool finished[kM];
            uint32_t bucket[kM];
#pragma unroll
            for (int q = 0; q < kM; ++q) finished[q] = false, bucket[q] = 0;
            if (kM == 1 && DDGI_AQ_SPEC_BURST && any)
            {
                // THE SPECULATIVE BURST (experiment; docs/LAB_NOTES.md "Round 4").  In march_step_frozen the bit a step looks up
                // decides whether the NEXT step moves: the trip to the LDS and the index arithmetic in front of it sit on the march's
                // dependent chain — 18 instructions and a memory access per step, of which 7 move the ray.  Here every lane keeps
                // stepping whatever it finds (positions past a march's end are looked up clamped into the box like any other), t
                // after the first step at which the march ended is recorded on the side, position and cell are recomputed from it
                // after the burst — bit for bit what march_step_frozen leaves behind.
                March& M = m[0];
                bool seen = !have[0] || (near_limit && kMarchIters - M.it <= 0);
                float rec_t = M.t;
                if (near_limit)
                {
#pragma unroll
                    for (int sub = 0; sub < kAqStepsPerTrip; ++sub)
                    {
                        const bool occ_s = march_step_burst(M, A.scene, s_bits, hi_v);
                        rec_t = seen ? rec_t : M.t;
                        seen = seen | occ_s | (M.t >= M.tl) | (kMarchIters - M.it <= sub + 1);
                    }
                }
                else
                {
#pragma unroll
                    for (int sub = 0; sub < kAqStepsPerTrip; ++sub)
                    {
                        const bool occ_s = march_step_burst(M, A.scene, s_bits, hi_v);
                        rec_t = seen ? rec_t : M.t;
                        seen = seen | occ_s | (M.t >= M.tl);
                    }
                }
                M.t = rec_t;
                M.p = ray_at(M.ro, M.dn, M.t);
                {
                    const float kx = __builtin_amdgcn_fmed3f(ceilf(M.p.x), A.scene.lo_f[0], hi_v.x);
                    const float ky = __builtin_amdgcn_fmed3f(ceilf(M.p.y), A.scene.lo_f[1], hi_v.y);
                    const float kz = __builtin_amdgcn_fmed3f(ceilf(M.p.z), A.scene.lo_f[2], hi_v.z);
                    M.cell = static_cast<int>(fmaf(kz, A.scene.nxy_f, fmaf(ky, A.scene.nx_f, kx)));
                }
                const uint32_t* __restrict__ bits_base = s_bits - (A.scene.bias32 >> 5);
                if (have[0])
                {
                    const bool occ = __builtin_amdgcn_ubfe(bits_base[M.cell >> 5], static_cast<uint32_t>(M.cell), 1u) != 0u;
                    M.it += kAqStepsPerTrip;
                    bool f = seen;
                    if (!f && ((trips & 3) == 3)) f = march_escaped(M, A.scene);
                    if (f)
                    {
                        const uint32_t hf = !occ ? 0u : ((fl[0] & kFlagFeeler) ? static_cast<uint32_t>(kFlagHit) : hit_flags<Cfg>(A, U0, static_cast<uint32_t>(hit_block_type(A.scene, A.scene_id, cell_id(M.p), M.cell)), M.p, false));
                        P.t[slot[0]] = M.t;
                        P.flags[slot[0]] = (fl[0] & 0xf000u) | ((fl[0] & kFlagFeeler) ? kSlotEvFeeler : kSlotEvPrimary) | (fl[0] & kFlagFeeler) | hf;
                        const bool block_wins = occ && (M.t < M.tl);
                        bucket[0] = (fl[0] & kFlagFeeler) ? kBucketFeeler : (block_wins ? primary_bucket(hf) : kBucketNoBlock);
                        have[0] = false;
                        finished[0] = true;
                    }
                }
            }
            else if (any)
            {
                bool fin[kM];
                if (near_limit)
                {
#pragma unroll
                    for (int q = 0; q < kM; ++q) fin[q] = !have[q] || (kMarchIters - m[q].it <= 0);
#pragma unroll
                    for (int sub = 0; sub < kAqStepsPerTrip; ++sub)
#pragma unroll
                        for (int q = 0; q < kM; ++q) fin[q] = fin[q] | march_step_frozen(m[q], A.scene, s_bits, hi_v, fin[q]) | (!DDGI_AQ_TL_AT_END && m[q].t >= m[q].tl) | (kMarchIters - m[q].it <= sub + 1);
                }
                else
                {
#pragma unroll
                    for (int q = 0; q < kM; ++q) fin[q] = !have[q];
                    if constexpr (kM == 2)
                    {
#pragma unroll
                        for (int sub = 0; sub < kAqStepsPerTrip; ++sub)
                        {
                            bool occ2[2];
                            march_step_frozen2(m, A.scene, s_bits, hi_v, fin, occ2);
#pragma unroll
                            for (int q = 0; q < 2; ++q) fin[q] = fin[q] | occ2[q] | (m[q].t >= m[q].tl);
                        }
                    }
                    else if (DDGI_AQ_MASKED && DDGI_AQ_TL_AT_END && kM == 1 && !kStats)
                    {
                        // (a lane whose march ended in an earlier burst and that got no new one: its state is that march's — where it
                        // ended in a voxel, it stands; else it steps on through whatever the clamped lookups give it, unread)
                        uint32_t occ = 0u;
#pragma unroll
                        for (int sub = 0; sub < kAqStepsPerTrip; ++sub) march_step_masked(m[0], A.scene, s_bits, hi_v, occ);
                        fin[0] = fin[0] | (occ != 0u);
                    }
                    else
                    {
#pragma unroll
                        for (int sub = 0; sub < kAqStepsPerTrip; ++sub)
                        {
                            if (DDGI_AQ_EARLY_OUT > 0 && kM == 1 && sub > 0 && sub % (DDGI_AQ_EARLY_OUT > 0 ? DDGI_AQ_EARLY_OUT : 1) == 0 && __ballot(!fin[0]) == 0ull) break;  // (every march of the wave has ended)
#pragma unroll
                            for (int q = 0; q < kM; ++q)
                            {
                                if (kStats) st_useful += static_cast<unsigned long long>(__popcll(__ballot(!fin[q])));  // lanes that take this step for a march
                                fin[q] = fin[q] | march_step_frozen(m[q], A.scene, s_bits, hi_v, fin[q]) | (!DDGI_AQ_TL_AT_END && m[q].t >= m[q].tl);
                            }
                        }
                    }
                }
                // THE LIGHT SPHERE, ONCE PER BURST.  grid_march (intersection.glsl:1051-1100) knows nothing of the light spheres: it
                // steps until a voxel is occupied, and intersect_scene keeps the block only if its t is below the nearest sphere's
                // (:1283-1291).  Ending a march where t passes the sphere's is this kernel's shortcut — every later block loses — and t
                // only grows, so the test need not sit in every step (a compare and a lane-mask OR in each of 24): a march that has
                // passed its sphere steps on to the end of the burst (the lanes are there anyway), any block it still finds has
                // t >= tl and loses as before, and the march ends here.  What an event reads of a march that the light wins is
                // tl and the light, never t or the hit flags (wf_event: block_wins).
                if (DDGI_AQ_TL_AT_END)
                {
#pragma unroll
                    for (int q = 0; q < kM; ++q) fin[q] = fin[q] | (m[q].t >= m[q].tl);
                }
                DDGI_MARK("march_burst_end");
                const uint32_t* __restrict__ bits_base = s_bits - (A.scene.bias32 >> 5);
#pragma unroll
                for (int q = 0; q < kM; ++q)
                {
                    if (!have[q]) continue;
                    DDGI_EXP_RECLAMP(m[q], A.scene, hi_v);
                    const bool occ = __builtin_amdgcn_ubfe(bits_base[m[q].cell >> 5], static_cast<uint32_t>(m[q].cell), 1u) != 0u;  // (march_step_frozen's own test)
                    m[q].it += kAqStepsPerTrip;
                    bool f = fin[q];
                    if (!f && ((trips & 3) == 3)) f = march_escaped(m[q], A.scene);
                    if (f)
                    {
                        const uint32_t hf = !occ ? 0u : ((fl[q] & kFlagFeeler) ? static_cast<uint32_t>(kFlagHit) : hit_flags<Cfg>(A, U0, static_cast<uint32_t>(hit_block_type(A.scene, A.scene_id, cell_id(m[q].p), m[q].cell)), m[q].p, false));
                        P.t[slot[q]] = m[q].t;
                        P.flags[slot[q]] = (fl[q] & 0xf000u) | ((fl[q] & kFlagFeeler) ? kSlotEvFeeler : kSlotEvPrimary) | (fl[q] & kFlagFeeler) | hf;
                        const bool block_wins = occ && (m[q].t < m[q].tl);
                        bucket[q] = (fl[q] & kFlagFeeler) ? kBucketFeeler : (block_wins ? primary_bucket(hf) : kBucketNoBlock);
                        have[q] = false;
                        finished[q] = true;
                    }
                }
            }
            ++trips;
            bool any_finished = false;
#pragma unroll
            for (int q = 0; q < kM; ++q) any_finished = any_finished || finished[q];
            if (__ballot(any_finished) != 0ull)
            {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                // a handful of lanes finish per trip, spread over six queues: one LDS atomic per lane is
                // cheaper than six wave-aggregated appends
#pragma unroll
                for (int q = 0; q < kM; ++q)
                    if (finished[q]) push_event(bucket[q], slot[q]);
            }
            DDGI_MARK("march_trip_end");
        }
    }
    else
    {
        // ================= event waves =================
        // The event waves are what limits the kernel: they issue ahead of the march waves, which fill the
        // slots that are left — with that the split settles at 7 march waves instead of 5 and a thin march wave need not step
        // aside any more (C3 1.909 -> 1.878 ms).  Raising the MARCH waves' priority instead changes nothing (round 2).
        __builtin_amdgcn_s_setprio(DDGI_EVENT_PRIO);
        unsigned guard_rot = static_cast<unsigned>(wave);
        (void)guard_rot;
        [[maybe_unused]] constexpr bool kUnified = DDGI_AQ_UNIFIED && !Cfg::kFast;
        [[maybe_unused]] int uni_waited = 0;
        for (;;)
        {
            if (++guard > (1u << 23)) sh->abort = 1u;
#if DDGI_AQ_TRIP_ARGS
            const TraceArgs& A = args_of_this_trip();  // (shadows the kernel's parameter for the trip)
#endif
            // Which group?  Lanes 0..5 look at one event queue each (polling is paid in VALU issue slots that
            // the marching waves of the same SIMD want, so it is kept to a handful of instructions).
            uint32_t b = 0, base = 0, k = 0;
            [[maybe_unused]] uint32_t qi = 0;  // the queue the group comes from (== b unless it is a staging ring)
            uint32_t avail = 0;
            uint32_t head_seen = 0;
            // (staging build, records + frames in flight: lanes kAqEventQueues .. also look at the staging rings — the same two loads, more lanes)
            [[maybe_unused]] const int n_queues = staging ? kAqAllQueues : kAqEventQueues;  // (variant 3: 14 rings by (bucket, parity), or the first 7 by bucket)
#if DDGI_AQ_STAGING
            if (lane < n_queues) head_seen = aq_load(&sh->eq_head[lane]), avail = aq_load(&sh->eq_tail[lane]) - head_seen;
#else
            if (lane < kAqEventQueues) head_seen = aq_load(&sh->eq_head[lane]), avail = aq_load(&sh->eq_tail[lane]) - head_seen;
#endif
#if DDGI_AQ_UNIFIED
            // UNIFIED WAVES.  With waves set aside for marching, a march wave steps whatever the queue holds at that moment (30 of 64 lanes
            // per burst, 0.30 of a burst's lane-steps useful: 54 % of the kernel's VALU instructions) while the slots pile up in front of
            // the event waves.  Here every wave serves every queue — lane 7 looks at the march queue, lane 8 at the free slots — and takes
            // a FULL group of whatever there is: 64 marches from the queue for one burst (the unfinished ones are queued again as (t,
            // iterations), like a march an event has taken the first steps of), 64 events of a bucket, or 64 free slots for new rays;
            // a wave settles for less only after it has waited.  No wave split to tune, and a burst's instructions carry 64 marches.
            constexpr int kLaneMq = kAqEventQueues, kLaneFq = kAqEventQueues + 1;
            if (kUnified && lane == kLaneMq) head_seen = aq_load(&sh->mq_head), avail = aq_load(&sh->mq_tail) - head_seen;
            if (kUnified && lane == kLaneFq) head_seen = aq_load(&sh->fq_head), avail = aq_load(&sh->fq_tail) - head_seen;
#endif
#if DDGI_AQ_REFILL_FIRST
            if (lane == kAqEventQueues) avail = aq_load(&sh->fq_tail) - aq_load(&sh->fq_head);
#endif
            if (avail > kCap) avail = 0;  // a claim in flight can make tail - head wrap for an instant
#if DDGI_AQ_STAGING
            const unsigned long long full = __ballot(avail >= 64u && lane < n_queues);
#else
            const unsigned long long full = __ballot(avail >= 64u && lane < kAqEventQueues);
#endif
            const bool no_more = aq_load(&sh->no_more) != 0u;
#if DDGI_AQ_UNIFIED
            bool uni_chosen = false;
            if (kUnified)
            {
                if (kStats) st_q[6] += 1;  // (unified waves: looks at the queues)
                const bool counts = lane <= kLaneFq && !(no_more && lane == kLaneFq);  // (no new rays: the free queue is not work)
                const unsigned long long m64 = __ballot(counts && avail >= 64u);
                int pick = -1;
                if (m64 != 0ull)
                {
                    const unsigned long long ev = m64 & ((1ull << kAqEventQueues) - 1ull);
#if DDGI_AQ_UNI_PICK == 0
                    pick = 63 - __clzll(static_cast<long long>(m64));
#elif DDGI_AQ_UNI_PICK == 1
                    pick = ev ? 63 - __clzll(static_cast<long long>(ev)) : ((m64 >> kLaneMq) & 1ull ? kLaneMq : kLaneFq);
#else
                    pick = (m64 >> kLaneMq) & 1ull ? kLaneMq : (ev ? 63 - __clzll(static_cast<long long>(ev)) : kLaneFq);
#endif
                }
                else
                {
                    const unsigned long long m32 = __ballot(counts && avail >= static_cast<uint32_t>(DDGI_AQ_UNI_T2)), m1 = __ballot(counts && avail >= 1u);
                    if (m1 == 0ull)
                    {
                        if ((no_more && aq_load(&sh->live) == 0u) || aq_load(&sh->abort) != 0u) break;
                        if (kStats) st_q[4] += 1;
                        __builtin_amdgcn_s_sleep(DDGI_AQ_UNI_IDLE);
                        continue;
                    }
                    const unsigned long long m = (m32 != 0ull && (no_more || uni_waited >= DDGI_AQ_UNI_W32)) ? m32 : ((no_more || uni_waited >= DDGI_AQ_UNI_W1) ? m1 : 0ull);
                    if (m == 0ull)
                    {
                        ++uni_waited;
                        __builtin_amdgcn_s_sleep(DDGI_AQ_SLEEP);
                        continue;
                    }
                    pick = 63 - __clzll(static_cast<long long>(m));
                }
                const uint32_t h0 = lane_bcast(head_seen, pick), n0 = lane_bcast(avail, pick);
                uint32_t* const q_head = pick == kLaneMq ? &sh->mq_head : (pick == kLaneFq ? &sh->fq_head : &sh->eq_head[pick]);
                const uint32_t* const q_tail = pick == kLaneMq ? &sh->mq_tail : (pick == kLaneFq ? &sh->fq_tail : &sh->eq_tail[pick]);
                if (lane == 0)
                {
                    // (the queue held n0 entries above the head this wave has just read: claimed with that value — one trip to the LDS —
                    // unless another wave got there first)
                    const uint32_t want = n0 < 64u ? n0 : 64u;
                    if (atomicCAS(q_head, h0, h0 + want) == h0) k = want, base = h0;
                    else k = aq_claim(q_head, q_tail, 64u, base);
                }
                k = lane_bcast(k, 0), base = lane_bcast(base, 0);
                if (k == 0u) continue;  // (another wave was quicker)
                uni_waited = 0;
                guard = 0;
                if (pick == kLaneMq)
                {
                    // ---- one burst for k marches from the queue ----
                    March m;
                    m.ro = m.rd = m.dn = m.inv = m.cc = m.p = mk3(0, 0, 0);  // (a lane without a march takes zero-length steps from this state: every index it computes is inside the bitmap)
                    m.t = 0.0f, m.tl = inf, m.it = 0, m.lid = -1, m.cell = 0;
                    f3 hi_v = f3{A.scene.hi_f[0], A.scene.hi_f[1], A.scene.hi_f[2]};
                    asm volatile("" : "+v"(hi_v.x), "+v"(hi_v.y), "+v"(hi_v.z));
                    const bool have = static_cast<uint32_t>(lane) < k;
                    uint32_t slot = 0, fl = 0;
                    if (have)
                    {
                        slot = aq_take<kCap>(ring_mq, base + lane, &sh->abort);
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                        fl = P.flags[slot];
                        m.ro = ld3(P.ro, slot);
                        m.dn = ld3(P.dn, slot);
                        m.inv = f3{axis_inv(m.dn.x), axis_inv(m.dn.y), axis_inv(m.dn.z)};  // P5; recomputed, not stored
                        m.t = P.t[slot];
                        m.tl = P.tl[slot];
                        m.it = static_cast<int>((fl >> 4) & 255u);
                        m.cc = f3{m.dn.x >= 0.0f ? 1.0f : 0.0f, m.dn.y >= 0.0f ? 1.0f : 0.0f, m.dn.z >= 0.0f ? 1.0f : 0.0f};
                        m.p = ray_at(m.ro, m.dn, m.t);
                    }
                    if (kStats) st_ma += 1, st_mb += k, st_q[5] += 1, st_q[7] += k;
                    if (DDGI_AQ_UNI_MPRIO != DDGI_EVENT_PRIO) __builtin_amdgcn_s_setprio(DDGI_AQ_UNI_MPRIO);
                    DDGI_MARK("march_burst_begin");
                    const bool near_limit = __ballot(have && kMarchIters - m.it < kAqStepsPerTrip) != 0ull;
                    bool fin = !have;
                    if (near_limit)
                    {
                        fin = fin || (kMarchIters - m.it <= 0);
#pragma unroll
                        for (int sub = 0; sub < kAqStepsPerTrip; ++sub)
                        {
                            if (kStats) st_useful += static_cast<unsigned long long>(__popcll(__ballot(!fin)));
                            fin = fin | march_step_frozen(m, A.scene, s_bits, hi_v, fin) | (m.t >= m.tl) | (kMarchIters - m.it <= sub + 1);
                        }
                    }
                    else
                    {
#pragma unroll
                        for (int sub = 0; sub < kAqStepsPerTrip; ++sub)
                        {
                            if (kStats) st_useful += static_cast<unsigned long long>(__popcll(__ballot(!fin)));
                            fin = fin | march_step_frozen(m, A.scene, s_bits, hi_v, fin) | (m.t >= m.tl);
                        }
                    }
                    DDGI_MARK("march_burst_end");
                    if (DDGI_AQ_UNI_MPRIO != DDGI_EVENT_PRIO) __builtin_amdgcn_s_setprio(DDGI_EVENT_PRIO);
                    bool finished = false, again = false;
                    uint32_t bucket = 0;
                    if (have)
                    {
                        const uint32_t* __restrict__ bits_base = s_bits - (A.scene.bias32 >> 5);
                        DDGI_EXP_RECLAMP(m, A.scene, hi_v);
                        const bool occ = __builtin_amdgcn_ubfe(bits_base[m.cell >> 5], static_cast<uint32_t>(m.cell), 1u) != 0u;  // (march_step_frozen's own test)
                        m.it += kAqStepsPerTrip;
                        bool f = fin;
                        // (a march that has left the box for good can only miss)
                        if (!f) f = march_escaped(m, A.scene);
                        P.t[slot] = m.t;
                        if (f)
                        {
                            const uint32_t hf = !occ ? 0u : ((fl & kFlagFeeler) ? static_cast<uint32_t>(kFlagHit) : hit_flags<Cfg>(A, U0, static_cast<uint32_t>(hit_block_type(A.scene, A.scene_id, cell_id(m.p), m.cell)), m.p, false));
                            P.flags[slot] = (fl & 0xf000u) | ((fl & kFlagFeeler) ? kSlotEvFeeler : kSlotEvPrimary) | (fl & kFlagFeeler) | hf;
                            const bool block_wins = occ && (m.t < m.tl);
                            bucket = (fl & kFlagFeeler) ? kBucketFeeler : (block_wins ? primary_bucket(hf) : kBucketNoBlock);
                            finished = true;
                        }
                        else
                        {
                            P.flags[slot] = (fl & ~0xff0u) | (static_cast<uint32_t>(m.it) << 4);  // goes on at (t, iterations), like a march an event has set up
                            again = true;
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    if (finished)  // a handful of lanes per queue: one LDS atomic per lane is cheaper than six wave-aggregated appends
                    {
                        const uint32_t at = atomicAdd(&sh->eq_tail[bucket], 1u);
                        (ring_eq + bucket * kCap)[at % kCap] = static_cast<uint16_t>(slot);
                    }
                    aq_push<kCap>(ring_mq, &sh->mq_tail, again, slot, lane);
                    DDGI_MARK("march_trip_end");
                    continue;
                }
                b = pick == kLaneFq ? kBucketRefill : static_cast<uint32_t>(pick);
                uni_chosen = true;
            }
#endif
#if DDGI_AQ_REFILL_FIRST
            // new rays BEFORE full event groups whenever 64 slots are free: the pool stays full, the march waves' lanes with it
            const bool refill_first = !no_more && lane_bcast(avail, kAqEventQueues) >= 64u;
#else
            constexpr bool refill_first = false;
#endif

#if DDGI_AQ_UNIFIED
            if (uni_chosen)
            {
            }
            else
#endif
            if (full != 0ull && !refill_first)
            {
                // 1) a full group
#if DDGI_AQ_PICK == 1   // the highest full bucket first: dead hits, feelers, misses (short events that free slots or unblock a ray) before the shading buckets
                b = static_cast<uint32_t>(63 - __clzll(static_cast<long long>(full)));  // (staging build: a QUEUE index — the staging rings lie above the buckets' and are served first)
                qi = b;
#if DDGI_AQ_STAGING == 3
                b = staging ? qi >> 1 : qi;
#else
                if (DDGI_AQ_STAGING && qi >= static_cast<uint32_t>(kAqEventQueues)) b = (qi - static_cast<uint32_t>(kAqEventQueues)) >> 1;
                else b = qi;
#endif
#elif DDGI_AQ_PICK == 3  // (experiment) a fixed order of the buckets, DDGI_AQ_PICK_ORDER
                {
                    constexpr int order[kAqEventQueues] = {DDGI_AQ_PICK_ORDER};
                    b = 0;
#pragma unroll
                    for (int o = kAqEventQueues - 1; o >= 0; --o)
                        if ((full >> order[o]) & 1ull) b = static_cast<uint32_t>(order[o]);
                }
#elif DDGI_AQ_PICK == 2  // (experiment) rotate with the wave's trip count
                {
                    const unsigned rot = (guard_rot++) % kAqEventQueues;
                    const unsigned long long f2 = ((full >> rot) | (full << (kAqEventQueues - rot))) & ((1ull << kAqEventQueues) - 1ull);
                    b = (static_cast<uint32_t>(__ffsll(static_cast<long long>(f2)) - 1) + rot) % kAqEventQueues;
                }
#else
                b = static_cast<uint32_t>(__ffsll(static_cast<long long>(full)) - 1);
#endif
#if DDGI_AQ_QUICK
                // the queue held 64 entries above the head this wave has just read: claim them with that value (one trip to the LDS
                // instead of aq_claim's two); only if another wave got there first is the queue looked at again
                const uint32_t h0 = lane_bcast(head_seen, static_cast<int>(qi));
                if (lane == 0)
                {
                    if (atomicCAS(&sh->eq_head[qi], h0, h0 + 64u) == h0) k = 64u, base = h0;
                    else k = aq_claim(&sh->eq_head[qi], &sh->eq_tail[qi], 64u, base);
                }
#else
                if (lane == 0) k = aq_claim(&sh->eq_head[b], &sh->eq_tail[b], 64u, base);
#endif
            }
            else
            {
                // 2) new rays into free slots
                if (!no_more)
                {
                    b = kBucketRefill;
                    if (lane == 0) k = aq_claim(&sh->fq_head, &sh->fq_tail, 64u, base);
                }
                k = lane_bcast(k, 0);
                // 3) the fullest partial group — unless the march side still has work queued: then a full
                //    group is worth waiting for
                if (k == 0u && (no_more || aq_load(&sh->mq_tail) - aq_load(&sh->mq_head) < kAqPartialBelow))
                {
                    uint32_t best = 0, best_n = 0;
#pragma unroll
                    for (int bb = 0; bb < kAqAllQueues; ++bb)
                    {
                        const uint32_t n = (!DDGI_AQ_STAGING || bb < n_queues) ? lane_bcast(avail, bb) : 0u;
                        if (n > best_n) best = static_cast<uint32_t>(bb), best_n = n;
                    }
                    if (best_n >= (no_more ? 1u : static_cast<uint32_t>(DDGI_AQ_PARTIAL_MIN)))
                    {
                        b = qi = best;
#if DDGI_AQ_STAGING == 3
                        b = staging ? qi >> 1 : qi;
#else
                        if (DDGI_AQ_STAGING && qi >= static_cast<uint32_t>(kAqEventQueues)) b = (qi - static_cast<uint32_t>(kAqEventQueues)) >> 1;
#endif
                        if (lane == 0) k = aq_claim(&sh->eq_head[best], &sh->eq_tail[best], 64u, base);
                    }
                }
            }
            k = lane_bcast(k, 0), base = lane_bcast(base, 0);
            if (k == 0u)
            {
                if ((no_more && aq_load(&sh->live) == 0u) || aq_load(&sh->abort) != 0u) break;
                if (kStats) st_q[4] += 1;
                __builtin_amdgcn_s_sleep(DDGI_AQ_SLEEP);
                continue;
            }
            const bool valid = static_cast<uint32_t>(lane) < k;
            guard = 0;
            if (kStats) st_a += 1, st_b += k;
            if (kStats)  // queue depths, sampled when a group starts (so weighted by work, not by idle polls)
            {
                uint32_t eq_sum = lane < kAqEventQueues ? avail : 0u;
                for (int mm = 4; mm >= 1; mm >>= 1) eq_sum += __shfl_xor(eq_sum, mm);
                const uint32_t mqd = aq_load(&sh->mq_tail) - aq_load(&sh->mq_head), fqd = aq_load(&sh->fq_tail) - aq_load(&sh->fq_head);
                st_q[0] += 1, st_q[1] += mqd <= kCap ? mqd : 0u, st_q[2] += fqd <= kCap ? fqd : 0u, st_q[3] += eq_sum;
            }
            uint32_t slot = 0;
            bool posted = false, freed = false;
            int ev_bucket = -1;
            if (b == kBucketRefill)
            {
                // k free slots: claim k rays of the update the workgroup is at for them
                uint32_t rbase = 0, cs = 0;
                if (lane == 0)
                {
                    atomicAdd(&sh->live, k);  // counted before they exist, so that `live` never reads 0 early
                    cs = aq_load(&sh->cur_seq);
                    rbase = atomicAdd(C.counters + (cs & (kAqCounters - 1u)), k);
                }
                rbase = lane_bcast(rbase, 0);
                if (C.chain_max) cs = lane_bcast(cs, 0);
                const uint32_t r = rbase + static_cast<uint32_t>(lane);
                const bool r_valid = valid && r < A.n_rays;
                if (valid) slot = aq_take<kCap>(ring_fq, base + lane, &sh->abort);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                if (r_valid)
                {
                    const uint32_t tag = C.chain_max ? cs - C.seq : 0u;  // which update after the launch's own (its texture pair, its record)
                    const typename Upd::Type U = Upd::make(A, upd_record(tag));
                    const int rc = wf_event<Cfg>(A, U, P, s_bits, kBucketRefill, slot, r, true, kStats ? &probe : nullptr, tag << kDstPairShift);
                    posted = rc == 1;
                    if (rc >= 2) ev_bucket = rc - 2;
                }
                freed = valid && !r_valid;  // more slots than rays left: hand them back
                const uint32_t n_back = static_cast<uint32_t>(__popcll(__ballot(freed)));
                uint32_t used_up = 0;  // 1: this update's rays are used up; 2: and the next update is published as its continuation
                if (lane == 0)
                {
                    if (n_back) atomicSub(&sh->live, n_back);
                    if (rbase + k >= A.n_rays)
                    {
                        // this update's rays are used up.  Has the host submitted the next one as a continuation of this one (its
                        // sequence number + 1 in its slot of the pinned host ring)?  Then go on with its rays; else the launch ends.
                        used_up = 1u;
                        if (cs - C.seq < C.chain_max && __hip_atomic_load(C.pub + ((cs + 1u) & (kAqPubRing - 1u)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == cs + 2u) used_up = 2u;  // (the record's loads below are issued after this value has arrived, and bypass the caches like it)
                    }
                }
                if (Cfg::kRecords && C.chain_max && lane_bcast(used_up, 0) == 2u) copy_record(cs + 1u);  // that update's record, before any of its rays starts here
                if (lane == 0 && used_up)
                {
                    if (used_up == 2u && atomicCAS(&sh->cur_seq, cs, cs + 1u) == cs) atomicAdd(C.continued, 1u);  // (another wave may have got there first)
                    else if (aq_load(&sh->cur_seq) == cs) __hip_atomic_store(&sh->no_more, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
            else
            {
                bool mine = valid;
                uint32_t t = 0u;  // which update after the launch's own the group's rays belong to
                if (valid)
                {
#if DDGI_AQ_STAGING == 3
                    slot = aq_take<kEvCap>(ring_eq + qi * kEvCap, base + lane, &sh->abort);
#elif DDGI_AQ_STAGING
                    if (qi >= static_cast<uint32_t>(kAqEventQueues))
                    {
                        slot = aq_take<kAqSq>(ring_sq + (qi - static_cast<uint32_t>(kAqEventQueues)) * kAqSq, base + lane, &sh->abort);
                        if (lane == 0) atomicAdd(&sh->sq_room[qi - static_cast<uint32_t>(kAqEventQueues)], k);  // (behind this wave's reads of the entries: LDS operations of a wave are in order)
                    }
                    else
                        slot = aq_take<kCap>(ring_eq + b * kCap, base + lane, &sh->abort);
#else
                    slot = aq_take<kCap>(ring_eq + b * kCap, base + lane, &sh->abort);
#endif
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                }
                if (Cfg::kRecords && C.chain_max)
                {
                    // A group is shaded with ONE record, wave-uniform (scalar loads, like kernel arguments): that of its first ray's
                    // update.  Where one update ends and the next begins a few groups hold rays of both: the others go back to the
                    // end of the queue they came from (their state is untouched) and are shaded with a later group.
                    const uint32_t tagv = valid ? P.dst[slot] >> kDstPairShift : 0u;
                    t = lane_bcast(tagv, 0);  // (lane 0 is valid: k > 0)
#if DDGI_AQ_MIX_PICK
                    {
                        // WHICH update's rays a mixed group keeps (DDGI_AQ_MIX_PICK; 0: the first ray's).  1: the OLDER update's — its rays are the
                        // ones the pool is waiting to see the back of: while they last, every group of every queue is mixed —, 2: the majority's.
                        const unsigned long long m_valid = __ballot(valid), m_same = __ballot(valid && tagv == t);
                        if (m_same != m_valid)
                        {
                            const uint32_t t1 = lane_bcast(tagv, __ffsll(static_cast<long long>(m_valid & ~m_same)) - 1);
                            if (DDGI_AQ_MIX_PICK == 1 ? t1 < t : 2 * __popcll(m_same) < __popcll(m_valid)) t = t1;
                        }
                    }
#endif
                    const bool other = valid && tagv != t;
                    if (__ballot(other) != 0ull)
                    {
#if DDGI_AQ_STAGING == 3
                        if (other) push_event(b, slot);  // (variant 3: each to the ring of its own update's parity — three updates meeting: k and k + 2 share one)
#else
                        const uint32_t at = wave_append(other, &sh->eq_tail[b], lane);
                        if (other) (ring_eq + b * kCap)[at % kCap] = static_cast<uint16_t>(slot);
#endif
                        mine = valid && !other;
                    }
                }
                if (mine)
                {
                    const typename Upd::Type U = Upd::make(A, upd_record(t));
                    const int rc = wf_event<Cfg>(A, U, P, s_bits, b, slot, 0u, false, kStats ? &probe : nullptr);
                    posted = rc == 1;
                    if (rc >= 2) ev_bucket = rc - 2;  // the new march ended within its first steps: straight to its event queue
                    freed = rc == 0;                  // the ray is finished: its output is written, the slot is empty
                }
            }
#ifdef DDGI_LAP
            if (kStats) probe.at(15);  // outside the event code: queue traffic, polling
#endif
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
#if DDGI_AQ_QUICK
            {
                // the three kinds of pushes of a group — marches that go on, freed slots, slots whose new march ended at once — reserve
                // their ring indices together (the atomics are issued back to back, ONE wait for all of them) and then write
                const bool to_mq = posted && !(Cfg::ablate(A) & 8);  // DDGI_ABLATE=8: fault injection for the safety-net test
                const unsigned long long m_mq = __ballot(to_mq), m_fq = __ballot(freed);
                const int l_mq = m_mq ? __ffsll(static_cast<long long>(m_mq)) - 1 : 0, l_fq = m_fq ? __ffsll(static_cast<long long>(m_fq)) - 1 : 0;
                uint32_t b_mq = 0, b_fq = 0, at_eq = 0;
                if (m_mq != 0ull && lane == l_mq) b_mq = atomicAdd(&sh->mq_tail, static_cast<uint32_t>(__popcll(m_mq)));
                if (m_fq != 0ull && lane == l_fq) b_fq = atomicAdd(&sh->fq_tail, static_cast<uint32_t>(__popcll(m_fq)));
                [[maybe_unused]] bool ev_staged = false;
#if DDGI_AQ_STAGING == 3
                if (ev_bucket >= 0)
                {
                    push_event(static_cast<uint32_t>(ev_bucket), slot);
                    ev_staged = true;
                }
#elif DDGI_AQ_STAGING
                // (a finished march's entry: the staging ring of its update's parity while that has room — one more trip to the LDS than the bucket's ring)
                if (ev_bucket >= 0 && staging)
                {
                    push_event(static_cast<uint32_t>(ev_bucket), slot);
                    ev_staged = true;
                }
#endif
                if (ev_bucket >= 0 && !ev_staged) at_eq = atomicAdd(&sh->eq_tail[ev_bucket], 1u);
                const unsigned long long below = (1ull << lane) - 1ull;
                if (to_mq) ring_mq[(lane_bcast(b_mq, l_mq) + static_cast<uint32_t>(__popcll(m_mq & below))) % kCap] = static_cast<uint16_t>(slot);
                if (freed) ring_fq[(lane_bcast(b_fq, l_fq) + static_cast<uint32_t>(__popcll(m_fq & below))) % kCap] = static_cast<uint16_t>(slot);
                if (ev_bucket >= 0 && !ev_staged) (ring_eq + ev_bucket * kCap)[at_eq % kCap] = static_cast<uint16_t>(slot);
            }
#else
            aq_push<kCap>(ring_mq, &sh->mq_tail, posted && !(Cfg::ablate(A) & 8), slot, lane);  // DDGI_ABLATE=8: fault injection for the safety-net test
            aq_push<kCap>(ring_fq, &sh->fq_tail, freed, slot, lane);
            if (ev_bucket >= 0)
            {
                const uint32_t at = atomicAdd(&sh->eq_tail[ev_bucket], 1u);
                (ring_eq + ev_bucket * kCap)[at % kCap] = static_cast<uint16_t>(slot);
            }
#endif
            if (b != kBucketRefill)
            {
                const uint32_t n_done = static_cast<uint32_t>(__popcll(__ballot(freed)));
                if (lane == 0 && n_done) atomicSub(&sh->live, n_done);
            }
        }
    }
#ifdef DDGI_LAP
    if (kStats && lane == 0 && role >= march_waves) probe.at(15);  // close the last lap
#endif
    if (kStats && A.stats)  // ddgi_trace_stats [32 + 2 s], [33 + 2 s]: visits of / lanes active in section s of the event code (LaneProbe)
        for (int s = 0; s < kProbeSections; ++s)
        {
            unsigned long long v = probe.visits[s], l = probe.lanes[s];
#ifdef DDGI_LAP
            v = (lane == 0 && probe.lap && role >= march_waves) ? probe.lap[2 + s] : 0ull;
#endif
            for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m), l += __shfl_xor(l, m);
            if (lane == 0 && l) atomicAdd(&A.stats[32 + 2 * s], v), atomicAdd(&A.stats[33 + 2 * s], l);
        }
    if (kStats && A.stats && lane == 0)  // ddgi_trace_stats: [0] march trips, [1] lanes marching in them, [2] event groups, [3] lanes in them, [4] waves
    {
        atomicAdd(&A.stats[role < march_waves ? 0 : 2], st_a);
        atomicAdd(&A.stats[role < march_waves ? 1 : 3], st_b);
        if (st_ma) atomicAdd(&A.stats[0], st_ma), atomicAdd(&A.stats[1], st_mb);
        atomicAdd(&A.stats[4], 1ull);
        atomicAdd(&A.stats[5], st_useful);
        for (int q = 0; q < 8; ++q) atomicAdd(&A.stats[8 + q], st_q[q]);  // (the slots of the round kernel's cycle counters)
    }
    if (status && lane == 0 && aq_load(&sh->abort) != 0u) atomicOr(status, 1u);  // the safety net tripped: the output is not valid
}

constexpr int kAqPool = DDGI_AQ_POOL_CT;      // the usual pool (ddgi_engine.cpp); other sizes take the generic instantiation
constexpr int kAqPoolFast = 1280;  // the fast build's: 16 dwords per slot and rings of 1536 entries next to the cave's 52 KB skip field

// fast: nwords = the skip field's, 16 dwords per slot; ring_cap: entries per ring (k_probe_trace_aq: kCap)
static size_t aq_lds_bytes(int nwords, int pool, bool fast = false, size_t ring_cap = kAqCap)
{
    size_t extra = 16;
#ifdef DDGI_LAP
    extra += 16 * 32 * 4;  // the lap timers' scratch, one row per wave
#endif
    if (DDGI_AQ_STAGING && DDGI_AQ_STAGING != 3) extra += kAqSq * 2 * kAqStagingQueues;  // the staging rings (the cave leaves 3.8 KB beside a pool of 1 344: 3.6 KB of rings + 128 B of control block fit)
    const size_t rings = DDGI_AQ_STAGING == 3 ? ring_cap * 2 * 2 + static_cast<size_t>(kAqEvCap3) * 2 * kAqStagingQueues : ring_cap * 2 * (2 + kAqEventQueues);
    return (kAqCtrlDwords + ((nwords + 3) & ~3)) * sizeof(uint32_t) + static_cast<size_t>(pool) * (fast ? kPoolDwordsFast : kPoolDwords) * 4 + rings + extra;
}

int aq_pool_size(int nwords, size_t lds_limit)
{
    int pool = static_cast<int>(kAqCap);
    while (pool >= kAqMinPool && aq_lds_bytes(nwords, pool) > lds_limit) pool -= 64;
    return pool >= kAqMinPool ? pool : 0;
}
int aq_threads() { return kAqThreads; }
int aq_wgs_per_cu() { return kAqWgsPerCU; }
int aq_pool_usual() { return kAqPool; }

// The fast build's pool for a skip field of nwords_skip words.  plain (one light: the compile-time instantiation, whose
// rings hold kAqCapFast entries): kAqPoolFast when that fits.  Otherwise the largest pool that fits next to rings of kAqCap
// entries; 0: the fast march is not available for this scene.
int aq_pool_size_fast(int nwords_skip, size_t lds_limit, bool plain)
{
    if (plain && aq_lds_bytes(nwords_skip, kAqPoolFast, true, kAqCapFast) <= lds_limit) return kAqPoolFast;
    int pool = kAqPoolFast - 64;
    while (pool >= 1024 && aq_lds_bytes(nwords_skip, pool, true) > lds_limit) pool -= 64;
    return pool >= 1024 ? pool : 0;
}

template <bool kStats, int kPool, class Cfg>
static hipError_t launch_aq(const TraceArgs& args, int pool, int grid_blocks, int march_waves, const AqChain& chain, uint32_t* status, hipStream_t stream)
{
    const size_t lds = Cfg::kFast ? aq_lds_bytes(args.scene.nwords_skip, pool, true, kPool > 0 ? kAqCapFast : kAqCap) : aq_lds_bytes(args.scene.nwords, pool);
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(k_probe_trace_aq<kStats, kPool, Cfg>), 160 * 1024);
    if (e != hipSuccess) return e;
    if (DDGI_AQ_UNIFIED && !Cfg::kFast) march_waves = 0;  // unified waves: every wave serves every queue
    hipLaunchKernelGGL((k_probe_trace_aq<kStats, kPool, Cfg>), dim3(grid_blocks), dim3(kAqThreads), lds, stream, args, pool, std::min(march_waves, kAqThreads / 64 - 1), chain, status);
    return hipGetLastError();
}

// chain.counters[chain.seq % kAqCounters] must be zero: every launch zeroes the counter of the launch kAqChainMax after it (k_probe_trace_aq)
hipError_t launch_probe_trace_aq(const TraceArgs& args, int pool, int grid_blocks, int march_waves, const AqChain& chain, uint32_t* status, hipStream_t stream)
{
    if (args.fast_march)
    {
        // (the tolerance-mode build: no counters kernel, no ablation switches)
        if (pool == kAqPoolFast && args.nl == 1)
            return args.ddgi ? launch_aq<false, kAqPoolFast, CfgPlain<1, true>>(args, pool, grid_blocks, march_waves, chain, status, stream)
                             : launch_aq<false, kAqPoolFast, CfgPlain<0, true>>(args, pool, grid_blocks, march_waves, chain, status, stream);
        return launch_aq<false, 0, CfgRuntimeT<true>>(args, pool, grid_blocks, march_waves, chain, status, stream);
    }
    // (the counters build of the headline instantiation — one light, REF, the usual pool: its queue traffic is what tools/aq_stats.py and
    // tools/valu_attribution.py are about; everything else counts in the generic one, which takes one inline step where this takes two)
    if (args.stats && pool == kAqPool && args.nl == 1 && args.ablate == 0 && !args.ddgi) return launch_aq<true, kAqPool, CfgPlain<0>>(args, pool, grid_blocks, march_waves, chain, status, stream);
    if (args.stats) return launch_aq<true, 0, CfgRuntime>(args, pool, grid_blocks, march_waves, chain, status, stream);
    if (pool == kAqPool && args.nl == 1 && args.ablate == 0)
        return args.ddgi ? launch_aq<false, kAqPool, CfgPlain<1>>(args, pool, grid_blocks, march_waves, chain, status, stream)
                         : launch_aq<false, kAqPool, CfgPlain<0>>(args, pool, grid_blocks, march_waves, chain, status, stream);
    if (pool == kAqPool && args.nl > 1 && args.ablate == 0)
    {
        if (args.nl == 4)
            return args.ddgi ? launch_aq<false, kAqPool, CfgMulti<1, false, 4>>(args, pool, grid_blocks, march_waves, chain, status, stream)
                             : launch_aq<false, kAqPool, CfgMulti<0, false, 4>>(args, pool, grid_blocks, march_waves, chain, status, stream);
        return args.ddgi ? launch_aq<false, kAqPool, CfgMulti<1>>(args, pool, grid_blocks, march_waves, chain, status, stream)
                         : launch_aq<false, kAqPool, CfgMulti<0>>(args, pool, grid_blocks, march_waves, chain, status, stream);
    }
    return launch_aq<false, 0, CfgRuntime>(args, pool, grid_blocks, march_waves, chain, status, stream);
}

// LDS bytes of k_probe_trace_wf for a pool of `pool` rays
static size_t wf_lds_bytes(int nwords, int pool, bool multi_light)
{
    return (32 + ((nwords + 3) & ~3)) * sizeof(uint32_t) + static_cast<size_t>(pool) * wf_dwords_per_ray(multi_light) * 4 +
           static_cast<size_t>(pool) * 6 + 16;  // three 16-bit slot lists
}

// Largest pool (multiple of 64, at most 2 per lane) that fits in `lds_limit` bytes; 0 if not even
// one ray per lane fits (then the caller uses k_probe_trace_ref).
int wf_pool_size(int nwords, bool multi_light, size_t lds_limit, int threads, int max_pool)
{
    int pool = 3 * threads;  // at most 3 slots per lane (the bucket pass keeps that many ranks in registers)
    if (max_pool > 0) pool = std::min(pool, std::max(threads, max_pool / 64 * 64));
    while (pool >= threads && wf_lds_bytes(nwords, pool, multi_light) > lds_limit) pool -= 64;
    return pool >= threads ? pool : 0;
}

template <int T, int B, bool kStats>
static hipError_t launch_wf(const TraceArgs& args, int pool, int grid_blocks, uint32_t* work_counter, hipStream_t stream)
{
    const size_t lds = wf_lds_bytes(args.scene.nwords, pool, args.nl > 1);
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(k_probe_trace_wf<T, B, kStats>), 160 * 1024 / B);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(work_counter, 0, sizeof(uint32_t), stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_probe_trace_wf<T, B, kStats>), dim3(grid_blocks), dim3(T), lds, stream, args, pool, work_counter);
    return hipGetLastError();
}

// threads: 1024 (one workgroup per CU) or 512 (two per CU, each with half the LDS)
hipError_t launch_probe_trace_wf(const TraceArgs& args_in, int threads, int pool, int grid_blocks, uint32_t* work_counter, hipStream_t stream)
{
    // rays per claim: small enough that a short launch (one rank's slab of a sharded grid) still gives
    // every workgroup several claims, at most kWfChunk
    TraceArgs args = args_in;
    if (args.wf_chunk <= 0)
    {
        const uint32_t per_claim = args.n_rays / (static_cast<uint32_t>(grid_blocks > 0 ? grid_blocks : 1) * 8u);
        args.wf_chunk = static_cast<int>(std::min<uint32_t>(kWfChunk, std::max<uint32_t>(256u, (per_claim + 63u) & ~63u)));
    }
    if (threads == 512) return launch_wf<512, 2, false>(args, pool, grid_blocks, work_counter, stream);
    if (args.stats) return launch_wf<1024, 1, true>(args, pool, grid_blocks, work_counter, stream);
    return launch_wf<1024, 1, false>(args, pool, grid_blocks, work_counter, stream);
}

}  // namespace ddgi
