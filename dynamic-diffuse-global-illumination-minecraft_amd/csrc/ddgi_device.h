// ddgi_device.h — device-side building blocks shared by the probe-trace kernels: per-ray RNG, the
// resumable voxel march, the light-sphere test, hemisphere sampling, texel packing.
// Pinned binary32 arithmetic throughout (DESIGN.md "Arithmetic pinning").
#pragma once

#include <hip/hip_runtime.h>

#include "ddgi_scene.h"
#include "ddgi_types.h"

namespace ddgi {

#define DDGI_D __device__ __forceinline__

// ---- per-ray RNG: wang_hash seed + xorshift32 (probe_pass.comp:45-71) -------------------------

DDGI_D uint32_t wang_hash(uint32_t seed)
{
    seed = (seed ^ 61u) ^ (seed >> 16);
    seed *= 9u;
    seed = seed ^ (seed >> 4);
    seed *= 0x27d4eb2du;
    seed = seed ^ (seed >> 15);
    return seed;
}

DDGI_D float rng_next(uint32_t& s)
{
    s ^= (s << 13);
    s ^= (s >> 17);
    s ^= (s << 5);
    return static_cast<float>(s) * 0x1.0p-32f;  // uint -> float (RNE), exact scale by 2^-32
}

// ---- one voxel march + the light spheres of intersect_scene, as resumable per-lane state ------

struct March
{
    f3 ro;   // ray origin
    f3 rd;   // ray direction exactly as given (bounce rays are not unit length)
    f3 dn;   // normalize(rd): the direction grid_march steps along (intersection.glsl:1055)
    f3 inv;  // 1/dn per axis (+inf where dn == 0)                                  [P5]
    f3 cc;   // 1 where dn >= 0 else 0: boundary distance = (cc - fract(p)) * inv   [P5]
    f3 p;    // current march position
    float t;   // curr_t
    float tl;  // nearest light-sphere hit along (ro, rd), +inf if none (intersection.glsl:1264-1279)
    int it;    // march iterations done
    int lid;   // which light gave tl
    int cell;  // linear (clamped) cell index reached by the last step
};

// d: a component of a normalised direction a * (1 / sqrt(dot(a, a))): |d| <= 1 + 2^-22, or 0 / inf / NaN for a degenerate a
// (dot under- or overflowed) — inside pm::rcp_upto_2p94's domain
DDGI_D float axis_inv(float d) { return d == 0.0f ? __builtin_inff() : pm::rcp_upto_2p94(d); }
// normalize3 (P3) of a direction that is itself the output of a normalisation or of hemisphere_dir: dot(d, d) is about 1,
// or 0 / inf / NaN when that normalisation was degenerate — never in (0, 2^-96), the part of the line pm::rcp_sqrt_core gets wrong
DDGI_D f3 normalize3_of_unit(f3 d) { return d * pm::rcp_sqrt_core(dot3(d, d)); }

// ---- XCD-aware block order ---------------------------------------------------------------------------------------------------------
// The hardware deals consecutive workgroups round-robin over the chip's 8 XCDs, each with its own L2 (MI355X_MICROARCH.md): blocks b, b + 1, ...
// b + 7 of a launch run on 8 different L2s.  A kernel whose CONSECUTIVE blocks share data (a batch of shading points in cage order: neighbouring
// rows of cages share half their probes' tiles) wants them behind ONE L2: XCD x takes the x-th contiguous eighth of the logical blocks.
// -> the logical block of hardware block b of nb; a bijection for every nb.
#ifndef DDGI_XCD_ORDER
#define DDGI_XCD_ORDER 1
#endif
DDGI_D uint32_t xcd_block(uint32_t b, uint32_t nb)
{
#if DDGI_XCD_ORDER
    const uint32_t q = nb >> 3, r = nb & 7u, x = b & 7u, j = b >> 3;
    return x * q + (x < r ? x : r) + j;
#else
    return b;
#endif
}

// ---- the per-update part of a launch's arguments (ddgi_types.h: UpdK), as the trace code reads it ----------------------------
// UpdOfArgs: straight from the kernel's own arguments (k_probe_trace_ref, k_probe_trace_wf, k_render_primary).
// UpdOfRing: from a record of the queue kernel's per-update ring, through the constant address space — with a wave-uniform
// record (k_probe_trace_aq makes it so) every access is a scalar load placed where the value is used, like a kernel argument's.
struct UpdOfArgs
{
    const TraceArgs& A;
    DDGI_D LightK light(int i) const { return A.lights[i]; }
    DDGI_D f3 light_pos(int i) const { return f3{A.lights[i].pos[0], A.lights[i].pos[1], A.lights[i].pos[2]}; }
    DDGI_D void rot(float m9[9]) const
    {
        for (int k = 0; k < 9; ++k) m9[k] = A.rot[k];
    }
    DDGI_D uint32_t frame_key() const { return A.frame_key; }
    DDGI_D const float4* rays() const { return A.rays; }
    DDGI_D float* rad_rgb() const { return A.rad_rgb; }
    DDGI_D float* rad_dd() const { return A.rad_dd; }
    DDGI_D const uint8_t* vis() const { return A.vis; }
    DDGI_D const uint32_t* vis_occ() const { return A.vis_occ; }
    DDGI_D const uint8_t* vis_more(int k) const { return A.vis_more[k]; }
};
struct UpdOfRing
{
    typedef const __attribute__((address_space(4))) uint32_t* Words;
    Words w;
    static constexpr int kLights = offsetof(UpdK, lights) / 4, kRot = offsetof(UpdK, rot) / 4, kKey = offsetof(UpdK, frame_key) / 4, kRays = offsetof(UpdK, rays) / 4,
                         kRgb = offsetof(UpdK, rad_rgb) / 4, kDd = offsetof(UpdK, rad_dd) / 4, kVis = offsetof(UpdK, vis) / 4, kOcc = offsetof(UpdK, vis_occ) / 4,
                         kMore = offsetof(UpdK, vis_more) / 4;
    // The record was WRITTEN by this kernel (k_probe_trace_aq: copy_record), and the constant address space promises LLVM memory that never changes:
    // nothing but the pointer ties these loads to the copy.  So the pointer comes out of an asm the optimizer cannot see through and that clobbers
    // memory, placed where the record is known to be complete (behind the workgroup barrier / the acquire of the group's slots): a load of the
    // record cannot be hoisted above it or merged with one made through an earlier UpdOfRing — by the language's rules, not by the code's layout.
    DDGI_D explicit UpdOfRing(const uint32_t* record)
    {
        uintptr_t a = reinterpret_cast<uintptr_t>(record);
#if !defined(DDGI_UPD_LAUNDER) || DDGI_UPD_LAUNDER
        // (the record is one per event group, wave-uniform by construction — k_probe_trace_aq — where the compiler cannot always see it)
        uint32_t lo = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(a)), hi = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(a >> 32));
        asm volatile("" : "+s"(lo), "+s"(hi) : : "memory");
        a = static_cast<uintptr_t>(lo) | (static_cast<uintptr_t>(hi) << 32);
#endif
        w = reinterpret_cast<Words>(a);
    }
    DDGI_D float f(int i) const { return __uint_as_float(w[i]); }
    template <class T>
    DDGI_D T* ptr(int i) const
    {
        // (through the global address space: a pointer made from an integer is otherwise generic — flat_load / flat_store)
        typedef __attribute__((address_space(1))) T* Global;
        return (T*)reinterpret_cast<Global>(static_cast<uintptr_t>(w[i]) | (static_cast<uintptr_t>(w[i + 1]) << 32));
    }
    DDGI_D LightK light(int i) const
    {
        const int o = kLights + 7 * i;
        LightK L;
        L.intensity = f(o), L.col[0] = f(o + 1), L.col[1] = f(o + 2), L.col[2] = f(o + 3), L.pos[0] = f(o + 4), L.pos[1] = f(o + 5), L.pos[2] = f(o + 6);
        return L;
    }
    DDGI_D f3 light_pos(int i) const { return f3{f(kLights + 7 * i + 4), f(kLights + 7 * i + 5), f(kLights + 7 * i + 6)}; }
    DDGI_D void rot(float m9[9]) const
    {
        for (int k = 0; k < 9; ++k) m9[k] = f(kRot + k);
    }
    DDGI_D uint32_t frame_key() const { return w[kKey]; }
    DDGI_D const float4* rays() const { return ptr<const float4>(kRays); }
    DDGI_D float* rad_rgb() const { return ptr<float>(kRgb); }
    DDGI_D float* rad_dd() const { return ptr<float>(kDd); }
    DDGI_D const uint8_t* vis() const { return ptr<const uint8_t>(kVis); }
    DDGI_D const uint32_t* vis_occ() const { return ptr<const uint32_t>(kOcc); }
    DDGI_D const uint8_t* vis_more(int k) const { return ptr<const uint8_t>(kMore + 2 * k); }
};
static_assert(sizeof(LightK) == 28, "UpdOfRing::light reads a light as 7 consecutive words");

// The light-sphere half of intersect_scene (intersection.glsl:1264-1279): nearest hit of the ray
// (o, d) with the radius-0.1 spheres around the lights; +inf / -1 if none.
// Unit-sphere quadratic in a space scaled by 10 (x/0.1 := x*10, P5).
// kNl > 0: the number of lights is known at compile time (the queue kernel's one-light instantiation)
template <int kNl = 0, class Upd>
DDGI_D void light_spheres(f3 o, f3 d, const TraceArgs& A, const Upd& U, float& tl_out, int& lid_out)
{
    float closest = __builtin_inff();
    int lid = -1;
    const int nl = kNl > 0 ? kNl : A.nl;
    for (int i = 0; i < nl; ++i)
    {
        const f3 lp = U.light_pos(i);
        const f3 so = (o - lp) * 10.0f;
        const f3 sd = d * 10.0f;
        const float qa = dot3(sd, sd);
        const float qb = -dot3(sd, so);
        const float qc = dot3(so, so) - 1.0f;
        float disc = qb * qb - qa * qc;
        // A ray that misses the sphere (disc <= 0, or NaN): the reference sets the root term to INF and both
        // candidate roots (qb -+ INF) / qa then fail 0 < t < closest whatever qa is (-inf, NaN, or inf * 0 = NaN),
        // so nothing changes — skipped as a whole (bounce rays almost never point at a radius-0.1 sphere)
        if (disc > 0.0f)
        {
            disc = sqrtf(disc);
            const float inv_a = 1.0f / qa;
            float t1 = (qb - disc) * inv_a;
            float t2 = (qb + disc) * inv_a;
            t1 = (0.0f < t1 && t1 < closest) ? t1 : __builtin_inff();
            t2 = (0.0f < t2 && t2 < closest) ? t2 : __builtin_inff();
            const float ts = gl_min(t1, t2);
            if (ts < closest) lid = i;
            closest = gl_min(ts, closest);
        }
    }
    tl_out = closest;
    lid_out = lid;
}

DDGI_D void start_march(March& m, f3 o, f3 d, const TraceArgs& A)
{
    m.ro = o;
    m.rd = d;
    m.dn = normalize3(d);
    m.inv = f3{axis_inv(m.dn.x), axis_inv(m.dn.y), axis_inv(m.dn.z)};
    m.cc = f3{m.dn.x >= 0.0f ? 1.0f : 0.0f, m.dn.y >= 0.0f ? 1.0f : 0.0f, m.dn.z >= 0.0f ? 1.0f : 0.0f};
    m.p = o;
    m.t = 0.0f;
    m.it = 0;
    light_spheres<0>(o, d, A, UpdOfArgs{A}, m.tl, m.lid);
}

// Raw linear cell index of voxel id (x,y,z) clamped into the baked box.  Outside the box the world is
// the extrusion of the border layer (ddgi_scene_bake.cpp), so clamping is exact.
DDGI_D int cell_index(const SceneK& S, int x, int y, int z)
{
    x = min(max(x, S.lo[0]), S.hi[0]);
    y = min(max(y, S.lo[1]), S.hi[1]);
    z = min(max(z, S.lo[2]), S.hi[2]);
    // clamped coordinates and pitches fit 24 bits: v_mad_i32_i24 instead of quarter-rate v_mul_lo_u32
    return __mul24(z, S.nxy) + __mul24(y, S.nx) + x;  // "raw" index; block-type index = raw - S.bias
}

// One grid_march iteration (intersection.glsl:1059-1069).  Returns true if the voxel reached is
// occupied.
DDGI_D bool march_step(March& m, const SceneK& S, const uint32_t* __restrict__ s_bits)
{
    const float fx = gl_fract(m.p.x), fy = gl_fract(m.p.y), fz = gl_fract(m.p.z);
    const float tx = (m.cc.x - fx) * m.inv.x;
    const float ty = (m.cc.y - fy) * m.inv.y;
    const float tz = (m.cc.z - fz) * m.inv.z;
    const float step = fminf(fminf(tx, ty), tz) + 0.0001f;
    m.t += step;
    m.p = ray_at(m.ro, m.dn, m.t);
    // voxel id = ceil(p) (Q5), clamped into the baked box and linearised IN FLOAT: every quantity is an
    // integer far below 2^24, so fmed3 / fma are exact and one conversion replaces three
    // (9 VALU instead of 15 for ceil+cvt+min/max+mad).
    const float kx = __builtin_amdgcn_fmed3f(ceilf(m.p.x), S.lo_f[0], S.hi_f[0]);
    const float ky = __builtin_amdgcn_fmed3f(ceilf(m.p.y), S.lo_f[1], S.hi_f[1]);
    const float kz = __builtin_amdgcn_fmed3f(ceilf(m.p.z), S.lo_f[2], S.hi_f[2]);
    const int idx = static_cast<int>(fmaf(kz, S.nxy_f, fmaf(ky, S.nx_f, kx)));
    m.it += 1;
    m.cell = idx;
    // the bitmap is stored so that the raw index addresses it directly (SceneK::bias32)
    const uint32_t* __restrict__ base = s_bits - (S.bias32 >> 5);
    return (base[idx >> 5] >> (idx & 31)) & 1u;
}

// march_step for the wavefront kernel's unrolled bursts: the box's upper corner comes in VGPRs (a VALU
// instruction can read only one SGPR, so with S.hi_f in SGPRs every v_med3 costs an extra v_mov), the
// iteration counter is left to the caller (one add per burst instead of one per step) and the
// occupancy bit is extracted with v_bfe_u32 (which takes the bit offset modulo 32 by itself).
typedef float f2v __attribute__((ext_vector_type(2)));
// The voxel id's clamp into the baked box, per axis.  DDGI_EXP_NOCLAMP (experiment, round 6: what the three v_med3 of a step cost — NOT exact, a march that leaves
// the box looks up whatever lies beside the bitmap): 1 = no clamp at all (the bound), 2 = x and z as they come, y against the box's top only (what a scene whose
// side and bottom layers are solid would need).
#if defined(DDGI_EXP_NOCLAMP) && DDGI_EXP_NOCLAMP == 1
#define DDGI_CLAMP_X(v, lo, hi) (v)
#define DDGI_CLAMP_Y(v, lo, hi) (v)
#define DDGI_CLAMP_Z(v, lo, hi) (v)
#elif defined(DDGI_EXP_NOCLAMP) && DDGI_EXP_NOCLAMP == 2
#define DDGI_CLAMP_X(v, lo, hi) (v)
#define DDGI_CLAMP_Y(v, lo, hi) fminf(v, hi)
#define DDGI_CLAMP_Z(v, lo, hi) (v)
#else
#define DDGI_CLAMP_X(v, lo, hi) __builtin_amdgcn_fmed3f(v, lo, hi)
#define DDGI_CLAMP_Y(v, lo, hi) __builtin_amdgcn_fmed3f(v, lo, hi)
#define DDGI_CLAMP_Z(v, lo, hi) __builtin_amdgcn_fmed3f(v, lo, hi)
#endif
#if defined(DDGI_EXP_NOCLAMP) && DDGI_EXP_NOCLAMP
#define DDGI_EXP_RECLAMP(m, S, hi) ((m).cell = static_cast<int>(fmaf(__builtin_amdgcn_fmed3f(ceilf((m).p.z), (S).lo_f[2], (hi).z), (S).nxy_f, fmaf(__builtin_amdgcn_fmed3f(ceilf((m).p.y), (S).lo_f[1], (hi).y), (S).nx_f, __builtin_amdgcn_fmed3f(ceilf((m).p.x), (S).lo_f[0], (hi).x)))))
#else
#define DDGI_EXP_RECLAMP(m, S, hi) ((void)0)
#endif
// ray_at with the y and z components as ONE v_pk_fma_f32 (the same two IEEE fused multiply-adds, one issue slot instead of two)
DDGI_D f3 ray_at_pk(f3 o, f3 d, float t)
{
#if defined(DDGI_RAY_PK) && !DDGI_RAY_PK
    return ray_at(o, d, t);
#else
    const f2v yz = __builtin_elementwise_fma(f2v{d.y, d.z}, f2v{t, t}, f2v{o.y, o.z});
    return f3{fmaf(d.x, t, o.x), yz.x, yz.y};
#endif
}
DDGI_D bool march_step_burst(March& m, const SceneK& S, const uint32_t* __restrict__ s_bits, f3 hi)
{
    const float fx = gl_fract(m.p.x), fy = gl_fract(m.p.y), fz = gl_fract(m.p.z);
    const float tx = (m.cc.x - fx) * m.inv.x;
    // y and z as a packed pair: v_pk_add_f32 / v_pk_mul_f32 are the same IEEE operations, two per issue
    const f2v tyz = (f2v{m.cc.y, m.cc.z} - f2v{fy, fz}) * f2v{m.inv.y, m.inv.z};
    const float step = fminf(fminf(tx, tyz.x), tyz.y) + 0.0001f;
    m.t += step;
    m.p = ray_at_pk(m.ro, m.dn, m.t);
    const float kx = DDGI_CLAMP_X(ceilf(m.p.x), S.lo_f[0], hi.x);
    const float ky = DDGI_CLAMP_Y(ceilf(m.p.y), S.lo_f[1], hi.y);
    const float kz = DDGI_CLAMP_Z(ceilf(m.p.z), S.lo_f[2], hi.z);
    const int idx = static_cast<int>(fmaf(kz, S.nxy_f, fmaf(ky, S.nx_f, kx)));
    m.cell = idx;
    const uint32_t* __restrict__ base = s_bits - (S.bias32 >> 5);
    return __builtin_amdgcn_ubfe(base[idx >> 5], static_cast<uint32_t>(idx), 1u) != 0u;
}

// march_step_burst for a burst WITHOUT exec-mask predication: a lane whose march has ended (`frozen`) takes a step of length 0 —
// t, hence the position and the voxel looked up, stay what they were, bit for bit — instead of being masked out of the step.  One
// v_cndmask per step in place of the s_and_saveexec / s_or pair and its bookkeeping around every step of the unrolled burst.
DDGI_D bool march_step_frozen(March& m, const SceneK& S, const uint32_t* __restrict__ s_bits, f3 hi, bool frozen)
{
    const float fx = gl_fract(m.p.x), fy = gl_fract(m.p.y), fz = gl_fract(m.p.z);
    const float tx = (m.cc.x - fx) * m.inv.x;
    const f2v tyz = (f2v{m.cc.y, m.cc.z} - f2v{fy, fz}) * f2v{m.inv.y, m.inv.z};
    const float step = fminf(fminf(tx, tyz.x), tyz.y) + 0.0001f;
    m.t += frozen ? 0.0f : step;
    m.p = ray_at_pk(m.ro, m.dn, m.t);
    const float kx = DDGI_CLAMP_X(ceilf(m.p.x), S.lo_f[0], hi.x);
    const float ky = DDGI_CLAMP_Y(ceilf(m.p.y), S.lo_f[1], hi.y);
    const float kz = DDGI_CLAMP_Z(ceilf(m.p.z), S.lo_f[2], hi.z);
    const int idx = static_cast<int>(fmaf(kz, S.nxy_f, fmaf(ky, S.nx_f, kx)));
    m.cell = idx;
    const uint32_t* __restrict__ base = s_bits - (S.bias32 >> 5);
    return __builtin_amdgcn_ubfe(base[idx >> 5], static_cast<uint32_t>(idx), 1u) != 0u;
}

// march_step_frozen WITHOUT a "has ended" state.  A march that has landed in an occupied voxel must stand still for the rest of
// the burst; the boolean version carries that as a lane mask, which the compiler keeps in an SGPR pair: v_bfe_u32, v_cmp_ne
// (VALU -> SGPR), s_or_b64, v_cndmask (SGPR -> VALU) per step — a scalar instruction and two trips between the register files on
// the march's dependent chain.  But a march that stands still looks the SAME voxel up again: the bit of the voxel it is in says
// "stand still" by itself, step after step.  So a step is v_bfe_i32 (the voxel's bit, sign-extended: `occ`, 0 or ~0) and the next
// step's length is step & ~occ (v_bfi_b32) — t + 0 == t (t >= +0), hence the same position, voxel and bit as before.
// occ: the previous step's (0 in front of a burst: a march in flight stands in an empty voxel, and grid_march never looks at the
// voxel it starts in, intersection.glsl:1059-1069).  A lane without a march steps along a zero direction (inv = 0: the position
// stays at ro whatever t does) and needs no mask either.
DDGI_D void march_step_masked(March& m, const SceneK& S, const uint32_t* __restrict__ s_bits, f3 hi, uint32_t& occ)
{
    const float fx = gl_fract(m.p.x), fy = gl_fract(m.p.y), fz = gl_fract(m.p.z);
    const float tx = (m.cc.x - fx) * m.inv.x;
    const f2v tyz = (f2v{m.cc.y, m.cc.z} - f2v{fy, fz}) * f2v{m.inv.y, m.inv.z};
    const float step = fminf(fminf(tx, tyz.x), tyz.y) + 0.0001f;
    uint32_t step_bits;
    asm("v_bfi_b32 %0, %1, 0, %2" : "=v"(step_bits) : "v"(occ), "v"(__float_as_uint(step)));  // occ ? +0.0f : step
    m.t += __uint_as_float(step_bits);
    m.p = ray_at_pk(m.ro, m.dn, m.t);
    const float kx = DDGI_CLAMP_X(ceilf(m.p.x), S.lo_f[0], hi.x);
    const float ky = DDGI_CLAMP_Y(ceilf(m.p.y), S.lo_f[1], hi.y);
    const float kz = DDGI_CLAMP_Z(ceilf(m.p.z), S.lo_f[2], hi.z);
    const int idx = static_cast<int>(fmaf(kz, S.nxy_f, fmaf(ky, S.nx_f, kx)));
    m.cell = idx;
    const uint32_t* __restrict__ base = s_bits - (S.bias32 >> 5);
    occ = static_cast<uint32_t>(__builtin_amdgcn_sbfe(static_cast<int>(base[idx >> 5]), static_cast<uint32_t>(idx), 1u));
}

// ---- the fast march (tolerance mode, opt-in: ddgi_set_tuning "fast_march") -----------------------------------------
// grid_march (intersection.glsl:1051-1100) re-derives every step from the position it has reached — t += (distance to
// the next voxel boundary) + 1e-4 — so the position after crossing a given plane is that plane's t + 1e-4 whatever came
// before it, up to rounding.  A march may therefore cross several planes in ONE step when the voxels in between are
// known to be empty, land where grid_march would have landed after as many unit steps, and go on exactly as grid_march
// does: same hit voxel, hit position equal to a few ulp — except for rays that graze a voxel edge within those ulp.
// What is known comes from the scene's skip field (ddgi_host.h: build_skip_field): code 0 = occupied, else every voxel
// within Chebyshev distance code - 1 is empty, so the next boundary that matters is `code` voxels away on each axis:
//     t_axis = (code - f) / |d_axis| going up,  (code - 1 + f) / |d_axis| going down,   f = fract(p_axis)
// (code = 1 is grid_march's own step).  grid_march's limit of 125 iterations is kept as a limit on the planes crossed
// (fast_march_planes): per axis the position moves one way only, so the planes crossed since the origin are the
// Manhattan distance between the voxel ids — no counter travels with the march.
// Results are NOT bit-equal to the exact march; tests/test_gpu_fast_march.py states and checks the tolerance.
struct FastMarch
{
    f3 ro, dn;
    f3 ainv;  // 1 / |dn| per axis (1e30 where dn == 0)
    f3 nsgn;  // -1 where dn >= 0 else +1
    f3 c1;    //  0 where dn >= 0 else  1
    f3 p;
    float t, tl;
    float code;  // skip code of the voxel that holds p, as a float (>= 1 while the march is going)
    int cell;
};

DDGI_D float fast_axis_ainv(float d) { return d == 0.0f ? 1.0e30f : __builtin_amdgcn_rcpf(fabsf(d)); }

DDGI_D void fast_march_begin(FastMarch& m, f3 ro, f3 dn, float t, float tl)
{
    m.ro = ro, m.dn = dn;
    m.ainv = f3{fast_axis_ainv(dn.x), fast_axis_ainv(dn.y), fast_axis_ainv(dn.z)};
    m.nsgn = f3{dn.x >= 0.0f ? -1.0f : 1.0f, dn.y >= 0.0f ? -1.0f : 1.0f, dn.z >= 0.0f ? -1.0f : 1.0f};
    m.c1 = f3{dn.x >= 0.0f ? 0.0f : 1.0f, dn.y >= 0.0f ? 0.0f : 1.0f, dn.z >= 0.0f ? 0.0f : 1.0f};
    m.t = t, m.tl = tl;
    m.p = ray_at(ro, dn, t);
    m.code = 1.0f;  // nothing known about the start voxel: the first step is grid_march's
    m.cell = 0;
}

// One step; returns the skip code of the voxel reached (0: occupied).  s_skip: the skip field (LDS).
// Distance to the boundary `code` voxels on, per axis, with f = fract(p):  going up (code - f) / |d|, going down
// (code - 1 + f) / |d| — for code = 1 exactly grid_march's max(-f / d, (1 - f) / d), INCLUDING its zero-length step from a
// position that sits on a plane and goes down (f = 0): probes stand on integer coordinates, and what grid_march looks up
// after that step (the voxel diagonally across) is part of the reference's result.
DDGI_D uint32_t fast_march_step(FastMarch& m, const SceneK& S, const uint32_t* __restrict__ s_skip, f3 hi)
{
    const float fx = gl_fract(m.p.x);
    const f2v fyz = f2v{gl_fract(m.p.y), gl_fract(m.p.z)};
    const float tx = fmaf(m.nsgn.x, fx, m.code - m.c1.x) * m.ainv.x;
    const f2v nyz = f2v{fmaf(m.nsgn.y, fyz.x, m.code - m.c1.y), fmaf(m.nsgn.z, fyz.y, m.code - m.c1.z)};
    const f2v tyz = nyz * f2v{m.ainv.y, m.ainv.z};
    m.t += fminf(fminf(tx, tyz.x), tyz.y) + 0.0001f;
    m.p = ray_at(m.ro, m.dn, m.t);
    const float kx = __builtin_amdgcn_fmed3f(ceilf(m.p.x), S.lo_f[0], hi.x);
    const float ky = __builtin_amdgcn_fmed3f(ceilf(m.p.y), S.lo_f[1], hi.y);
    const float kz = __builtin_amdgcn_fmed3f(ceilf(m.p.z), S.lo_f[2], hi.z);
    const int idx = static_cast<int>(fmaf(kz, S.nxy_f, fmaf(ky, S.nx_f, kx)));
    m.cell = idx;
    const uint32_t* __restrict__ base = s_skip - (S.bias32 >> 4);
    const uint32_t code = __builtin_amdgcn_ubfe(base[idx >> 4], static_cast<uint32_t>(idx) << 1, 2u);
    m.code = static_cast<float>(code);
    return code;
}

// Planes crossed between the march's origin and its position = the iterations grid_march would have spent to get here
// (one plane per iteration; two planes within 1e-4 of each other count twice here, once there).
DDGI_D float fast_march_planes(const FastMarch& m)
{
    return fabsf(ceilf(m.p.x) - ceilf(m.ro.x)) + fabsf(ceilf(m.p.y) - ceilf(m.ro.y)) + fabsf(ceilf(m.p.z) - ceilf(m.ro.z));
}

// Block type of the voxel a march ended in (its id `cell` = ceil(p), raw linear index `raw`).  The
// baked table is exact inside the box and for everything that is an extrusion of its border layer;
// the one exception is the cave's floor band (y < -15): there getBlockAt decides 11/12/13 from an
// fbm of (x, z) BEFORE it looks at the hollow (intersection.glsl:726-742), so it continues under the
// solid rock outside the box.  No ray can reach those voxels from the hollow, but a probe placed in
// the rock there starts inside one — evaluate the rule itself for them.
// (kept out of line: it is reached by almost no ray, and inlined into the march loops its fbm — sines in
// binary64 — costs every ray registers and instruction-cache space)
__device__ __attribute__((noinline, cold)) inline int cave_floor_band_type(f3 cell) { return block_at(cell, 0); }

DDGI_D int hit_block_type(const SceneK& S, int scene_id, f3 cell, int raw)
{
    if (scene_id == 0 && cell.y < -15.0f && (cell.x < S.lo_f[0] || cell.x > S.hi_f[0] || cell.z < S.lo_f[2] || cell.z > S.hi_f[2]))
        return cave_floor_band_type(cell);
    return S.types[static_cast<uint32_t>(raw - S.bias)];  // (raw >= bias: the clamped voxel is in the box; unsigned: the base stays in scalar registers)
}

// march_step_frozen for TWO marches of a lane at once, statement by statement: the two dependent chains alternate in the
// instruction stream, so that one's next instruction is ready when the other's has just issued (a wave issues in order).
DDGI_D void march_step_frozen2(March (&m)[2], const SceneK& S, const uint32_t* __restrict__ s_bits, f3 hi, const bool (&frozen)[2], bool (&occ)[2])
{
    float fx[2], fy[2], fz[2], tx[2], step[2], kx[2], ky[2], kz[2];
    f2v tyz[2];
    int idx[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) fx[q] = gl_fract(m[q].p.x);
#pragma unroll
    for (int q = 0; q < 2; ++q) fy[q] = gl_fract(m[q].p.y);
#pragma unroll
    for (int q = 0; q < 2; ++q) fz[q] = gl_fract(m[q].p.z);
#pragma unroll
    for (int q = 0; q < 2; ++q) tx[q] = (m[q].cc.x - fx[q]) * m[q].inv.x;
#pragma unroll
    for (int q = 0; q < 2; ++q) tyz[q] = (f2v{m[q].cc.y, m[q].cc.z} - f2v{fy[q], fz[q]}) * f2v{m[q].inv.y, m[q].inv.z};
#pragma unroll
    for (int q = 0; q < 2; ++q) step[q] = fminf(fminf(tx[q], tyz[q].x), tyz[q].y) + 0.0001f;
#pragma unroll
    for (int q = 0; q < 2; ++q) m[q].t += frozen[q] ? 0.0f : step[q];
#pragma unroll
    for (int q = 0; q < 2; ++q) m[q].p.x = fmaf(m[q].dn.x, m[q].t, m[q].ro.x);
#pragma unroll
    for (int q = 0; q < 2; ++q) m[q].p.y = fmaf(m[q].dn.y, m[q].t, m[q].ro.y);
#pragma unroll
    for (int q = 0; q < 2; ++q) m[q].p.z = fmaf(m[q].dn.z, m[q].t, m[q].ro.z);
#pragma unroll
    for (int q = 0; q < 2; ++q) kx[q] = __builtin_amdgcn_fmed3f(ceilf(m[q].p.x), S.lo_f[0], hi.x);
#pragma unroll
    for (int q = 0; q < 2; ++q) ky[q] = __builtin_amdgcn_fmed3f(ceilf(m[q].p.y), S.lo_f[1], hi.y);
#pragma unroll
    for (int q = 0; q < 2; ++q) kz[q] = __builtin_amdgcn_fmed3f(ceilf(m[q].p.z), S.lo_f[2], hi.z);
#pragma unroll
    for (int q = 0; q < 2; ++q) idx[q] = static_cast<int>(fmaf(kz[q], S.nxy_f, fmaf(ky[q], S.nx_f, kx[q])));
    const uint32_t* __restrict__ base = s_bits - (S.bias32 >> 5);
    uint32_t word[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) m[q].cell = idx[q], word[q] = base[idx[q] >> 5];
#pragma unroll
    for (int q = 0; q < 2; ++q) occ[q] = __builtin_amdgcn_ubfe(word[q], static_cast<uint32_t>(idx[q]), 1u) != 0u;
}

DDGI_D void march_step_frozen2(March (&m)[1], const SceneK& S, const uint32_t* __restrict__ s_bits, f3 hi, const bool (&frozen)[1], bool (&occ)[2])
{
    occ[0] = march_step_frozen(m[0], S, s_bits, hi, frozen[0]), occ[1] = false;  // (one march per lane: never called, only compiled)
}

// True when the march can no longer hit a block: the position is outside the baked box on some
// axis, moving away from it, and the border layer it left through is entirely empty (so the whole
// half space beyond is empty).  Skipping the remaining iterations does not change any result.
template <class M>
DDGI_D bool march_escaped(const M& m, const SceneK& S)
{
    const int x = static_cast<int>(ceilf(m.p.x)), y = static_cast<int>(ceilf(m.p.y)), z = static_cast<int>(ceilf(m.p.z));
    const unsigned fe = S.face_empty;
    bool out = false;
    out |= (x < S.lo[0]) && (m.dn.x <= 0.0f) && (fe & 1u);
    out |= (x > S.hi[0]) && (m.dn.x >= 0.0f) && (fe & 2u);
    out |= (y < S.lo[1]) && (m.dn.y <= 0.0f) && (fe & 4u);
    out |= (y > S.hi[1]) && (m.dn.y >= 0.0f) && (fe & 8u);
    out |= (z < S.lo[2]) && (m.dn.z <= 0.0f) && (fe & 16u);
    out |= (z > S.hi[2]) && (m.dn.z >= 0.0f) && (fe & 32u);
    return out;
}

// calculate_random_dir_hemisphere (probe_pass.comp:147-178)
DDGI_D f3 hemisphere_dir(f3 n, uint32_t& rng)
{
    const float kTwoPi = 6.2831853071795864769252867665590057683943f;
    const float kSqrtThird = 0.5773502691896257645091487805019574556476f;
    // rng_next is 0 or in [2^-32, 1], 1 - up * up is 0 or in [2^-24, 1]: inside pm::sqrt_core's exact range, no range test
    const float up = pm::sqrt_core(rng_next(rng));
    const float over = pm::sqrt_core(1.0f - up * up);
    const float around = rng_next(rng) * kTwoPi;
    f3 other;
    if (fabsf(n.x) < kSqrtThird) other = mk3(1, 0, 0);
    else if (fabsf(n.y) < kSqrtThird) other = mk3(0, 1, 0);
    else other = mk3(0, 0, 1);
    // For a unit axis normal both cross products are unit axis vectors (exact products of 0 and +-1)
    // and normalize() of one is the identity (x * (1/sqrt(1))): skip the two square roots and divisions.
    const bool axis = (fabsf(n.x) + fabsf(n.y) + fabsf(n.z) == 1.0f) && (fabsf(n.x) == 1.0f || fabsf(n.y) == 1.0f || fabsf(n.z) == 1.0f);
    f3 p1 = cross3(n, other);
    if (!axis) p1 = normalize3(p1);
    f3 p2 = cross3(n, p1);
    if (!axis) p2 = normalize3(p2);
    float sn, cs;
    pm::sincos_small(around, sn, cs);  // P6b
    const float ca = cs * over, sa = sn * over;
    return (n * up + p1 * ca) + p2 * sa;
}

// rgba8 UNORM pack: clamp to [0,1], *255, round to nearest even; NaN -> 0
DDGI_D uint32_t unorm8(float x)
{
    if (!(x > 0.0f)) return 0u;
    x = x > 1.0f ? 1.0f : x;
    return static_cast<uint32_t>(rintf(x * 255.0f));
}

// Slab-major probe slot of reference probe index p = y*cx*cz + z*cx + x  ->  (z*cy + y)*cx + x
DDGI_D int slab_slot(const GridK& G, int p)
{
    const int cxz = G.cx * G.cz;
    const int y = p / cxz;
    const int rem = p - y * cxz;
    const int z = rem / G.cx;
    const int x = rem - z * G.cx;
    return (z * G.cy + y) * G.cx + x;
}

// slab_slot of the probe index sy*cx*cz + sz*cx + sx that get_diffuse_gi makes of a cage corner's grid coordinates
// (intersection.glsl:1336-1340).  Inside the grid that is (sz*cy + sy)*cx + sx with no arithmetic at all; the reference's
// index WRAPS where a coordinate is outside (Q4: x == cx is x = 0 of the next z row, ...), and there the index is decoded
// like any other — two integer divisions, ~70 instructions, which every corner of every shading point used to pay.
DDGI_D int slab_slot_of_corner(const GridK& G, int sx, int sy, int sz, int idx)
{
    const bool inside = static_cast<unsigned>(sx) < static_cast<unsigned>(G.cx) && static_cast<unsigned>(sy) < static_cast<unsigned>(G.cy) && static_cast<unsigned>(sz) < static_cast<unsigned>(G.cz);
    int slot = (sz * G.cy + sy) * G.cx + sx;
    if (__builtin_expect(!inside, 0)) slot = slab_slot(G, idx);
    return slot;
}

}  // namespace ddgi
