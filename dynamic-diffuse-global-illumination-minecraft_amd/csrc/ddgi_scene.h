// ddgi_scene.h — the reference's procedural voxel scenes in the engine's pinned binary32
// arithmetic (DESIGN.md "Arithmetic pinning"), usable from host (scene bake) and device (hit
// shading).  What each function computes is defined by the cited reference lines
// (assets/shaders/intersection.glsl unless noted); how it is evaluated is this engine's.
#pragma once

#include "ddgi_pinned_math.h"

namespace ddgi {

struct f3
{
    float x, y, z;
};
struct f2
{
    float x, y;
};

DDGI_HD f3 mk3(float x, float y, float z) { return f3{x, y, z}; }
DDGI_HD f3 operator+(f3 a, f3 b) { return f3{a.x + b.x, a.y + b.y, a.z + b.z}; }
DDGI_HD f3 operator-(f3 a, f3 b) { return f3{a.x - b.x, a.y - b.y, a.z - b.z}; }
DDGI_HD f3 operator*(f3 a, f3 b) { return f3{a.x * b.x, a.y * b.y, a.z * b.z}; }
DDGI_HD f3 operator*(f3 a, float s) { return f3{a.x * s, a.y * s, a.z * s}; }
DDGI_HD f3 div3(f3 a, float s) { return f3{a.x / s, a.y / s, a.z / s}; }  // IEEE division

// GLSL min/max/clamp/sign/mix with the operand order the spec gives (defines NaN behaviour)
DDGI_HD float gl_max(float x, float y) { return x < y ? y : x; }
DDGI_HD float gl_min(float x, float y) { return y < x ? y : x; }
DDGI_HD float gl_clamp(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }
DDGI_HD float gl_sign(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }
DDGI_HD float gl_mix(float a, float b, float t) { return a * (1.0f - t) + b * t; }
DDGI_HD f3 gl_mix3(f3 a, f3 b, float t) { return f3{gl_mix(a.x, b.x, t), gl_mix(a.y, b.y, t), gl_mix(a.z, b.z, t)}; }
DDGI_HD float gl_mod(float x, float y) { return x - y * floorf(x / y); }

// int(x): truncation, NaN -> 0, saturating — exactly what v_cvt_i32_f32 does, so the device takes the instruction itself: written
// out (C++'s conversion is undefined outside int's range) the compiler makes three nested exec-mask branches of it, ~16
// instructions apiece, and the noise functions convert a lattice coordinate 94 times over the event code (11 per cave-wall hit)
DDGI_HD int gl_int(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    int r;
    asm("v_cvt_i32_f32 %0, %1" : "=v"(r) : "v"(x));
    return r;
#endif
    if (x != x) return 0;
    if (x >= 2147483648.0f) return 2147483647;
    if (x <= -2147483648.0f) return -2147483647 - 1;
    return static_cast<int>(x);
}

// P7: fract(x) = min(x - floor(x), 1 - 2^-24)   (v_fract_f32)
DDGI_HD float gl_fract(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_fractf(x);
#else
    float f = x - floorf(x);
    return f >= 1.0f ? 0x1.fffffep-1f : f;
#endif
}

// P2: dot products are fma chains
DDGI_HD float dot3(f3 a, f3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
DDGI_HD float dot2(f2 a, f2 b) { return fmaf(a.y, b.y, a.x * b.x); }
DDGI_HD float length3(f3 a) { return sqrtf(dot3(a, a)); }
DDGI_HD float length2(f2 a) { return sqrtf(dot2(a, a)); }
// P3: normalize(v) = v * (1 / sqrt(dot(v,v)))
DDGI_HD f3 normalize3(f3 a)
{
    const float inv = 1.0f / sqrtf(dot3(a, a));
    return a * inv;
}
DDGI_HD f2 normalize2(f2 a)
{
    const float inv = 1.0f / sqrtf(dot2(a, a));
    return f2{a.x * inv, a.y * inv};
}
// P4: a point on a ray is one fma per component
DDGI_HD f3 ray_at(f3 o, f3 d, float t) { return f3{fmaf(d.x, t, o.x), fmaf(d.y, t, o.y), fmaf(d.z, t, o.z)}; }
DDGI_HD f3 cross3(f3 a, f3 b)
{
    return f3{a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y};
}

// ---- noise (intersection.glsl:400-499) --------------------------------------------------------

DDGI_HD float hash_sin(float x) { return pm::sinf_pinned(x); }
// Hash arguments reach 1e6..1e8 where one ulp of the argument is a different sine altogether, so
// their dot products are evaluated literally (two products, plain adds), not as fma chains.
DDGI_HD float hash_dot3(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
DDGI_HD float hash_dot2(f2 a, f2 b) { return a.x * b.x + a.y * b.y; }

DDGI_HD float random1(f3 p)  // :400
{
    return gl_fract(hash_sin(hash_dot3(p, mk3(127.1f, 311.7f, 191.999f))) * 43758.5453f);
}
struct NoiseLut;
DDGI_HD float random1_at(f3 cell, const NoiseLut& L);
DDGI_HD float noise2D(float px, float py)  // :402
{
    return gl_fract(hash_sin(hash_dot2(f2{px, py}, f2{127.1f, 311.7f})) * 43758.5453f);
}
// Memoised lattice hashes.  Every hash above is a pure function of INTEGER lattice coordinates, so
// the host evaluates it once over the rectangle of lattice points the scenes actually touch
// (ddgi_host.cpp: build_noise_lut) and the kernels load the stored binary32 value instead of
// re-evaluating a binary64 sine; outside the rectangle (or with null tables) the hash is computed.
// A stored value IS the function's value, so results are unchanged bit for bit.
// The tables' extents are compile-time constants (ddgi_host.cpp: build_noise_lut fills exactly these): the
// kernels compare against immediates instead of carrying two dozen scalars per launch.
namespace lut {
// noise2D(ix, iy): ix in [x0, x0+nx), iy in [y0, y0+ny).  Wide enough for the cave's most expensive albedo as well: the mushroom
// stems' fbm(5 u, z) reaches ix = 5 * 256 and |iy| = 22 * 256 in its last octave (stems stand at |z| <= 22); 60 MB, of which
// the cave's other block types touch the few rows they always did
constexpr int kN2X0 = -2, kN2NX = 1288, kN2Y0 = -5888, kN2NY = 11776;
constexpr int kN1I0 = -8192, kN1N = 16384;                           // noise1(i)
constexpr int kWpC0 = -16, kWpN = 32;                                // worley_point(cx, cy)
// random1(cell) over [-80, 80) x [-48, 48) x [-80, 80) voxel ids (9.8 MB): the cave's bake box (-42..32, -21..18, -38..31) and
// the rock around it as far as BASELINE's largest grid reaches (C5: 128 x 64 x 128 probes, spacing 1).  A probe grid that
// reaches past the box hits wall voxels out there — 0.7 % of C3's wall hits, 3 % of C5's — and ONE such lane sends its
// whole 64-lane group through the out-of-line albedo.
constexpr int kR1Lo0 = -80, kR1Lo1 = -48, kR1Lo2 = -80;
constexpr int kR1N0 = 160, kR1N1 = 96, kR1N2 = 160;
}  // namespace lut

struct NoiseLut
{
    const float* n2 = nullptr;    // noise2D, index (ix-x0)*ny + (iy-y0)
    const float* n1 = nullptr;    // noise1, index i - i0
    const float* wp = nullptr;    // worley_point, 2 floats each, index ((cx-c0)*n + (cy-c0))*2
    // cave wall: fbm2(kWallFbmX, y) has a constant x, so each octave's two x-interpolations
    // mix(noise2D(ix,iy), noise2D(ix+1,iy), fract(x*freq)) depend on iy alone:
    // wall[o*ny + (iy-y0)] for octave o = 0..7 (freq 2..256) holds that value
    const float* wall = nullptr;
    const float* r1 = nullptr;    // random1(cell), x fastest
};
// table[i] with the byte offset computed in 32 bits (every table is far below 4 GiB): on the device the load then takes the
// table's base from scalar registers and a 32-bit offset per lane (global_load ... v, s[base:base+1]) instead of a 64-bit
// address per lane made by v_lshl_add_u64 / v_mad_u64_u32 — VALU instructions, in the kernel the VALU bounds.
// (DDGI_EXP_LUT_HOT: timing experiment, WRONG colours — every lookup lands in the table's first kilobyte, i.e. in cache: what the tables' misses cost)
#ifndef DDGI_EXP_LUT_HOT
#define DDGI_EXP_LUT_HOT 0
#endif
DDGI_HD float lut_f32(const float* table, unsigned i)
{
    if (DDGI_EXP_LUT_HOT) i &= 255u;
    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(table) + (i << 2));
}
DDGI_HD f2 lut_f32x2(const float* table, unsigned i)  // table[i], table[i + 1] (i need not be even: a dwordx2 load takes any 4-byte alignment)
{
    if (DDGI_EXP_LUT_HOT) i &= 255u;
    const float* q = reinterpret_cast<const float*>(reinterpret_cast<const char*>(table) + (i << 2));
    return f2{q[0], q[1]};
}
constexpr float kWallFbmX = 0.05f;
DDGI_HD float random1_at(f3 cell, const NoiseLut& L)  // random1 of a voxel id (integer-valued floats)
{
    const unsigned ux = static_cast<unsigned>(gl_int(cell.x) - lut::kR1Lo0), uy = static_cast<unsigned>(gl_int(cell.y) - lut::kR1Lo1),
                   uz = static_cast<unsigned>(gl_int(cell.z) - lut::kR1Lo2);
    if (L.r1 && ux < static_cast<unsigned>(lut::kR1N0) && uy < static_cast<unsigned>(lut::kR1N1) && uz < static_cast<unsigned>(lut::kR1N2))
        return lut_f32(L.r1, (uz * static_cast<unsigned>(lut::kR1N1) + uy) * static_cast<unsigned>(lut::kR1N0) + ux);
    return random1(cell);
}

DDGI_HD float interp_noise2D(float x, float y, const NoiseLut& L = NoiseLut())  // :404-419
{
    const float fx0 = floorf(x), fy0 = floorf(y);
    const int ix = gl_int(fx0), iy = gl_int(fy0);
    const float tx = gl_fract(x), ty = gl_fract(y);
    float a, b, c, d;
    const unsigned ux = static_cast<unsigned>(ix - lut::kN2X0), uy = static_cast<unsigned>(iy - lut::kN2Y0);
    if (L.n2 && ux < static_cast<unsigned>(lut::kN2NX - 1) && uy < static_cast<unsigned>(lut::kN2NY - 1))
    {
        const unsigned at = ux * static_cast<unsigned>(lut::kN2NY) + uy;  // (< 2^24)
        unsigned at_next = at + static_cast<unsigned>(lut::kN2NY);
#if defined(__HIP_DEVICE_COMPILE__)
        asm("" : "+v"(at_next));  // (its own 32-bit offset: seen as `at` + 47 104 bytes, past a load's immediate, the compiler goes back to a 64-bit address per lane)
#endif
        const f2 ac = lut_f32x2(L.n2, at), bd = lut_f32x2(L.n2, at_next);
        a = ac.x, c = ac.y, b = bd.x, d = bd.y;
    }
    else
    {
        const float x0 = static_cast<float>(ix), x1 = static_cast<float>(ix + 1);
        const float y0 = static_cast<float>(iy), y1 = static_cast<float>(iy + 1);
        a = noise2D(x0, y0), b = noise2D(x1, y0), c = noise2D(x0, y1), d = noise2D(x1, y1);
    }
    return gl_mix(gl_mix(a, b, tx), gl_mix(c, d, tx), ty);
}
// :421-435 — freq = 2^i, amp = 2^-i for i = 1..8 (P8: exact powers of two)
DDGI_HD float fbm2(float x, float y, const NoiseLut& L = NoiseLut())
{
    float total = 0.0f;
    float freq = 1.0f, amp = 1.0f;
    for (int i = 1; i <= 8; ++i)
    {
        freq *= 2.0f;
        amp *= 0.5f;
        total += interp_noise2D(x * freq, y * freq, L) * amp;
    }
    return total;
}
// fbm2(kWallFbmX, y, L): the same operations in the same order, with the x-interpolations of every
// octave read from L.wall (each entry was computed by the very expression it replaces)
DDGI_HD float fbm2_wall(float y, const NoiseLut& L = NoiseLut())
{
    if (L.wall && fabsf(y) < 7.9f)  // 7.9 * 256 < 2047: every octave's iy, iy+1 are inside the table
    {
        float total = 0.0f;
        float freq = 1.0f, amp = 1.0f;
        unsigned row = static_cast<unsigned>(-lut::kN2Y0);
        for (int i = 1; i <= 8; ++i)
        {
            freq *= 2.0f;
            amp *= 0.5f;
            const float yo = y * freq;
            const f2 q = lut_f32x2(L.wall, row + static_cast<unsigned>(gl_int(floorf(yo))));  // (|iy| < 2047: the sum is a small positive index)
            total += gl_mix(q.x, q.y, gl_fract(yo)) * amp;
            row += static_cast<unsigned>(lut::kN2NY);
        }
        return total;
    }
    return fbm2(kWallFbmX, y, L);
}
DDGI_HD float noise1(float i) { return gl_fract(hash_sin(203.311f * i)); }  // :437-439 (.x only)
DDGI_HD float noise1_at(float i, const NoiseLut& L)
{
    const unsigned u = static_cast<unsigned>(gl_int(i) - lut::kN1I0);
    if (L.n1 && u < static_cast<unsigned>(lut::kN1N)) return lut_f32(L.n1, u);
    return noise1(i);
}
DDGI_HD float interp_noise1D(float x, const NoiseLut& L = NoiseLut())  // :441-448
{
    const float i0 = floorf(x);
    return gl_mix(noise1_at(i0, L), noise1_at(i0 + 1.0f, L), gl_fract(x));
}
DDGI_HD float fbm1(float x, const NoiseLut& L = NoiseLut())  // :450-463 — i = 0..7
{
    float total = 0.0f;
    float freq = 1.0f, amp = 1.0f;
    for (int i = 0; i < 8; ++i)
    {
        total += interp_noise1D(x * freq, L) * amp;
        freq *= 2.0f;
        amp *= 0.5f;
    }
    return total;
}
DDGI_HD f2 worley_point_eval(f2 cell)  // generate_point :467-471 (cell_size 5)
{
    const float a = hash_sin(hash_dot2(cell, f2{127.1f, 311.7f}));
    const float b = hash_sin(hash_dot2(cell, f2{269.5f, 183.3f}) * 43758.5453f);
    return f2{(cell.x + gl_fract(a)) * 5.0f, (cell.y + gl_fract(b)) * 5.0f};
}
DDGI_HD f2 worley_point(f2 cell, const NoiseLut& L)
{
    const unsigned ux = static_cast<unsigned>(gl_int(cell.x) - lut::kWpC0), uy = static_cast<unsigned>(gl_int(cell.y) - lut::kWpC0);
    if (L.wp && ux < static_cast<unsigned>(lut::kWpN) && uy < static_cast<unsigned>(lut::kWpN))
    {
        return lut_f32x2(L.wp, (ux * static_cast<unsigned>(lut::kWpN) + uy) * 2u);
    }
    return worley_point_eval(cell);
}
DDGI_HD float worley(f2 pixel, const NoiseLut& L = NoiseLut())  // :473-499
{
    const f2 cell{floorf(pixel.x / 5.0f), floorf(pixel.y / 5.0f)};
    f2 q = worley_point(cell, L);
    float best = length2(f2{pixel.x - q.x, pixel.y - q.y});
    for (int i = -1; i <= 1; ++i)
    {
        const float cxn = cell.x + static_cast<float>(i);
        for (int j = -1; j <= 1; ++j)
        {
            q = worley_point(f2{cxn, cell.y + static_cast<float>(j)}, L);
            const float d = length2(f2{pixel.x - q.x, pixel.y - q.y});
            if (d < best) best = d;
        }
    }
    return best / 5.0f;
}

// ---- block types (getBlockAt :699-826 and the mushroom SDFs :538-697) -------------------------
// Evaluated on the HOST once per configuration (scene bake); the kernels traverse the bake.

DDGI_HD float sd_round_box(f3 p, f3 b, float r)  // :538-542
{
    const f3 q{fabsf(p.x) - b.x, fabsf(p.y) - b.y, fabsf(p.z) - b.z};
    const f3 qp{gl_max(q.x, 0.0f), gl_max(q.y, 0.0f), gl_max(q.z, 0.0f)};
    return length3(qp) + gl_min(gl_max(q.x, gl_max(q.y, q.z)), 0.0f) - r;
}

// cap: half extents of the cap box, round: its rounding radius; t_up/t_mid/t_dn: block type of
// the cap above / at / below the stem top.  (tiny :544, small :554, medium :572, large :596)
DDGI_HD int mushroom_tiny(f3 p)
{
    if (sd_round_box(p, mk3(1.0f, 0.5f, 1.0f), 0.0f) <= 0.0f) return 7;
    return (p.x == 0.0f && p.z == 0.0f && p.y < 0.0f) ? 9 : 0;
}
DDGI_HD int mushroom_small(f3 p)
{
    if (sd_round_box(p, mk3(1.0f, 0.5f, 1.0f), 1.0f) <= 0.0f)
    {
        if (p.y > 0.0f) return 8;
        if (p.y == 0.0f) return 7;
        if (p.y < 0.0f) return 6;
    }
    return (p.x == 0.0f && p.z == 0.0f && p.y < 0.0f) ? 9 : 0;
}
DDGI_HD int mushroom_medium(f3 p)
{
    if (sd_round_box(p, mk3(2.0f, 0.5f, 2.0f), 1.0f) <= 0.0f)
    {
        if (p.y > 0.0f) return 6;
        if (p.y == 0.0f) return 7;
        if (p.y < 0.0f) return 8;
    }
    if (p.z != 0.0f) return 0;
    if (p.x == 0.0f && p.y < 0.0f && p.y > -7.0f) return 9;
    if (p.x == 1.0f && p.y < -5.0f && p.y > -12.0f) return 9;
    if (p.x == 2.0f && p.y < -10.0f) return 9;
    return 0;
}
DDGI_HD int mushroom_large(f3 p, float dir)
{
    if (sd_round_box(p, mk3(3.0f, 0.5f, 3.0f), 1.5f) <= 0.0f)
    {
        if (p.y > 0.0f) return 6;
        if (p.y == 0.0f) return 8;
        if (p.y < 0.0f) return 7;
    }
    if (p.x != 0.0f) return 0;
    if (p.z == 0.0f && p.y < 0.0f && p.y > -9.0f) return 9;
    if (p.z == dir && p.y < -7.0f && p.y > -18.0f) return 9;
    if (p.z == 2.0f * dir && p.y < -16.0f) return 9;
    return 0;
}

// all_mushrooms :630-697 — a decision tree over (x,z) quadrants selecting which mushroom(s) to test
DDGI_HD int cave_mushrooms(f3 c)
{
    if (c.x < 0.0f && c.z > 0.0f)
    {
        if (c.x < -16.0f)
        {
            if (c.z > 20.0f) return mushroom_tiny(c - mk3(-19, -12, 22));
            if (c.z < 4.0f) return mushroom_tiny(c - mk3(-18, -12, 2));
            const int big = mushroom_large(c - mk3(-22, 3, 8), -1.0f);
            if (big) return big;
            return mushroom_medium(c - mk3(-27, -4, 16));
        }
        if (c.z > 10.0f && c.x > -6.0f) return mushroom_tiny(c - mk3(-4, -14, 12));
        if (c.z < 14.0f) return mushroom_medium(c - mk3(-4, -1, 6));
        return mushroom_small(c - mk3(-10, -8, 18));
    }
    if (c.x < 0.0f && c.z < 0.0f)
    {
        if (c.x < -16.0f)
        {
            if (c.x < -28.0f)
                return (c.z < -16.0f) ? mushroom_tiny(c - mk3(-32, -14, -20)) : mushroom_tiny(c - mk3(-30, -12, -12));
            if (c.z > -10.0f) return mushroom_small(c - mk3(-25, -7, -4));
            return mushroom_medium(c - mk3(-20, -3, -20));
        }
        if (c.x < -12.0f && c.z > -12.0f) return mushroom_tiny(c - mk3(-14, -15, -10));
        if (c.z > -10.0f && c.x > -4.0f) return mushroom_tiny(c - mk3(-2, -12, -2));
        if (c.z < -10.0f) return mushroom_small(c - mk3(-5, -9, -14));
        return mushroom_large(c - mk3(-8, 8, -6), 1.0f);
    }
    if (c.x > 0.0f && c.z < 0.0f)
    {
        if (c.z > -5.0f) return mushroom_tiny(c - mk3(6, -14, -3));
        if (c.z < -14.0f)
            return (c.x > 18.0f) ? mushroom_tiny(c - mk3(20, -7, -16)) : mushroom_large(c - mk3(14, 10, -20), -1.0f);
        return mushroom_medium(c - mk3(6, -6, -10));
    }
    return 0;
}

DDGI_HD bool outside_sphere(f3 c, f3 centre, float r) { return length3(c - centre) - r > 0.0f; }

// Block type 0..13 of the voxel whose id (= ceil of a point inside it, Q5) is c.
DDGI_HD int block_at(f3 c, int scene)
{
    if (scene == 0)  // cave :720-756
    {
        if (c.y > 17.0f) return 0;
        if (c.y < -15.0f)
        {
            if (c.y < -18.0f)
            {
                const float moss = fbm2(c.x * 0.3f, c.z * 0.3f);
                if (gl_int(floorf(moss * 2.0f)) == 0) return 12;
            }
            const float h = fbm2(c.x * 0.058f, c.z * 0.058f);
            const int d = gl_int(floorf(h * 5.0f));
            if (static_cast<float>(-21 + d) >= c.y) return (c.y == -18.0f) ? 13 : 11;
        }
        // hollow = union of four spheres; c + v in the GLSL is c - (-v) here
        if (outside_sphere(c, mk3(0, 0, 0), 20.0f) && outside_sphere(c, mk3(-16, -8, 10), 20.0f) &&
            outside_sphere(c, mk3(13, 1, -19), 18.0f) && outside_sphere(c, mk3(-20, -15, -15), 21.0f))
            return 10;
        return cave_mushrooms(c);
    }
    if (scene == 1)  // Cornell box :758-791
    {
        const bool in_y = fabsf(c.y) < 10.0f, in_z = fabsf(c.z - 15.0f) < 10.0f, in_x = fabsf(c.x) < 10.0f;
        if (c.x == -10.0f && in_y && in_z) return 2;
        if (c.x == 10.0f && in_y && in_z) return 3;
        if (fabsf(c.y) == 10.0f && in_x && in_z) return 5;
        if (c.z == 25.0f && in_x && in_y) return 5;
        if (fabsf(c.x + 3.0f) < 3.0f && fabsf(c.y + 7.0f) < 3.0f && fabsf(c.z - 13.0f) < 3.0f) return 5;
        if (fabsf(c.x - 4.0f) < 3.0f && fabsf(c.y + 4.0f) < 6.0f && fabsf(c.z - 16.0f) < 3.0f) return 5;
        return 0;
    }
    if (scene == 2)  // house :793-820
    {
        if (c.y == -5.0f) return 1;
        if (fabsf(c.x) == 25.0f && fabsf(c.y) < 5.0f && fabsf(c.z) < 15.0f) return 2;
        if (c.y == 5.0f && fabsf(c.x) < 25.0f && fabsf(c.z) < 15.0f) return 5;
        if (c.z == -15.0f && fabsf(c.x) < 25.0f && fabsf(c.y) < 5.0f) return 3;
        if (c.z == 15.0f)
        {
            if (fabsf(c.x - 10.0f) < 2.0f && fabsf(c.y + 1.0f) < 4.0f) return 0;
            if (fabsf(c.x) < 25.0f && fabsf(c.y) < 5.0f) return 3;
        }
        return 0;
    }
    return 0;
}

// ---- hit albedo (getUVs :828-863, dotsPattern :865-870, getColorAt :872-1047) -----------------

// Face parametrisation: which two coordinates run along the face, and which of them is mirrored.
DDGI_HD f2 face_uv(f3 p, f3 n)
{
    const float fx = p.x - floorf(p.x), fy = p.y - floorf(p.y), fz = p.z - floorf(p.z);
    const float cxr = ceilf(p.x) - p.x, czr = ceilf(p.z) - p.z;
    if (n.y == 0.0f)
    {
        if (n.x == 0.0f) return f2{gl_sign(n.z) > 0.0f ? cxr : fx, fy};
        return f2{gl_sign(n.x) < 1.0f ? czr : fz, fy};
    }
    return f2{fx, gl_sign(n.y) < 0.0f ? czr : fz};
}

DDGI_HD float dots_pattern(f2 q, float radius, float cell)
{
    const float c = 4.0f * radius * cell;
    const float h = c / 2.0f;
    const f2 w{gl_mod(q.x + h, c) - h, gl_mod(q.y + h, c) - h};
    return length2(w) - radius;
}

DDGI_HD f3 cell_id(f3 p) { return f3{ceilf(p.x), ceilf(p.y), ceilf(p.z)}; }

DDGI_HD f3 block_albedo(f3 p, int type, f3 n, const NoiseLut& L = NoiseLut())
{
    switch (type)
    {
        case 1:  // house floor: quadrant colours (:889-907; the random1 draw is overwritten by 0.3)
        {
            const float r = 0.3f;
            if (p.x < 0.0f && p.z > 0.0f) return (p.x < -16.0f) ? mk3(0.8f, 0.4f, 0.2f) : mk3(0.1f, r, 0.2f);
            if (p.x < 0.0f && p.z < 0.0f) return (p.x < -16.0f) ? mk3(0.4f, 0.8f, 0.2f) : mk3(0.99f, r, r);
            if (p.x > 0.0f && p.z < 0.0f) return mk3(0.1f, r, 0.5f);
            return mk3(0.99f, r, r);
        }
        case 2: return mk3(0.95f, 0.0f, 0.0f);
        case 3: return mk3(0.0f, 0.95f, 0.0f);
        case 4: return mk3(0.0f, 0.0f, 0.95f);
        case 5: return mk3(0.95f, 0.95f, 0.95f);
        case 6:  // :920-927
            return worley(f2{p.x, p.z}, L) < 0.35f ? mk3(1.0f, 0.0f, 0.223f) : mk3(1.0f, 0.2f, 0.0f);
        case 7:  // :928-936
        {
            const float w = worley(f2{p.x + 5.0f, p.z + 5.0f}, L);
            if (w < 0.25f) return mk3(0.8f - (w * (0.5f - 0.8f)), 1.0f - (w * (0.5f - 1.0f)), 0.0f - (w * (0.5f - 0.0f)));
            return mk3(1.0f, 0.0f, 0.011f);
        }
        case 8:  // :937-953 — rotated dot pattern
        {
            const f2 g = face_uv(p, n);
            const f2 uv{0.707f * g.x + 0.707f * g.y, -0.707f * g.x + 0.707f * g.y};
            const float radius = 0.05f;
            const float alpha = gl_clamp((radius - dots_pattern(uv, radius, 1.8f)) * 100.0f, 0.0f, 1.0f);
            return gl_mix3(mk3(1.0f, 0.313f, 0.0f), mk3(1.0f, 0.0f, 0.223f), alpha);
        }
        case 9:  // :954-963 — stem
        {
            const f2 g = face_uv(p, n);
            float v = fbm2(g.x * 5.0f, p.z, L);
            v += 0.5f * fbm1(p.x, L);
            v = gl_clamp(v, 0.0f, 1.0f);
            return gl_mix3(mk3(0.3f, 0.1f, 0.3f), mk3(0.9f, 0.9f, 0.9f), v);
        }
        case 10:  // :964-1006 — cave wall: height bands mixed with a blue/red gradient in x
        {
            f3 band = mk3(0.568f, 0.133f, 0.439f);
            if (p.y < -8.0f) band = mk3(0.349f, 0.133f, 0.427f);
            else if (p.y < -6.0f) band = mk3(0.568f, 0.133f, 0.439f);
            else if (p.y < -5.0f) band = mk3(0.639f, 0.176f, 0.725f);
            else if (p.y < 0.0f) band = mk3(0.274f, 0.188f, 0.772f);
            else if (p.y < 4.0f) band = mk3(0.341f, 0.270f, 0.768f);
            else if (p.y < 6.0f) band = mk3(0.368f, 0.203f, 0.415f);
            else if (p.y < 11.0f) band = mk3(0.470f, 0.270f, 0.729f);
            const f2 g = face_uv(p, n);
            const float r = fbm2_wall((g.y + p.y) * 0.3f, L);
            const f3 blue = mk3(0.0f, 0.666f, 1.0f), red = mk3(0.294f, 0.007f, 0.152f);
            f3 wall = blue;
            if (p.x < -1.0f) wall = red;
            else if (p.x < 6.0f && p.x >= -1.0f)
                wall = (random1_at(cell_id(p), L) < p.x / 7.0f) ? blue : red;
            return gl_mix3(wall, band, r);
        }
        case 11:  // :1007-1021 — cave ground
        {
            const f3 base = mk3(0.294f, 0.007f, 0.152f);
            float r = random1_at(cell_id(p), L) / 3.0f;
            f3 c = gl_mix3(base, mk3(0.901f, 0.992f, 0.427f), r);
            const f2 g = face_uv(p, n);
            r = fbm2(g.x * 2.0f, g.y * 2.0f, L);
            return gl_mix3(c, base, r / 2.0f);
        }
        case 12:  // :1022-1034 moss, :1035-1046 mold — same pattern, different base colour
        case 13:
        {
            const f2 g = face_uv(p, n);
            const f2 c{g.x - 0.5f, g.y - 0.5f};
            const f2 axis = normalize2(c);
            const float r = interp_noise2D(axis.x, axis.y, L);
            const f3 base = (type == 12) ? mk3(0.356f, 1.0f, 0.101f) : mk3(0.803f, 1.0f, 0.341f);
            return gl_mix3(base, mk3(0.619f, 1.0f, 0.278f), 2.0f * length2(c) + r * 0.3f);
        }
        default: return mk3(0.0f, 0.0f, 0.0f);
    }
}

}  // namespace ddgi
