// ddgi_pinned_math.h — the engine's pinned elementary functions, host + device (gfx950).
//
// GLSL lets the driver choose the precision of sin/cos/acos (Vulkan: abs. error 2^-11 for sin/cos
// on [-pi,pi]); the reference's noise hashes fract(sin(x)*43758.5453)
// (assets/shaders/intersection.glsl:400-402,438,469) amplify that freedom into visible
// differences between GPUs.  This engine pins ONE definition that a CPU and a gfx950 evaluate to
// the same bits: binary64 +, *, fma, sqrt, rint only, one final rounding to binary32.
// See DESIGN.md "Arithmetic pinning" (P6).  tests/ compare these with libm (closeness) and with
// the oracle's independent restatement (bit equality).
#pragma once

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#define DDGI_HD __host__ __device__ __forceinline__

namespace ddgi {
namespace pm {

struct SinCos
{
    double s, c;
};

// Reduce x to r in [-pi/4, pi/4] with k = rint(x*2/pi): two fma's against a 106-bit pi/2 give
// |error| < 2^-53 for |x| < 2^31; then 13th/14th-order Taylor polynomials (truncation < 2e-14).
__host__ __device__ __attribute__((noinline)) inline SinCos sincos_core(float xf)
{
    const double kTwoOverPi = 0x1.45f306dc9c883p-1;
    const double kPio2Hi = 0x1.921fb54442d18p+0;
    const double kPio2Lo = 0x1.1a62633145c07p-54;
    SinCos out;
    const double x = static_cast<double>(xf);
    if (!(fabs(x) < 2147483648.0))
    {
        out.s = out.c = __builtin_nan("");
        return out;
    }
    const double k = rint(x * kTwoOverPi);
    double r = fma(-k, kPio2Hi, x);
    r = fma(-k, kPio2Lo, r);
    const double z = r * r;

    double ps = 0x1.6124613a86d09p-33;           //  1/13!
    ps = fma(ps, z, -0x1.ae64567f544e4p-26);     // -1/11!
    ps = fma(ps, z, 0x1.71de3a556c734p-19);      //  1/9!
    ps = fma(ps, z, -0x1.a01a01a01a01ap-13);     // -1/7!
    ps = fma(ps, z, 0x1.1111111111111p-7);       //  1/5!
    ps = fma(ps, z, -0x1.5555555555555p-3);      // -1/3!
    const double sr = fma(r * z, ps, r);

    double pc = -0x1.93974a8c07c9dp-37;          // -1/14!
    pc = fma(pc, z, 0x1.1eed8eff8d898p-29);      //  1/12!
    pc = fma(pc, z, -0x1.27e4fb7789f5cp-22);     // -1/10!
    pc = fma(pc, z, 0x1.a01a01a01a01ap-16);      //  1/8!
    pc = fma(pc, z, -0x1.6c16c16c16c17p-10);     // -1/6!
    pc = fma(pc, z, 0x1.5555555555555p-5);       //  1/4!
    pc = fma(pc, z, -0x1.0p-1);                  // -1/2!
    const double cr = fma(z, pc, 1.0);

    const int q = static_cast<int>(static_cast<long long>(k) & 3);
    const bool swap = (q & 1) != 0;
    const double s0 = swap ? cr : sr;
    const double c0 = swap ? sr : cr;
    out.s = (q & 2) ? -s0 : s0;
    out.c = ((q + 1) & 2) ? -c0 : c0;
    return out;
}

// P6b — sine and cosine of a SMALL angle (the hemisphere sample's `around` in [0, 2 pi), probe_pass.comp:153,176):
// binary32 throughout, every operation written out (fmaf = one rounding), so host, device and the oracle's copy
// (oracle/pinned_math.h: opm_sincos_small) agree bit for bit.  Cody-Waite reduction by k = rint(x * 2/pi) against
// pi/2 split in two binary32 parts, then the degree-7 / degree-8 minimax polynomials of Cephes' sinf/cosf
// (S. Moshier, 1992) on [-pi/4, pi/4].  |error| < 2e-7 for |x| < 64 — the Vulkan spec lets a driver's sin/cos be
// off by 2^-11.  A third of the instructions of sincos_core (binary64, for the hashes' huge arguments).
DDGI_HD void sincos_small(float x, float& s_out, float& c_out)
{
    const float k = rintf(x * 0.636619747f);           // fl(2/pi)
    float r = fmaf(-k, 1.57079637f, x);               // fl(pi/2)
    r = fmaf(-k, -4.37113883e-08f, r);                // fl(pi/2 - fl(pi/2))
    const float z = r * r;
    float ps = fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f);
    ps = fmaf(ps, z, -1.6666654611e-1f);
    const float sr = fmaf(r * z, ps, r);
    float pc = fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f);
    pc = fmaf(pc, z, 4.166664568298827e-2f);
    const float cr = fmaf(z * z, pc, fmaf(-0.5f, z, 1.0f));
    const int q = static_cast<int>(k) & 3;
    const float s0 = (q & 1) ? cr : sr, c0 = (q & 1) ? sr : cr;
    s_out = (q & 2) ? -s0 : s0;
    c_out = ((q + 1) & 2) ? -c0 : c0;
}

DDGI_HD float sinf_pinned(float x) { return static_cast<float>(sincos_core(x).s); }
DDGI_HD float cosf_pinned(float x) { return static_cast<float>(sincos_core(x).c); }

// (asin(x) - x)/x^3 on z = x^2 <= 0.25: the rational approximation published in fdlibm's
// e_asin.c (Sun Microsystems, 1993), numerator pS0..pS5 and denominator 1,qS1..qS4.
DDGI_HD double asin_ratio(double z)
{
    double num = fma(3.47933107596021167570e-05, z, 7.91534994289814532176e-04);
    num = fma(num, z, -4.00555345006794114027e-02);
    num = fma(num, z, 2.01212532134862925881e-01);
    num = fma(num, z, -3.25565818622400915405e-01);
    num = fma(num, z, 1.66666666666666657415e-01);
    num = num * z;
    double den = fma(7.70381505559019352791e-02, z, -6.88283971605453293030e-01);
    den = fma(den, z, 2.02094576023350569471e+00);
    den = fma(den, z, -2.40339491173441421878e+00);
    den = fma(den, z, 1.0);
    return num / den;
}

DDGI_HD float acosf_pinned(float xf)
{
    const double kPi = 0x1.921fb54442d18p+1;
    const double kPio2 = 0x1.921fb54442d18p+0;
    const double x = static_cast<double>(xf);
    if (!(fabs(x) <= 1.0)) return __builtin_nanf("");
    if (fabs(x) < 0.5)
    {
        const double a = fma(x, asin_ratio(x * x), x);
        return static_cast<float>(kPio2 - a);
    }
    const bool neg = x < 0.0;
    const double z = (neg ? (1.0 + x) : (1.0 - x)) * 0.5;
    const double s = sqrt(z);
    const double a = fma(s, asin_ratio(z), s);
    return static_cast<float>(neg ? (kPi - 2.0 * a) : (2.0 * a));
}

// ---- IEEE 1/x and sqrt(x) in fewer instructions, on arguments known to lie in a domain (device) -----------------------
// The compiler's correctly rounded binary32 1/x is 11 instructions and its sqrt 16 (scaling for the ends of the exponent
// range, fix-ups for the special values); a ray segment runs one of them a dozen times (normalisation, the three axis
// reciprocals of a march, the hemisphere sample).  Where the argument's range is known by construction, shorter BRANCH-FREE
// sequences give the same bits (a version that tested the range and branched to the compiler's sequence saved 3.4 % of the
// trace kernel's VALU instructions and lost 1.8 % of its time to the branches' scalar work):
//   sqrt_core(x)      v_sqrt_f32 + the compiler's own +-1 ulp fix            == sqrtf(x)          for x in {+0} u [2^-96, +inf] u NaN
//   rcp_sqrt_core(x)  ... + v_rcp_f32, one Newton step in fma, v_div_fixup   == 1.0f / sqrtf(x)   on the same domain
//   rcp_upto_2p94(x)  the same reciprocal of x * 2^32, times 2^32             == 1.0f / x          for |x| <= 2^94 (zero and
//                     subnormals included), +-inf and NaN
// bit for bit: tests/exact_rcp_sqrt_check.hip walks all 2^32 arguments on the GPU and checks every one that lies in the
// stated domain (tests/test_gpu_device_math.py).  Each use site says why its arguments are in the domain.
__device__ __forceinline__ float sqrt_core(float x)
{
    float s = __builtin_amdgcn_sqrtf(x);
    const float sd = __uint_as_float(__float_as_uint(s) - 1u), su = __uint_as_float(__float_as_uint(s) + 1u);
    const float rd = fmaf(-sd, s, x), ru = fmaf(-su, s, x);
    s = rd <= 0.0f ? sd : s;
    s = ru > 0.0f ? su : s;
    return s;
}
__device__ __forceinline__ float rcp_fixed(float x)  // 1.0f / x for x = 0, 2^-126 <= |x| <= 2^126, +-inf, NaN
{
    const float r = __builtin_amdgcn_rcpf(x);
    return __builtin_amdgcn_div_fixupf(fmaf(fmaf(-x, r, 1.0f), r, r), x, 1.0f);
}
__device__ __forceinline__ float rcp_sqrt_core(float x) { return rcp_fixed(sqrt_core(x)); }  // sqrt_core's range maps into rcp_fixed's
__device__ __forceinline__ float rcp_upto_2p94(float x)
{
    // x * 2^32 is exact and normal for every nonzero |x| <= 2^94 (subnormals included) and lies in rcp_fixed's range; scaling
    // the correctly rounded quotient back by 2^32 is exact again, or overflows exactly where 1.0f / x does
    return rcp_fixed(x * 0x1.0p32f) * 0x1.0p32f;
}


// n / d for many numerators against one denominator, 6 instructions per quotient instead of 11: the compiler's correctly rounded
// division is v_div_scale (x 2), v_rcp_f32 + one Newton step on the scaled denominator, q = n r and two residual corrections (the
// second one in v_div_fmas), v_div_fixup.  Wherever v_div_scale leaves BOTH operands alone (ISA, V_DIV_SCALE_F32: d normal with
// 1/d normal, n / d normal, exponent(n) - exponent(d) < 96, |n| >= 2^-103; there v_div_fmas is a plain fma) the quotient is a
// function of n, d and the refined reciprocal only — and the reciprocal depends on d alone.  Domain used here, with margins:
//   2^-24 <= |d| < 2^25, or +-inf (quotient 0: v_div_fixup);   n = +-0, or 2^-100 <= |n| <= 2^60 (kDivPreparedLo/Hi)
// (exponent differences within (-126, 96); the residuals n - d q are exact even where they are subnormal)
// Same bits as `n / d` there (tests/exact_rcp_sqrt_check.hip: every mantissa of d against numerators across the domain and at
// its edges).  A caller with numerators outside the domain uses `/`.
struct DivBy
{
    float d, r;
};
constexpr uint32_t kDivPreparedLo = 0x0d800000u, kDivPreparedHi = 0x5d800000u;  // 2^-100, 2^60 as bits
__device__ __forceinline__ DivBy div_by(float d)
{
    const float r0 = __builtin_amdgcn_rcpf(d);
    return DivBy{d, fmaf(fmaf(-d, r0, 1.0f), r0, r0)};
}
__device__ __forceinline__ float div_prepared(float n, const DivBy& by)
{
    float q = n * by.r;
    q = fmaf(fmaf(-by.d, q, n), by.r, q);
    q = fmaf(fmaf(-by.d, q, n), by.r, q);
    return __builtin_amdgcn_div_fixupf(q, by.d, n);
}
// Are all of a set of numerators in div_prepared's domain?  Two integer instructions per value: for values >= +0 the bit patterns
// order like the values, and `u - 1` sends zero past every threshold.  A negative, infinite or NaN value counts as outside
// (-0 too: conservative).  tests/exact_rcp_sqrt_check.hip walks all 2^32 bit patterns: whatever passes here divides like `/`.
struct DivDomainCheck
{
    uint32_t hi = 0u, lo = 0xffffffffu;
    __device__ __forceinline__ void add(float v)
    {
        const uint32_t u = __float_as_uint(v);
        hi = hi > u ? hi : u, lo = lo < u - 1u ? lo : u - 1u;
    }
    __device__ __forceinline__ bool outside() const { return hi > kDivPreparedHi || lo < kDivPreparedLo - 1u; }
};

// two numerators against the same denominator: the same operations as v_pk_mul_f32 / v_pk_fma_f32 — half the instructions to
// issue (not half the time in the ALU: the packed forms run at the scalar forms' lane rate; it is the issue slots that are
// scarce where this is used, beside waves that run MFMA chains)
typedef float f2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2v div_prepared2(f2v n, const DivBy& by)
{
    const f2v nd = {-by.d, -by.d}, r = {by.r, by.r};
    f2v q = n * r;
    q = __builtin_elementwise_fma(__builtin_elementwise_fma(nd, q, n), r, q);
    q = __builtin_elementwise_fma(__builtin_elementwise_fma(nd, q, n), r, q);
    return f2v{__builtin_amdgcn_div_fixupf(q.x, by.d, n.x), __builtin_amdgcn_div_fixupf(q.y, by.d, n.y)};
}

}  // namespace pm
}  // namespace ddgi
