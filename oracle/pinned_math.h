/*
 * oracle/pinned_math.h — TEST INFRASTRUCTURE ONLY (part of the CPU oracle; never linked into or
 * imported by the product).
 *
 * The oracle's own restatement of the "pinned" elementary functions ("Arithmetic
 * pinning").  GLSL leaves sin/cos/acos precision to the driver (Vulkan: sin/cos abs. error
 * 2^-11 on [-pi,pi]), so the reference is not bit-reproducible across GPUs; this build pins one
 * definition built only from IEEE-754 binary64 +,*,fma,sqrt,rint and one final rounding to
 * binary32, so a CPU and a gfx950 evaluate it to the same bits.  The product carries its own
 * copy (csrc/ddgi_pinned_math.h); tests compare the two bit-for-bit and both against libm.
 *
 *   sin/cos : k = rint(x * 2/pi);  r = fma(-k, PIO2_HI, x);  r = fma(-k, PIO2_LO, r);
 *             Taylor polynomials to r^13 / r^14 on |r| <= pi/4 (truncation < 2e-14), Horner
 *             with fma; quadrant from k mod 4; |x| >= 2^31, inf, nan -> nan.
 *   acos    : fdlibm's published rational approximation R(z) of (asin(x)-x)/x^3 (e_asin.c,
 *             Sun Microsystems 1993; coefficients pS0..pS5, qS1..qS4) evaluated in binary64
 *             without the hi/lo tail (only a binary32 result is needed).
 */
#ifndef ORACLE_PINNED_MATH_H
#define ORACLE_PINNED_MATH_H

#include <math.h>
#include <stdint.h>

static inline void opm_sincos_core(float xf, double* s_out, double* c_out)
{
    const double TWO_OVER_PI = 0x1.45f306dc9c883p-1;
    const double PIO2_HI = 0x1.921fb54442d18p+0;  /* pi/2 rounded to binary64            */
    const double PIO2_LO = 0x1.1a62633145c07p-54; /* pi/2 - PIO2_HI rounded to binary64  */
    double x = (double)xf;
    if (!(fabs(x) < 2147483648.0))
    {
        *s_out = NAN;
        *c_out = NAN;
        return;
    }
    double k = rint(x * TWO_OVER_PI);
    double r = fma(-k, PIO2_HI, x);
    r = fma(-k, PIO2_LO, r);
    double r2 = r * r;
    /* sin r = r + r^3 * (S1 + r2*(S2 + ...)) */
    double ps = 0x1.6124613a86d09p-33;            /*  1/13! */
    ps = fma(ps, r2, -0x1.ae64567f544e4p-26);     /* -1/11! */
    ps = fma(ps, r2, 0x1.71de3a556c734p-19);      /*  1/9!  */
    ps = fma(ps, r2, -0x1.a01a01a01a01ap-13);     /* -1/7!  */
    ps = fma(ps, r2, 0x1.1111111111111p-7);       /*  1/5!  */
    ps = fma(ps, r2, -0x1.5555555555555p-3);      /* -1/3!  */
    double s = fma(r * r2, ps, r);
    /* cos r = 1 + r2 * (C1 + r2*(C2 + ...)) */
    double pc = -0x1.93974a8c07c9dp-37;           /* -1/14! */
    pc = fma(pc, r2, 0x1.1eed8eff8d898p-29);      /*  1/12! */
    pc = fma(pc, r2, -0x1.27e4fb7789f5cp-22);     /* -1/10! */
    pc = fma(pc, r2, 0x1.a01a01a01a01ap-16);      /*  1/8!  */
    pc = fma(pc, r2, -0x1.6c16c16c16c17p-10);     /* -1/6!  */
    pc = fma(pc, r2, 0x1.5555555555555p-5);       /*  1/4!  */
    pc = fma(pc, r2, -0x1.0p-1);                  /* -1/2!  */
    double c = fma(r2, pc, 1.0);
    int q = (int)((int64_t)k & 3);
    switch (q)
    {
        case 0: *s_out = s; *c_out = c; break;
        case 1: *s_out = c; *c_out = -s; break;
        case 2: *s_out = -s; *c_out = -c; break;
        default: *s_out = -c; *c_out = s; break;
    }
}

/* P6b — sine and cosine of a small angle (the hemisphere sample's `around` in [0, 2 pi)): binary32 Cody-Waite
 * reduction + Cephes' sinf/cosf minimax polynomials, every operation one binary32 rounding (fmaf exact-then-round).
 * The product's copy: csrc/ddgi_pinned_math.h sincos_small. */
static inline void opm_sincos_small(float x, float* s_out, float* c_out)
{
    const float k = rintf(x * 0.636619747f);
    float r = fmaf(-k, 1.57079637f, x);
    r = fmaf(-k, -4.37113883e-08f, r);
    const float z = r * r;
    float ps = fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f);
    ps = fmaf(ps, z, -1.6666654611e-1f);
    const float sr = fmaf(r * z, ps, r);
    float pc = fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f);
    pc = fmaf(pc, z, 4.166664568298827e-2f);
    const float cr = fmaf(z * z, pc, fmaf(-0.5f, z, 1.0f));
    const int q = (int)k & 3;
    const float s0 = (q & 1) ? cr : sr, c0 = (q & 1) ? sr : cr;
    *s_out = (q & 2) ? -s0 : s0;
    *c_out = ((q + 1) & 2) ? -c0 : c0;
}

static inline float opm_sinf(float x)
{
    double s, c;
    opm_sincos_core(x, &s, &c);
    return (float)s;
}

static inline float opm_cosf(float x)
{
    double s, c;
    opm_sincos_core(x, &s, &c);
    return (float)c;
}

static inline double opm_asin_R(double z)
{
    const double pS0 = 1.66666666666666657415e-01, pS1 = -3.25565818622400915405e-01,
                 pS2 = 2.01212532134862925881e-01, pS3 = -4.00555345006794114027e-02,
                 pS4 = 7.91534994289814532176e-04, pS5 = 3.47933107596021167570e-05,
                 qS1 = -2.40339491173441421878e+00, qS2 = 2.02094576023350569471e+00,
                 qS3 = -6.88283971605453293030e-01, qS4 = 7.70381505559019352791e-02;
    double p = fma(pS5, z, pS4);
    p = fma(p, z, pS3);
    p = fma(p, z, pS2);
    p = fma(p, z, pS1);
    p = fma(p, z, pS0);
    p = p * z;
    double q = fma(qS4, z, qS3);
    q = fma(q, z, qS2);
    q = fma(q, z, qS1);
    q = fma(q, z, 1.0);
    return p / q;
}

static inline float opm_acosf(float xf)
{
    const double PI_D = 0x1.921fb54442d18p+1;
    const double PIO2_D = 0x1.921fb54442d18p+0;
    double x = (double)xf;
    if (!(fabs(x) <= 1.0)) return NAN;
    if (fabs(x) < 0.5)
    {
        double z = x * x;
        double a = fma(x, opm_asin_R(z), x); /* asin(x) */
        return (float)(PIO2_D - a);
    }
    if (x < 0.0)
    {
        double z = (1.0 + x) * 0.5;
        double s = sqrt(z);
        double a = fma(s, opm_asin_R(z), s);
        return (float)(PI_D - 2.0 * a);
    }
    {
        double z = (1.0 - x) * 0.5;
        double s = sqrt(z);
        double a = fma(s, opm_asin_R(z), s);
        return (float)(2.0 * a);
    }
}

#endif
