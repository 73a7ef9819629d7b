// Exhaustive check, on the GPU, of the short branch-free 1/x and sqrt(x) sequences the kernels use on arguments of known
// range (csrc/ddgi_pinned_math.h: pm::sqrt_core, rcp_sqrt_core, rcp_fixed, rcp_upto_2p94; div_prepared) against the compiler's correctly
// rounded `/` and sqrtf: every one of the 2^32 binary32 arguments that lies in a function's stated domain, bit for bit.
// Prints OK or the mismatch counts.  Run by tests/test_gpu_device_math.py.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

#include "../dynamic-diffuse-global-illumination-minecraft_amd/csrc/ddgi_pinned_math.h"

using namespace ddgi;

__device__ __forceinline__ bool same(float a, float b) { return __float_as_uint(a) == __float_as_uint(b); }
// the references, kept out of line so that nothing of the sequences under test can be folded into them
__device__ __attribute__((noinline)) float ref_rcp(float x) { return 1.0f / x; }
__device__ __attribute__((noinline)) float ref_sqrt(float x) { return sqrtf(x); }

__global__ void k_check(unsigned long long* bad, unsigned long long* checked)
{
    const uint64_t tid = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x, stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    unsigned long long b[4] = {0, 0, 0, 0}, n[4] = {0, 0, 0, 0};
    for (uint64_t u = tid; u < (1ull << 32); u += stride)
    {
        const float x = __uint_as_float(static_cast<uint32_t>(u));
        const float ax = fabsf(x);
        const bool nan = x != x;
        const float want_r = ref_rcp(x), want_s = ref_sqrt(x);
        // sqrt_core / rcp_sqrt_core: {+0} u [2^-96, +inf] u NaN
        if (u == 0u || nan || (x >= 0x1.0p-96f))
        {
            n[0]++, n[1]++;
            if (!same(pm::sqrt_core(x), want_s)) b[0]++;
            if (!same(pm::rcp_sqrt_core(x), ref_rcp(want_s))) b[1]++;
        }
        // rcp_fixed: +-0, 2^-126 <= |x| <= 2^126, +-inf, NaN
        if (ax == 0.0f || nan || ax == __builtin_inff() || (ax >= 0x1.0p-126f && ax <= 0x1.0p126f))
        {
            n[2]++;
            if (!same(pm::rcp_fixed(x), want_r)) b[2]++;
        }
        // rcp_upto_2p94: |x| <= 2^94 (zero, subnormals), +-inf, NaN
        if (nan || ax == __builtin_inff() || ax <= 0x1.0p94f)
        {
            n[3]++;
            if (!same(pm::rcp_upto_2p94(x), want_r)) b[3]++;
        }
    }
    for (int i = 0; i < 4; ++i)
    {
        if (b[i]) atomicAdd(&bad[i], b[i]);
        atomicAdd(&checked[i], n[i]);
    }
}

// pm::div_prepared against `/`: every mantissa of the denominator at a spread of its exponents (and +inf), against numerators
// spread over the domain (pseudo-random mantissas at every exponent of [2^-100, 2^60], the domain's two ends, zero)
__device__ __attribute__((noinline)) float ref_div(float n, float d) { return n / d; }
__global__ void k_check_div(unsigned long long* bad, unsigned long long* checked)
{
    const uint64_t tid = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x, stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    unsigned long long b = 0, n_checked = 0;
    const int d_exps[7] = {-24, -20, -1, 0, 1, 8, 24};
    for (uint64_t m = tid; m < (1ull << 23); m += stride)
        for (int de = 0; de < 8; ++de)
        {
            const float d = de < 7 ? __uint_as_float((static_cast<uint32_t>(127 + d_exps[de]) << 23) | static_cast<uint32_t>(m)) : __builtin_inff();
            const pm::DivBy by = pm::div_by(d);
            uint32_t h = static_cast<uint32_t>(m) * 2654435761u + static_cast<uint32_t>(de);
            for (int k = 0; k < 24; ++k)
            {
                h = h * 1664525u + 1013904223u;
                float n;
                if (k == 0) n = 0.0f;
                else if (k == 1) n = __uint_as_float(pm::kDivPreparedLo);
                else if (k == 2) n = __uint_as_float(pm::kDivPreparedHi);
                else
                {
                    const uint32_t e = 27u + (h >> 9) % 160u;                     // biased exponents 27 .. 186 = 2^-100 .. 2^59
                    n = __uint_as_float((e << 23) | (h & 0x7fffffu));
                }
                if (!same(pm::div_prepared(n, by), ref_div(n, d))) b++;
                const pm::f2v two = pm::div_prepared2(pm::f2v{n, -n}, by);  // the packed form, and negative numerators
                if (!same(two.x, ref_div(n, d)) || !same(two.y, ref_div(-n, d))) b++;
                n_checked += 3;
            }
        }
    if (b) atomicAdd(&bad[0], b);
    atomicAdd(&checked[0], n_checked);
}

// The guard and the division together: every binary32 bit pattern as a numerator.  What pm::DivDomainCheck lets through must be
// zero or lie in [2^-100, 2^60], and must divide like `/` (here: by six divisors of the range the blend kernels use, and +inf);
// what it holds back must be outside.
__global__ void k_check_guard(unsigned long long* bad, unsigned long long* passed)
{
    const uint64_t tid = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x, stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    const float ds[7] = {1.0e-6f, 0.37f, 1.0f, 3.7f, 99.99f, 255.9f, __builtin_inff()};
    pm::DivBy by[7];
    for (int i = 0; i < 7; ++i) by[i] = pm::div_by(ds[i]);
    unsigned long long b = 0, n = 0;
    for (uint64_t u = tid; u < (1ull << 32); u += stride)
    {
        const float v = __uint_as_float(static_cast<uint32_t>(u));
        pm::DivDomainCheck c;
        c.add(v);
        const bool in_domain = (u == 0u) || (v >= 0x1.0p-100f && v <= 0x1.0p60f);
        if (c.outside() == in_domain) b++;
        if (!c.outside())
        {
            n++;
            for (int i = 0; i < 7; ++i)
                if (!same(pm::div_prepared(v, by[i]), ref_div(v, ds[i]))) b++;
        }
    }
    if (b) atomicAdd(&bad[0], b);
    atomicAdd(&passed[0], n);
}

int main()
{
    {
        unsigned long long* dg = nullptr;
        if (hipMalloc(&dg, 2 * sizeof(unsigned long long)) != hipSuccess || hipMemset(dg, 0, 2 * sizeof(unsigned long long)) != hipSuccess) return 2;
        hipLaunchKernelGGL(k_check_guard, dim3(4096), dim3(256), 0, 0, dg, dg + 1);
        unsigned long long hg[2] = {1, 0};
        if (hipMemcpy(hg, dg, sizeof(hg), hipMemcpyDeviceToHost) != hipSuccess) return 2;
        std::printf("%-14s %llu mismatches; %llu of 2^32 numerators pass the guard, each divided by 7 divisors\n", "DivDomainCheck", hg[0], hg[1]);
        if (hg[0] || hg[1] != 160ull * (1ull << 23) + 2ull)   // biased exponents 27 .. 186, every mantissa; +0; 2^60 itself
        {
            std::printf("FAILED\n");
            return 1;
        }
    }
    {
        unsigned long long* dd = nullptr;
        if (hipMalloc(&dd, 2 * sizeof(unsigned long long)) != hipSuccess || hipMemset(dd, 0, 2 * sizeof(unsigned long long)) != hipSuccess) return 2;
        hipLaunchKernelGGL(k_check_div, dim3(4096), dim3(256), 0, 0, dd, dd + 1);
        unsigned long long hd[2] = {1, 0};
        if (hipMemcpy(hd, dd, sizeof(hd), hipMemcpyDeviceToHost) != hipSuccess) return 2;
        std::printf("%-14s %llu mismatches in %llu quotients\n", "div_prepared", hd[0], hd[1]);
        if (hd[0] || hd[1] < (1ull << 30))
        {
            std::printf("FAILED\n");
            return 1;
        }
    }
    unsigned long long* d = nullptr;
    if (hipMalloc(&d, 8 * sizeof(unsigned long long)) != hipSuccess || hipMemset(d, 0, 8 * sizeof(unsigned long long)) != hipSuccess) return 2;
    hipLaunchKernelGGL(k_check, dim3(4096), dim3(256), 0, 0, d, d + 4);
    unsigned long long h[8] = {1, 1, 1, 1, 0, 0, 0, 0};
    if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return 2;
    const char* names[4] = {"sqrt_core", "rcp_sqrt_core", "rcp_fixed", "rcp_upto_2p94"};
    for (int i = 0; i < 4; ++i) std::printf("%-14s %llu mismatches in %llu arguments of its domain\n", names[i], h[i], h[4 + i]);
    const bool ok = !(h[0] | h[1] | h[2] | h[3]) && h[4] > (1ull << 30) && h[6] > (1ull << 31) && h[7] > (1ull << 31);
    std::printf(ok ? "OK\n" : "FAILED\n");
    return ok ? 0 : 1;
}
