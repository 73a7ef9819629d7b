// Microbenchmark (GPU box): VALU issue on gfx950 as a function of the independent fma chains per wave (ILP) and the waves per
// SIMD (TLP).  The trace kernel runs 4 waves per SIMD of mostly dependent code: is it bound by the VALU's issue rate (then only
// fewer instructions help) or by the latency of dependent instructions (then independent work interleaved into a wave is free)?
// hipcc --offload-arch=gfx950 -O3 valu_latency.hip -o valu_latency.bin 2>/dev/null && ./valu_latency.bin
#include <hip/hip_runtime.h>
#include <cstdio>

template <int CHAINS>
__global__ __launch_bounds__(256) void k(float* out, const float* in, int iters)
{
    const int tid = threadIdx.x;
    float acc[CHAINS];
    for (int q = 0; q < CHAINS; ++q) acc[q] = in[(tid + q) & 1023];
    const float w = in[(tid + 40) & 1023], c = in[(tid + 41) & 1023];
    for (int it = 0; it < iters; ++it)
    {
#pragma unroll
        for (int rep = 0; rep < 16 / CHAINS; ++rep)
#pragma unroll
            for (int q = 0; q < CHAINS; ++q) acc[q] = fmaf(acc[q], w, c);
    }
    float r = 0;
    for (int q = 0; q < CHAINS; ++q) r += acc[q];
    out[blockIdx.x * 256 + tid] = r;
}

template <int CHAINS>
float run(float* out, float* in, int iters, int blocks_per_cu)
{
    hipEvent_t a, b;
    (void)hipEventCreate(&a), (void)hipEventCreate(&b);
    // 256 threads = one wave per SIMD; dynamic LDS sized so that exactly blocks_per_cu blocks are resident per CU
    const size_t lds = 160 * 1024 / blocks_per_cu - 1024;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k<CHAINS>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    hipLaunchKernelGGL(k<CHAINS>, dim3(256 * blocks_per_cu), dim3(256), lds, 0, out, in, iters);
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(k<CHAINS>, dim3(256 * blocks_per_cu), dim3(256), lds, 0, out, in, iters);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms;
    (void)hipEventElapsedTime(&ms, a, b);
    return ms;
}

int main()
{
    float *out, *in;
    (void)hipMalloc(&out, 256 * 16 * 256 * 4);
    (void)hipMalloc(&in, 4096);
    (void)hipMemset(in, 0, 4096);
    const int iters = 2048;
    printf("cycles per wave64 v_fma_f32 and SIMD (2.4 GHz), by independent chains per wave x waves per SIMD\n");
    printf("%8s %10s %10s %10s %10s\n", "chains", "1 wave", "2 waves", "4 waves", "8 waves");
    for (int ci = 0; ci < 5; ++ci)
    {
        const int chains = 1 << ci;
        printf("%8d", chains);
        for (int w : {1, 2, 4, 8})
        {
            float ms = 0;
            if (chains == 1) ms = run<1>(out, in, iters, w);
            if (chains == 2) ms = run<2>(out, in, iters, w);
            if (chains == 4) ms = run<4>(out, in, iters, w);
            if (chains == 8) ms = run<8>(out, in, iters, w);
            if (chains == 16) ms = run<16>(out, in, iters, w);
            const double insts_per_simd = static_cast<double>(w) * iters * 16;
            printf(" %10.2f", ms * 1e-3 * 2.4e9 / insts_per_simd);
        }
        printf("\n");
    }
    return 0;
}
