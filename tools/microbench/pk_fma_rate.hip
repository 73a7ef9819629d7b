// Microbenchmark (GPU box): issue rate of v_pk_fma_f32 vs v_fma_f32 on gfx950, with VGPR and SGPR multiplicands.
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off pk_fma_rate.hip -o pk_fma_rate && ./pk_fma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2v __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, const float* in, int iters)
{
    const int tid = threadIdx.x;
    f2v acc[16];
    for (int k = 0; k < 16; ++k) acc[k] = f2v{in[tid + k], in[tid + 16 + k]};
    f2v w = f2v{in[tid + 40], in[tid + 41]};
    const float su = in[blockIdx.x & 1];  // uniform
    const f2v s2 = f2v{in[(blockIdx.x & 1) + 2], in[(blockIdx.x & 1) + 3]};
    for (int it = 0; it < iters; ++it)
    {
#pragma unroll
        for (int k = 0; k < 16; ++k)
        {
            if (MODE == 0) acc[k] = __builtin_elementwise_fma(acc[k], w, w);            // pk_fma, VGPR operands
            if (MODE == 1) { acc[k].x = fmaf(acc[k].x, w.x, w.y); acc[k].y = fmaf(acc[k].y, w.x, w.y); }  // 2 scalar fma
            if (MODE == 2) acc[k] = __builtin_elementwise_fma(s2, f2v{w.x, w.x}, acc[k]);  // pk_fma with an SGPR pair
            if (MODE == 3) { acc[k].x = fmaf(su, w.x, acc[k].x); acc[k].y = fmaf(su, w.y, acc[k].y); }
        }
    }
    float r = 0;
    for (int k = 0; k < 16; ++k) r += acc[k].x + acc[k].y;
    out[blockIdx.x * 256 + tid] = r;
}

template <int MODE>
float run(float* out, float* in, int iters)
{
    hipEvent_t a, b;
    hipEventCreate(&a), hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(256 * 8), dim3(256), 0, 0, out, in, iters);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(256 * 8), dim3(256), 0, 0, out, in, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    return ms;
}

int main()
{
    float *out, *in;
    hipMalloc(&out, 256 * 8 * 256 * 4);
    hipMalloc(&in, 4096);
    hipMemset(in, 0, 4096);
    const int iters = 4096;
    const double fmas = 256.0 * 8 * 256 * iters * 32;  // lane-fmas
    const char* names[4] = {"pk_fma vgpr", "2x fma vgpr", "pk_fma sgpr-pair", "2x fma sgpr"};
    float ms[4] = {run<0>(out, in, iters), run<1>(out, in, iters), run<2>(out, in, iters), run<3>(out, in, iters)};
    for (int m = 0; m < 4; ++m) printf("%-18s %.3f ms  %.1f T lane-fma/s\n", names[m], ms[m], fmas / (ms[m] * 1e-3) / 1e12);
    return 0;
}
