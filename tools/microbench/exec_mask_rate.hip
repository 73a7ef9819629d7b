// Microbenchmark (GPU box): does a wave64 VALU instruction on gfx950 cost fewer issue cycles when part of EXEC is zero?
// A wave64 op runs as four 16-lane passes; if the hardware skipped the passes whose lanes are all inactive, packing a thin march
// wave's live lanes into the low lanes would make its steps cheaper.  Independent fma chains (no dependency stalls), four waves per
// SIMD, the same instruction stream under EXEC = all 64 / low 32 / low 16 / every 4th lane (16 lanes spread over all four passes).
// hipcc --offload-arch=gfx950 -O3 exec_mask_rate.hip -o exec_mask_rate.bin && ./exec_mask_rate.bin
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, const float* in, int iters)
{
    const int tid = threadIdx.x, lane = tid & 63;
    float acc[16];
    for (int q = 0; q < 16; ++q) acc[q] = in[(tid + q) & 1023];
    const float w = in[(tid + 40) & 1023], c = in[(tid + 41) & 1023];
    bool on = true;
    if (MODE == 1) on = lane < 32;
    if (MODE == 2) on = lane < 16;
    if (MODE == 3) on = (lane & 3) == 0;
    if (MODE == 4) on = lane < 48;
    if (on)
        for (int it = 0; it < iters; ++it)
        {
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[q] = fmaf(acc[q], w, c);
        }
    float r = 0;
    for (int q = 0; q < 16; ++q) r += acc[q];
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
    const double wave_insts = 256.0 * 8 * 4 * iters * 16;  // wave-instructions
    const char* names[5] = {"exec = all 64 lanes", "exec = lanes 0..31", "exec = lanes 0..15", "exec = every 4th lane (16)", "exec = lanes 0..47"};
    float ms[5] = {run<0>(out, in, iters), run<1>(out, in, iters), run<2>(out, in, iters), run<3>(out, in, iters), run<4>(out, in, iters)};
    for (int m = 0; m < 5; ++m)
        printf("%-28s %.3f ms  %.2f G wave-instructions/s  (%.2f cycles per instruction and SIMD at 2.4 GHz)\n", names[m], ms[m], wave_insts / (ms[m] * 1e-3) / 1e9,
               ms[m] * 1e-3 * 2.4e9 / (wave_insts / 1024.0));
    return 0;
}
