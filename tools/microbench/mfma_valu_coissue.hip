// Microbenchmark (GPU box): what a VALU / LDS / division stream of one wave gets done while OTHER waves of the same SIMD run
// dependent chains of v_mfma_f32_32x32x2_f32 (the blend kernels' situation: ddgi_blend_sample.hip, blend_depth_resident).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off mfma_valu_coissue.hip -o mfma_valu_coissue.bin && ./mfma_valu_coissue.bin
// One workgroup of 12 waves on one CU (waves go round the four SIMDs: waves w, w + 4, w + 8 share a SIMD — printed from HW_ID).
// The instruction streams are inline assembly so that the mix is what the label says.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));

#define FMA8(D) \
    "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n" \
    "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
#define DEP8 \
    "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n" \
    "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n"
#define RCP8 \
    "v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n"
#define VREGS "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7])

// VKIND: 0 independent fma, 1 dependent fma, 2 v_rcp, 3 IEEE division (compiler's sequence), 4 LDS reads,
//        5 / 6: the MFMA wave itself issues 8 / 14 independent fmas behind every MFMA (no separate VALU wave)
template <int SHAPE, int VKIND, int PRIO>
__global__ __launch_bounds__(768) void k(uint32_t mfma_mask, uint32_t valu_mask, int mfma_n, int valu_n, float* io, unsigned long long* cycles, uint32_t* hwid)
{
    __shared__ float lds[4096];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 4096; i += 768) lds[i] = io[i & 255];
    float a = io[lane], b = io[lane + 64];
    f16v acc;
    for (int r = 0; r < 16; ++r) acc[r] = io[lane + r];
    f4v acc4 = {io[lane], io[lane + 1], io[lane + 2], io[lane + 3]};
    float v[8];
    for (int r = 0; r < 8; ++r) v[r] = io[lane + 100 + r];
    if (lane == 0) hwid[wave] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));  // HW_REG_HW_ID
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    const bool is_mfma = (mfma_mask >> wave) & 1u, is_valu = (valu_mask >> wave) & 1u;
    if (is_mfma)
    {
        if (VKIND == 5)
            for (int i = 0; i < mfma_n; ++i)
                asm volatile("v_mfma_f32_32x32x2_f32 %10, %8, %9, %10\n" FMA8() : VREGS : "v"(a), "v"(b), "v"(acc));
        else if (VKIND == 6)
            for (int i = 0; i < mfma_n; ++i)
                asm volatile("v_mfma_f32_32x32x2_f32 %10, %8, %9, %10\n" FMA8() "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n"
                             "v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n" : VREGS : "v"(a), "v"(b), "v"(acc));
        else if (SHAPE == 0)
            for (int i = 0; i < mfma_n; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        else
            for (int i = 0; i < mfma_n; ++i) acc4 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc4, 0, 0, 0);
    }
    else if (is_valu)
    {
        if (PRIO) __builtin_amdgcn_s_setprio(3);
        for (int i = 0; i < valu_n; ++i)
        {
            if (VKIND == 0) asm volatile(FMA8() FMA8() FMA8() FMA8() : VREGS : "v"(a), "v"(b));
            if (VKIND == 1) asm volatile(DEP8 DEP8 DEP8 DEP8 : VREGS : "v"(a), "v"(b));
            if (VKIND == 2) asm volatile(RCP8 RCP8 RCP8 RCP8 : VREGS : "v"(a), "v"(b));
            if (VKIND == 3)
#pragma unroll
                for (int r = 0; r < 8; ++r) v[r] = v[r] / a;
            if (VKIND == 4)
            {
#pragma unroll
                for (int r = 0; r < 8; ++r) v[r] += lds[(lane * 33 + r * 64 + i) & 4095];
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = a;
    for (int r = 0; r < 16; ++r) s += acc[r];
    for (int r = 0; r < 8; ++r) s += v[r];
    s += acc4[0] + acc4[1] + acc4[2] + acc4[3];
    io[4096 + threadIdx.x] = s;
    if (lane == 0) cycles[wave] = t1 - t0;
}

static float* io;
static unsigned long long* cyc;
static uint32_t* hw;

template <int SHAPE, int VKIND, int PRIO>
void run(const char* name, uint32_t mm, uint32_t vm, int per_iter, const char* unit)
{
    const int N = 4096, V = 1024;
    unsigned long long h[12];
    uint32_t hid[12];
    for (int rep = 0; rep < 2; ++rep)
    {
        hipLaunchKernelGGL((k<SHAPE, VKIND, PRIO>), dim3(1), dim3(768), 0, 0, mm, vm, N, V, io, cyc, hw);
        (void)hipDeviceSynchronize();
    }
    (void)hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    (void)hipMemcpy(hid, hw, sizeof hid, hipMemcpyDeviceToHost);
    printf("%-44s", name);
    for (int w = 0; w < 12; ++w)
        if (((mm | vm) >> w) & 1u)
        {
            const bool m = (mm >> w) & 1u;
            printf("  w%d@simd%u %s %.1f", w, (hid[w] >> 4) & 3u, m ? "cyc/mfma" : unit, m ? double(h[w]) / N : double(h[w]) / (double(V) * per_iter));
        }
    printf("\n");
}

// The whole chip running f32 MFMA chains: 2 waves per SIMD on every CU.  Reports the time per MFMA per SIMD from HIP events (what a
// kernel pays), and the shader clock it implies if an MFMA is 64 cycles; s_memtime against the 100 MHz s_memrealtime gives
// the frequency of the counter the single-workgroup cases above are quoted in.
__global__ __launch_bounds__(512) void k_chip(int mfma_n, float* io, unsigned long long* stamps)
{
    const int lane = threadIdx.x & 63;
    float a = io[lane], b = io[lane + 64];
    f16v acc;
    for (int r = 0; r < 16; ++r) acc[r] = io[lane + r];
    const unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (int i = 0; i < mfma_n; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    const unsigned long long t1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    float s = 0;
    for (int r = 0; r < 16; ++r) s += acc[r];
    io[4096 + (threadIdx.x & 1023)] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) stamps[0] = t1 - t0, stamps[1] = w1 - w0;
}
static void run_chip(int blocks, const char* name)
{
    const int N = 1 << 16;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k_chip, dim3(blocks), dim3(512), 0, 0, N, io, cyc);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k_chip, dim3(blocks), dim3(512), 0, 0, N, io, cyc);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long st[2];
    (void)hipMemcpy(st, cyc, sizeof st, hipMemcpyDeviceToHost);
    const double ns_per_mfma = ms * 1e6 / (2.0 * N);  // two waves share a SIMD's matrix pipe
    printf("%-28s %.3f ms: %.1f ns per MFMA per SIMD = %.2f GHz if an MFMA is 64 cycles;  s_memtime: %.1f ticks per MFMA pair, %.0f MHz\n", name, ms, ns_per_mfma,
           64.0 / ns_per_mfma, double(st[0]) / N, double(st[0]) / (double(st[1]) / 100.0));
}

int main()
{
    (void)hipMalloc(&io, 65536), (void)hipMalloc(&cyc, 12 * 8), (void)hipMalloc(&hw, 12 * 4);
    (void)hipMemset(io, 0, 65536);
    run<0, 0, 0>("mfma w0 alone", 0x001, 0, 1, "");
    run<0, 0, 0>("mfma w0,w4 (one SIMD)", 0x011, 0, 1, "");
    run<0, 0, 0>("fma-indep w8 alone", 0, 0x100, 32, "cyc/fma");
    run<0, 0, 0>("fma-indep w8 + mfma w0", 0x001, 0x100, 32, "cyc/fma");
    run<0, 0, 0>("fma-indep w8 + mfma w0,w4", 0x011, 0x100, 32, "cyc/fma");
    run<0, 0, 1>("fma-indep w8 prio 3 + mfma w0,w4", 0x011, 0x100, 32, "cyc/fma");
    run<0, 0, 0>("fma-indep w9 (other SIMD) + mfma w0,w4", 0x011, 0x200, 32, "cyc/fma");
    run<0, 0, 0>("fma-indep w4,w8 alone (two VALU waves)", 0, 0x110, 32, "cyc/fma");
    run<0, 1, 0>("fma-dep w8 alone", 0, 0x100, 32, "cyc/fma");
    run<0, 1, 0>("fma-dep w8 + mfma w0,w4", 0x011, 0x100, 32, "cyc/fma");
    run<0, 2, 0>("rcp w8 alone", 0, 0x100, 32, "cyc/rcp");
    run<0, 2, 0>("rcp w8 + mfma w0,w4", 0x011, 0x100, 32, "cyc/rcp");
    run<0, 3, 0>("div w8 alone", 0, 0x100, 8, "cyc/div");
    run<0, 3, 0>("div w8 + mfma w0", 0x001, 0x100, 8, "cyc/div");
    run<0, 3, 0>("div w8 + mfma w0,w4", 0x011, 0x100, 8, "cyc/div");
    run<0, 4, 0>("lds w8 alone", 0, 0x100, 8, "cyc/ds_read");
    run<0, 4, 0>("lds w8 + mfma w0,w4", 0x011, 0x100, 8, "cyc/ds_read");
    run<0, 5, 0>("mfma w0, 8 fma behind each (same wave)", 0x001, 0, 1, "");
    run<0, 6, 0>("mfma w0, 14 fma behind each (same wave)", 0x001, 0, 1, "");
    run<0, 5, 0>("mfma w0,w4, 8 fma behind each", 0x011, 0, 1, "");
    run<0, 6, 0>("mfma w0,w4, 14 fma behind each", 0x011, 0, 1, "");
    run<1, 0, 0>("16x16x4: mfma w0 alone", 0x001, 0, 1, "");
    run<1, 0, 0>("16x16x4: mfma w0,w4", 0x011, 0, 1, "");
    run<1, 0, 0>("16x16x4: fma-indep w8 + mfma w0,w4", 0x011, 0x100, 32, "cyc/fma");
    run<1, 3, 0>("16x16x4: div w8 + mfma w0,w4", 0x011, 0x100, 8, "cyc/div");
    run_chip(1, "one workgroup (one CU)");
    run_chip(32, "32 workgroups");
    run_chip(256, "256 workgroups (every CU)");
    run_chip(256, "256 workgroups again");
    return 0;
}
