// write_size_calib.hip — what does rocprofv3's WRITE_SIZE count for 4-byte scattered stores?  (MI355X_MICROARCH.md: the
// counter is uncalibrated on gfx950.)  Three kernels that each store exactly N dwords (N * 4 bytes of payload):
//   k_coalesced   lane i -> word i                      (a wave writes 256 contiguous bytes)
//   k_strided64   lane i -> word 16 i                   (every store alone in its 64-byte line; lines are never completed)
//   k_texel_like  lane i -> word permuted inside 1 KiB tiles chosen at random (the trace kernel's pattern: a ray's texel is 4 bytes
//                 of its probe's 1 KiB tile; the tile's other texels arrive from other waves at other times)
//   hipcc --offload-arch=gfx950 -O2 write_size_calib.hip -o write_size_calib.bin
//   rocprofv3 --kernel-trace --pmc WRITE_SIZE -d out -o w --output-format csv -- ./write_size_calib.bin
#include <hip/hip_runtime.h>

#include <cstdio>

__global__ void k_coalesced(uint32_t* p, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = i;
}
__global__ void k_strided64(uint32_t* p, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[static_cast<size_t>(i) * 16] = i;
}
__device__ uint32_t mix(uint32_t x)
{
    x ^= x >> 16, x *= 0x7feb352du, x ^= x >> 15, x *= 0x846ca68bu, x ^= x >> 16;
    return x;
}
__global__ void k_texel_like(uint32_t* p, uint32_t n)
{
    // a bijection of [0, n) (n a power of two): the tile index and the texel inside the tile are both scrambled
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t j = (i * 2654435761u) & (n - 1u);  // odd multiplier: a permutation of [0, n)
    p[j] = mix(i);
}

int main()
{
    const uint32_t n = 1u << 22;  // 4 Mi dwords = 16 MiB of payload, the size of C3's albedo texture
    uint32_t* p = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&p), static_cast<size_t>(n) * 16 * 4) != hipSuccess) return 1;
    hipMemset(p, 0, static_cast<size_t>(n) * 16 * 4);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 3; ++rep)
    {
        hipLaunchKernelGGL(k_coalesced, dim3(n / 256), dim3(256), 0, 0, p, n);
        hipLaunchKernelGGL(k_strided64, dim3(n / 256), dim3(256), 0, 0, p, n);
        hipLaunchKernelGGL(k_texel_like, dim3(n / 256), dim3(256), 0, 0, p, n);
    }
    hipDeviceSynchronize();
    std::printf("each kernel stores %u dwords = %.1f MB of payload\n", n, n * 4 / 1e6);
    return 0;
}
