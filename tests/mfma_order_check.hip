// tests/mfma_order_check.hip — the premise of the MFMA blend (csrc/ddgi_blend_sample.hip): on gfx950
// v_mfma_f32_32x32x2_f32 and v_mfma_f32_16x16x4_f32 accumulate k IN ORDER as one binary32 fma chain,
//     D = fma(a[k1], b[k1], fma(a[k0], b[k0], C)),   one rounding per product-and-add, nothing wider inside,
// so a contraction  sum_i w[t][i] * v[i][c]  issued as a sequence of MFMAs over ascending ray pairs is
// bit-identical to the oracle's per-texel loop  s = fmaf(v_i, w_i, s), i = 0..n-1.
// Inputs are chosen to make any other order or a wider accumulator visible: magnitudes spanning 2^-40..2^40,
// mixed signs (catastrophic cancellation), subnormal products.  Prints "OK" or the first mismatches.
// Built and run by tests/test_gpu_device_math.py.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));

// D[32x32] = sum_k A[32xK] * B[Kx32], A row-major [i][k], B row-major [k][j]
__global__ void k32(const float* A, const float* B, float* D, int K)
{
    const int l = threadIdx.x, i = l & 31, h = l >> 5;
    f16v acc = {0};
    for (int k = 0; k < K; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[i * K + k + h], B[(k + h) * 32 + i], acc, 0, 0, 0);
    // C/D map: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + i] = acc[r];
}

// D[16x16] = sum_k A[16xK] * B[Kx16]
__global__ void k16(const float* A, const float* B, float* D, int K)
{
    const int l = threadIdx.x, i = l & 15, h = l >> 4;
    f4v acc = {0};
    for (int k = 0; k < K; k += 4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[i * K + k + h], B[(k + h) * 16 + i], acc, 0, 0, 0);
    // C/D map: col = lane & 15, row = (lane >> 4) * 4 + reg
    for (int r = 0; r < 4; ++r) D[(h * 4 + r) * 16 + i] = acc[r];
}

static int check(int M, int K, const std::vector<float>& A, const std::vector<float>& B, const std::vector<float>& D, const char* name)
{
    int bad = 0;
    for (int i = 0; i < M; ++i)
        for (int j = 0; j < M; ++j)
        {
            float s = 0.0f;
            for (int k = 0; k < K; ++k) s = fmaf(A[i * K + k], B[k * M + j], s);
            uint32_t u, v;
            std::memcpy(&u, &s, 4), std::memcpy(&v, &D[i * M + j], 4);
            if (u != v && !(std::isnan(s) && std::isnan(D[i * M + j])))
                if (bad++ < 5) std::printf("%s [%d][%d]: fmaf chain %.9g (%08x)  mfma %.9g (%08x)\n", name, i, j, s, u, D[i * M + j], v);
        }
    return bad;
}

int main()
{
    const int K = 512;
    std::mt19937 rng(1234);
    std::uniform_real_distribution<float> uni(-1.0f, 1.0f);
    std::uniform_int_distribution<int> ex(-40, 40);
    int bad = 0;
    for (int trial = 0; trial < 6; ++trial)
        for (int M : {32, 16})
        {
            std::vector<float> A(M * K), B(K * M), D(M * M);
            for (auto& x : A) x = trial < 2 ? uni(rng) : std::ldexp(uni(rng), ex(rng));
            for (auto& x : B) x = trial < 2 ? uni(rng) : std::ldexp(uni(rng), trial >= 4 ? ex(rng) - 100 : ex(rng));  // trials 4, 5: subnormal products
            if (trial == 1)  // the blend's case: non-negative weights, values of one sign
            {
                for (auto& x : A) x = std::fabs(x);
                for (auto& x : B) x = std::fabs(x) * 40.0f;
            }
            float *dA, *dB, *dD;
            (void)hipMalloc(&dA, A.size() * 4), (void)hipMalloc(&dB, B.size() * 4), (void)hipMalloc(&dD, D.size() * 4);
            (void)hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
            (void)hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
            if (M == 32) hipLaunchKernelGGL(k32, dim3(1), dim3(64), 0, 0, dA, dB, dD, K);
            else hipLaunchKernelGGL(k16, dim3(1), dim3(64), 0, 0, dA, dB, dD, K);
            if (hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost) != hipSuccess)
            {
                std::printf("HIP error\n");
                return 2;
            }
            bad += check(M, K, A, B, D, M == 32 ? "32x32x2" : "16x16x4");
            (void)hipFree(dA), (void)hipFree(dB), (void)hipFree(dD);
        }
    std::printf(bad ? "MISMATCHES %d\n" : "OK\n", bad);
    return bad ? 1 : 0;
}
