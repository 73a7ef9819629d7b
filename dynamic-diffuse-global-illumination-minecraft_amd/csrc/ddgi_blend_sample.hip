// ddgi_blend_sample.hip — DDGI-mode kernels (the reference's dormant pieces switched on):
//   k_probe_blend        octahedral irradiance (8x8 rgba f32) + depth-moment (16x16 rg f32) tile update
//                        with temporal hysteresis — the dormant line probe_pass.comp:298-299
//                        `color = mix(old, new, hysteresis)` applied to DDGI-paper tiles
//   k_probe_sample_ddgi  get_diffuse_gi (intersection.glsl:1306-1409) with its dormant Chebyshev
//                        lines (1363-1383) enabled and octahedral bilinear tile fetches
#include "ddgi_device.h"
#include "ddgi_oct.h"
#include "ddgi_sampler.h"

namespace ddgi {

// ------------------------------------------------------------------------------------------------
// k_probe_blend — one 256-lane workgroup per probe; lane = texel (36 irradiance + 196 depth
// interior texels = 232 lanes).  The probe's n ray records (radiance rgb + first-hit distance,
// 16 B each, coalesced) and the frame's n ray directions are staged in LDS; every texel lane then
// walks the rays in order i = 0..n-1 (LDS broadcast reads, no bank conflicts), which is exactly
// the summation order of the oracle — no cross-lane reduction, no order ambiguity.  New texels go
// through LDS so the border wrap can be applied before one coalesced store of both tiles.
// HBM traffic per probe: 16 n B ray records in, 3 KB old tiles in, 3 KB new tiles out.
// ------------------------------------------------------------------------------------------------
constexpr int kBlendBlock = 256;
constexpr int kIrrInterior = (kIrrTile - 2) * (kIrrTile - 2);  // 36
constexpr int kDepInterior = (kDepTile - 2) * (kDepTile - 2);  // 196

__global__ __launch_bounds__(kBlendBlock) void k_probe_blend(const BlendArgs A)
{
    extern __shared__ __attribute__((aligned(16))) float blend_lds[];
    const GridK& G = A.grid;
    const int n = G.s * G.s;
    float4* s_rad = reinterpret_cast<float4*>(blend_lds);  // n
    float* s_dir = blend_lds + 4 * n;                      // 3 n
    float* s_irr = s_dir + 3 * n;                          // 8*8*4
    float* s_dep = s_irr + kIrrTile * kIrrTile * 4;        // 16*16*2
    const int tid = threadIdx.x;

    for (uint32_t pl = blockIdx.x; pl < A.n_local_probes; pl += gridDim.x)
    {
        // local (y, zl, x) enumeration -> slab-major tile slot
        const int slab_row = G.czl * G.cx;
        const int y = static_cast<int>(pl) / slab_row;
        const int rem = static_cast<int>(pl) - y * slab_row;
        const int zl = rem / G.cx;
        const int x = rem - zl * G.cx;
        const size_t slot = (static_cast<size_t>(G.z0 + zl) * G.cy + y) * G.cx + x;
        float* g_irr = A.irradiance + slot * (kIrrTile * kIrrTile * 4);
        float* g_dep = A.depth + slot * (kDepTile * kDepTile * 2);

        __syncthreads();  // previous probe's LDS fully consumed
        for (int i = tid; i < n; i += kBlendBlock)
        {
            s_rad[i] = A.radiance[static_cast<size_t>(pl) * n + i];
            const f3 d = fibonacci_dir(i, n, A.rot);
            s_dir[3 * i] = d.x, s_dir[3 * i + 1] = d.y, s_dir[3 * i + 2] = d.z;
        }
        __syncthreads();

        const float hyst = G.hysteresis;
        if (tid < kIrrInterior)
        {
            const int tx = 1 + tid % (kIrrTile - 2), ty = 1 + tid / (kIrrTile - 2);
            const f3 td = texel_dir(tx, ty, kIrrTile);
            float sw = 0.0f, sr = 0.0f, sg = 0.0f, sb = 0.0f;
            for (int i = 0; i < n; ++i)
            {
                const float4 r = s_rad[i];
                const float w = gl_max(0.0f, dot3(td, f3{s_dir[3 * i], s_dir[3 * i + 1], s_dir[3 * i + 2]}));
                sr += r.x * w, sg += r.y * w, sb += r.z * w;
                sw += w;
            }
            float res[3] = {0.0f, 0.0f, 0.0f};
            if (sw > 1e-6f) res[0] = sr / sw, res[1] = sg / sw, res[2] = sb / sw;
            const int o = (ty * kIrrTile + tx) * 4;
            const float4 old = *reinterpret_cast<const float4*>(g_irr + o);
            s_irr[o + 0] = gl_mix(old.x, res[0], hyst);
            s_irr[o + 1] = gl_mix(old.y, res[1], hyst);
            s_irr[o + 2] = gl_mix(old.z, res[2], hyst);
            s_irr[o + 3] = 1.0f;
        }
        else if (tid < kIrrInterior + kDepInterior)
        {
            const int k = tid - kIrrInterior;
            const int tx = 1 + k % (kDepTile - 2), ty = 1 + k / (kDepTile - 2);
            const f3 td = texel_dir(tx, ty, kDepTile);
            const float max_dist = static_cast<float>(G.side) * 1.5f;
            float sw = 0.0f, s1 = 0.0f, s2 = 0.0f;
            for (int i = 0; i < n; ++i)
            {
                const float w = pow50(gl_max(0.0f, dot3(td, f3{s_dir[3 * i], s_dir[3 * i + 1], s_dir[3 * i + 2]})));
                const float d = gl_min(s_rad[i].w, max_dist);
                s1 += d * w, s2 += (d * d) * w;
                sw += w;
            }
            float r1 = 0.0f, r2 = 0.0f;
            if (sw > 1e-6f) r1 = s1 / sw, r2 = s2 / sw;
            const int o = (ty * kDepTile + tx) * 2;
            const float2 old = *reinterpret_cast<const float2*>(g_dep + o);
            s_dep[o + 0] = gl_mix(old.x, r1, hyst);
            s_dep[o + 1] = gl_mix(old.y, r2, hyst);
        }
        __syncthreads();
        // both tiles out, border texels taken from their wrap source
        for (int t = tid; t < kIrrTile * kIrrTile; t += kBlendBlock)
        {
            int sx = t % kIrrTile, sy = t / kIrrTile;
            if (sx == 0 || sy == 0 || sx == kIrrTile - 1 || sy == kIrrTile - 1) border_source(sx, sy, kIrrTile, sx, sy);
            const float* src = s_irr + (sy * kIrrTile + sx) * 4;
            *reinterpret_cast<float4*>(g_irr + t * 4) = float4{src[0], src[1], src[2], src[3]};
        }
        for (int t = tid; t < kDepTile * kDepTile; t += kBlendBlock)
        {
            int sx = t % kDepTile, sy = t / kDepTile;
            if (sx == 0 || sy == 0 || sx == kDepTile - 1 || sy == kDepTile - 1) border_source(sx, sy, kDepTile, sx, sy);
            const float* src = s_dep + (sy * kDepTile + sx) * 2;
            *reinterpret_cast<float2*>(g_dep + t * 2) = float2{src[0], src[1]};
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_probe_sample_ddgi — one lane per shading point
// ------------------------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void k_probe_sample_ddgi(const SampleArgs A)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= A.n) return;
    int cage[8];
    const f3 out = diffuse_gi_ddgi(A.grid, A.irradiance, A.depth, f3{A.pos[3 * i], A.pos[3 * i + 1], A.pos[3 * i + 2]},
                                   f3{A.nrm[3 * i], A.nrm[3 * i + 1], A.nrm[3 * i + 2]}, cage);
    A.rgb[3 * i] = out.x, A.rgb[3 * i + 1] = out.y, A.rgb[3 * i + 2] = out.z;
    if (A.cage)
        for (int k = 0; k < 8; ++k) A.cage[8 * i + k] = cage[k];
}

// ---- launchers -----------------------------------------------------------------------------------

hipError_t launch_probe_blend(const BlendArgs& args, int grid_blocks, hipStream_t stream)
{
    const int n = args.grid.s * args.grid.s;
    const size_t lds = (static_cast<size_t>(7) * n + kIrrTile * kIrrTile * 4 + kDepTile * kDepTile * 2) * sizeof(float);
    if (grid_blocks < 1) return hipSuccess;
    hipLaunchKernelGGL(k_probe_blend, dim3(grid_blocks), dim3(kBlendBlock), lds, stream, args);
    return hipGetLastError();
}

hipError_t launch_probe_sample_ddgi(const SampleArgs& args, hipStream_t stream)
{
    const unsigned blocks = (args.n + 255u) / 256u;
    if (blocks == 0) return hipSuccess;
    hipLaunchKernelGGL(k_probe_sample_ddgi, dim3(blocks), dim3(256), 0, stream, args);
    return hipGetLastError();
}

}  // namespace ddgi
