// ddgi_blend_sample.hip — DDGI-mode kernels (the reference's dormant pieces switched on):
//   k_probe_blend        octahedral irradiance (8x8 rgba f32) + depth-moment (16x16 rg f32) tile update
//                        with temporal hysteresis — the dormant line probe_pass.comp:298-299
//                        `color = mix(old, new, hysteresis)` applied to DDGI-paper tiles
//   k_probe_sample_ddgi  get_diffuse_gi (intersection.glsl:1306-1409) with its dormant Chebyshev
//                        lines (1363-1383) enabled and octahedral bilinear tile fetches
#include "ddgi_device.h"
#include "ddgi_oct.h"

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

template <int kSide, int kCh>
DDGI_D void tile_fetch(const float* tile, f3 dir, float* out)
{
    const f2 uv = oct_encode(normalize3(dir));
    const float inner = static_cast<float>(kSide - 2);
    const float fx = (uv.x * 0.5f + 0.5f) * inner + 0.5f;  // texel-centre coordinate inside the bordered tile
    const float fy = (uv.y * 0.5f + 0.5f) * inner + 0.5f;
    const float bx = floorf(fx), by = floorf(fy);
    const float tx = fx - bx, ty = fy - by;
    int x0 = gl_int(bx), y0 = gl_int(by);
    int x1 = x0 + 1, y1 = y0 + 1;
    x0 = max(x0, 0), y0 = max(y0, 0);
    x1 = min(x1, kSide - 1), y1 = min(y1, kSide - 1);
    for (int c = 0; c < kCh; ++c)
    {
        const float a = tile[(y0 * kSide + x0) * kCh + c], b = tile[(y0 * kSide + x1) * kCh + c];
        const float cc = tile[(y1 * kSide + x0) * kCh + c], d = tile[(y1 * kSide + x1) * kCh + c];
        out[c] = gl_mix(gl_mix(a, b, tx), gl_mix(cc, d, tx), ty);
    }
}

__global__ __launch_bounds__(256) void k_probe_sample_ddgi(const SampleArgs A)
{
    const GridK& G = A.grid;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= A.n) return;
    const f3 pos{A.pos[3 * i], A.pos[3 * i + 1], A.pos[3 * i + 2]};
    const f3 N = normalize3(f3{A.nrm[3 * i], A.nrm[3 * i + 1], A.nrm[3 * i + 2]});
    const f3 origin{G.origin[0], G.origin[1], G.origin[2]};
    const float side = static_cast<float>(G.side);

    int cage[8];
    for (int k = 0; k < 8; ++k) cage[k] = -1;
    f3 out = mk3(1, 0, 1);
    bool ok = true;
    const f3 rel = div3(pos - origin, side);
    const int bx = gl_int(floorf(rel.x)), by = gl_int(floorf(rel.y)), bz = gl_int(floorf(rel.z));
    const int lo = gl_int(-floorf(static_cast<float>(G.cx) / 2.0f));  // Q6: x count for every axis
    const int hi = gl_int(floorf(static_cast<float>(G.cx) / 2.0f) - 1.0f);
    if (bx < lo || bx > hi || by < lo || by > hi || bz < lo || bz > hi) ok = false;
    if (ok)
    {
        const f3 base_world = f3{static_cast<float>(bx * G.side), static_cast<float>(by * G.side), static_cast<float>(bz * G.side)} + origin;
        const f3 a = div3(pos - base_world, side);
        const f3 alpha{gl_clamp(a.x, 0.0f, 1.0f), gl_clamp(a.y, 0.0f, 1.0f), gl_clamp(a.z, 0.0f, 1.0f)};
        f3 irr = mk3(0, 0, 0);
        float sum_w = 0.0f;
        const int n_probes = G.cx * G.cy * G.cz;
        for (int k = 0; k < 8; ++k)
        {
            const int ox = (k >> 2) & 1, oy = (k >> 1) & 1, oz = k & 1;                            // Q7
            const int sx = bx + ox + G.cx / 2, sy = by + oy + G.cy / 2, sz = bz + oz + G.cz / 2;  // Q4
            const int idx = sy * G.cx * G.cz + sz * G.cx + sx;
            if (idx < 0 || idx >= n_probes)
            {
                ok = false;
                break;
            }
            cage[k] = idx;
            const f3 tri{ox ? alpha.x : 1.0f - alpha.x, oy ? alpha.y : 1.0f - alpha.y, oz ? alpha.z : 1.0f - alpha.z};
            const f3 probe_pos = base_world + f3{static_cast<float>(ox * G.side), static_cast<float>(oy * G.side), static_cast<float>(oz * G.side)};
            const f3 dir = normalize3(probe_pos - pos);
            float tmp = gl_max(0.0001f, (dot3(dir, N) + 1.0f) * 0.5f);
            float weight = tmp * tmp + 0.2f;
            const size_t slot = static_cast<size_t>(slab_slot(G, idx));
            // moment visibility test (intersection.glsl:1363-1383, enabled)
            const float dist = length3(pos - probe_pos);
            float mms[2];
            tile_fetch<kDepTile, 2>(A.depth + slot * (kDepTile * kDepTile * 2), f3{-dir.x, -dir.y, -dir.z}, mms);
            const float mean = mms[0];
            const float variance = fabsf(mean * mean - mms[1]);
            tmp = gl_max(dist - mean, 0.0f);
            float cheb = variance / (variance + tmp * tmp);
            cheb = gl_max(cheb * cheb * cheb, 0.0f);
            if (!(dist <= mean)) weight *= cheb;
            weight = gl_max(0.000001f, weight);
            const float crush = 0.2f;
            if (weight < crush) weight *= weight * weight * (1.f / (crush * crush));
            weight *= tri.x * tri.y * tri.z;
            float c4[4];
            tile_fetch<kIrrTile, 4>(A.irradiance + slot * (kIrrTile * kIrrTile * 4), N, c4);
            irr = irr + f3{c4[0], c4[1], c4[2]} * weight;
            sum_w += weight;
        }
        if (ok) out = div3(irr, sum_w);
    }
    if (!ok)
    {
        out = mk3(1, 0, 1);
        for (int k = 0; k < 8; ++k) cage[k] = -1;
    }
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
