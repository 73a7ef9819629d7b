// ddgi_blend_sample.hip — DDGI-mode kernels (the reference's dormant pieces switched on):
//   k_blend_weights + k_probe_blend_s (and the one-probe-per-workgroup k_probe_blend)
//                        octahedral irradiance (8x8 rgba f32) + depth-moment (16x16 rg f32) tile update
//                        with temporal hysteresis — the dormant line probe_pass.comp:298-299
//                        `color = mix(old, new, hysteresis)` applied to DDGI-paper tiles
//   k_probe_sample_ddgi  get_diffuse_gi (intersection.glsl:1306-1409) with its dormant Chebyshev
//                        lines (1363-1383) enabled and octahedral bilinear tile fetches
#include "ddgi_device.h"
#include "ddgi_oct.h"
#include "ddgi_sampler.h"

#include <algorithm>

namespace ddgi {

// ------------------------------------------------------------------------------------------------
// The octahedral blend (DDGI paper; hysteresis: dormant probe_pass.comp:298-299).
//
// For every probe and interior texel:  sum_i w(texel, ray i) * value(probe, ray i)  over the frame's
// rays IN ORDER i = 0..n-1 as one fma chain per channel (the oracle's order), divided by the weight
// sum, mixed with the old texel, border texels copied from their octahedral-wrap source.
//
//  * A texel's weights depend on the frame's ray directions and the texel direction only — not on
//    the probe.  k_blend_weights evaluates them once per update: w[i][c] for ray i and texel column c
//    (256 columns: [0,196) depth texels, [196,232) irradiance texels, the rest zero) plus every
//    column's weight sum.
//  * A ray's values are the same for every texel lane, so the trace kernel stores them as
//    records of 8 probes (ddgi_types.h: kRecGroup) and k_probe_blend_s reads them with SCALAR
//    loads: the inner loop is one coalesced w load per ray and one fma (SGPR x VGPR + VGPR) per
//    (texel, ray, probe, channel).  No LDS, no barrier: a wave owns a set of texel columns (one
//    depth wave of 49 lanes x 4 texels, one irradiance wave of 36 lanes) for kNG record groups at a
//    time, finishes its texels (division, hysteresis) and writes them and their border copies itself.
//  HBM traffic per probe: 20 n B of ray records in, 3 KB old tiles in, 3 KB new tiles out.
//  Measured on C3 (16 384 probes x 256 rays): 0.157 ms + 0.017 ms for the weights (one-probe-per-
//  workgroup kernel: 0.52 ms); of it ~70 us is the fma stream at the VALU's rate (2.1 G lane-fma,
//  49/64 and 36/64 lanes used) and ~50 us the tile read-modify-write at ~4 TB/s, executed by the same
//  waves one after the other.
// ------------------------------------------------------------------------------------------------
constexpr int kIrrInterior = (kIrrTile - 2) * (kIrrTile - 2);  // 36
constexpr int kDepInterior = (kDepTile - 2) * (kDepTile - 2);  // 196
constexpr int kBlendCols = 256;
// tuning knob of k_probe_blend_s: record groups (of 8 probes) per pass
#ifndef DDGI_BLEND_NG
#define DDGI_BLEND_NG 1
#endif

typedef float f2v __attribute__((ext_vector_type(2)));
typedef float f8v __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
// Scalar-memory requests whose position in the instruction stream is fixed (volatile asm), and the wait
// that makes their results readable; the "+s" operands tie the wait to every later use of the registers.
DDGI_D void sload16(f16v& dst, const float* p)
{
    asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=s"(dst) : "s"(p));
}
DDGI_D void sload8(f8v& dst, const float* p) { asm volatile("s_load_dwordx8 %0, %1, 0x0" : "=s"(dst) : "s"(p)); }
template <int N>
DDGI_D void swait(f16v (&r)[N])
{
    if constexpr (N == 1) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(r[0]));
    if constexpr (N == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(r[0]), "+s"(r[1]));
    if constexpr (N == 4) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(r[0]), "+s"(r[1]), "+s"(r[2]), "+s"(r[3]));
    static_assert(N == 1 || N == 2 || N == 4, "record groups per pass");
}
DDGI_D void swait2(f16v& a, f8v& b) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a), "+s"(b)); }

DDGI_D void blend_column_texel(int c, bool& is_dep, int& tx, int& ty)
{
    is_dep = c < kDepInterior;
    const int k = is_dep ? c : c - kDepInterior;
    const int inner = is_dep ? (kDepTile - 2) : (kIrrTile - 2);
    tx = 1 + k % inner, ty = 1 + k / inner;
}

__global__ __launch_bounds__(kBlendCols) void k_blend_weights(const BlendArgs A)
{
    extern __shared__ __attribute__((aligned(16))) float blend_lds[];
    const int n = A.grid.n;
    const int c = threadIdx.x;
    for (int i = c; i < n; i += kBlendCols)
    {
        const f3 d = fibonacci_dir(i, n, A.rot);
        blend_lds[3 * i] = d.x, blend_lds[3 * i + 1] = d.y, blend_lds[3 * i + 2] = d.z;
    }
    __syncthreads();
    if (c >= kDepInterior + kIrrInterior)
    {
        for (int i = 0; i < n; ++i) A.w[static_cast<size_t>(i) * kBlendCols + c] = 0.0f;
        A.w_sum[c] = 0.0f;
        return;
    }
    bool is_dep;
    int tx, ty;
    blend_column_texel(c, is_dep, tx, ty);
    const f3 td = texel_dir(tx, ty, is_dep ? kDepTile : kIrrTile);
    float sw = 0.0f;
    for (int i = 0; i < n; ++i)
    {
        float w = gl_max(0.0f, dot3(td, f3{blend_lds[3 * i], blend_lds[3 * i + 1], blend_lds[3 * i + 2]}));
        if (is_dep) w = pow50(w);
        A.w[static_cast<size_t>(i) * kBlendCols + c] = w;
        sw += w;
    }
    A.w_sum[c] = sw;
}

// slab-major tile slot of local probe pl ((y, zl, x) enumeration of the rank's slab)
DDGI_D size_t blend_tile_slot(const GridK& G, uint32_t pl)
{
    const int slab_row = G.czl * G.cx;
    const int y = static_cast<int>(pl) / slab_row;
    const int rem = static_cast<int>(pl) - y * slab_row;
    const int zl = rem / G.cx;
    const int x = rem - zl * G.cx;
    return (static_cast<size_t>(G.z0 + zl) * G.cy + y) * G.cx + x;
}

// texel (tx, ty) of a tile of side S and the border texels that copy it (octahedral wrap: ddgi_oct.h
// border_source, inverted); unused entries are -1
DDGI_D void blend_destinations(int tx, int ty, int S, int (&dst)[4])
{
    const int last = S - 1;
    dst[0] = ty * S + tx, dst[1] = dst[2] = dst[3] = -1;
    int nd = 1;
    if (ty == 1) dst[nd++] = last - tx;                      // (last-tx, 0)
    if (ty == last - 1) dst[nd++] = last * S + (last - tx);  // (last-tx, last)
    if (tx == 1) dst[nd++] = (last - ty) * S;                // (0, last-ty)
    if (tx == last - 1) dst[nd++] = (last - ty) * S + last;  // (last, last-ty)
    if (tx == 1 && ty == 1) dst[nd++] = last * S + last;
    if (tx == last - 1 && ty == last - 1) dst[nd++] = 0;
    if (tx == 1 && ty == last - 1) dst[nd++] = last;      // (last, 0)
    if (tx == last - 1 && ty == 1) dst[nd++] = last * S;  // (0, last)
}

// Workgroup = 2 waves working on the same kNG record groups: wave 0 the 196 depth texels (49 lanes x
// kDepT = 4 consecutive texel columns each, so that every scalar record value feeds 4 fma's per lane —
// the scalar data path returns only a few bytes per clock), wave 1 the 36 irradiance texels.
constexpr int kDepT = 4, kDepLanes = kDepInterior / kDepT;  // 49
static_assert(kDepLanes * kDepT == kDepInterior && kDepLanes <= 64 && kIrrInterior <= 64, "texel columns per lane");

template <int kNG>
__global__ __launch_bounds__(128) void k_probe_blend_s(const BlendArgs A, const float* __restrict__ rad_rgb, const float* __restrict__ rad_dd,
                                                       const float* __restrict__ w_table, const float* __restrict__ w_sum)
{
    // the read-only inputs come as separate noalias kernel arguments (not through BlendArgs) so that the
    // compiler can prove the tile stores do not clobber them and selects scalar loads for the ray records
    const GridK& G = A.grid;
    const int n = G.n;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool is_dep = wave == 0;  // wave-uniform
    const float hyst = G.hysteresis;
    const uint32_t n_groups = (A.n_local_probes + kRecGroup - 1) / kRecGroup;
    const uint32_t n_super = (n_groups + kNG - 1) / kNG;
    const int n_even = n & ~1, last = n - 1;

    if (is_dep)
    {
        const bool valid = lane < kDepLanes;
        const int c0 = valid ? lane * kDepT : kBlendCols - kDepT;  // columns 252..255: zero weights
        int dst[kDepT][4];
#pragma unroll
        for (int t = 0; t < kDepT; ++t)
        {
            const int k = (valid ? c0 : 0) + t;
            blend_destinations(1 + k % (kDepTile - 2), 1 + k / (kDepTile - 2), kDepTile, dst[t]);
        }
        const float4 sw4 = *reinterpret_cast<const float4*>(w_sum + c0);
        const float sw[kDepT] = {sw4.x, sw4.y, sw4.z, sw4.w};
        const float4* __restrict__ wcol = reinterpret_cast<const float4*>(w_table + c0);  // row stride kBlendCols / 4
        constexpr int kRow = kBlendCols / 4;

        for (uint32_t sg = blockIdx.x; sg < n_super; sg += gridDim.x)
        {
            const uint32_t g0 = sg * kNG;
            // accumulators as (probe 2k, probe 2k+1) pairs: [group][pair][texel]
            f2v a1[kNG][4][kDepT], a2[kNG][4][kDepT];
#pragma unroll
            for (int g = 0; g < kNG; ++g)
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int t = 0; t < kDepT; ++t) a1[g][k][t] = a2[g][k][t] = f2v{0.0f, 0.0f};
            const float* __restrict__ rec = rad_dd + static_cast<size_t>(g0) * n * 16;
            // Software pipeline over ray pairs with two scalar register sets X, Y: the records of ray
            // i+1 are requested (s_load_dwordx16, written as asm so that the request stays where it is)
            // right after ray i's have arrived and before ray i's fma's are issued.
            f16v X[kNG], Y[kNG];
#pragma unroll
            for (int g = 0; g < kNG; ++g) sload16(X[g], rec + (static_cast<size_t>(g) * n) * 16);
            // weights run two ray pairs ahead (their L2 latency is longer than one pair's arithmetic)
            float4 wx = wcol[0], wy = wcol[static_cast<size_t>(min(1, last)) * kRow];
            float4 wx_n = wcol[static_cast<size_t>(min(2, last)) * kRow], wy_n = wcol[static_cast<size_t>(min(3, last)) * kRow];
            auto accumulate = [&](const f16v(&R)[kNG], const float4& w4) {
                const float w[kDepT] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
                for (int g = 0; g < kNG; ++g)
#pragma unroll
                    for (int k = 0; k < 4; ++k)
#pragma unroll
                        for (int t = 0; t < kDepT; ++t)
                        {
                            a1[g][k][t] = __builtin_elementwise_fma(f2v{R[g][2 * k], R[g][2 * k + 1]}, f2v{w[t], w[t]}, a1[g][k][t]);
                            a2[g][k][t] = __builtin_elementwise_fma(f2v{R[g][8 + 2 * k], R[g][9 + 2 * k]}, f2v{w[t], w[t]}, a2[g][k][t]);
                        }
            };
            for (int i = 0; i < n_even; i += 2)
            {
                const int i2 = min(i + 2, last);  // requests past the end repeat the last ray and are dropped
                const float4 wx_nn = wcol[static_cast<size_t>(min(i + 4, last)) * kRow], wy_nn = wcol[static_cast<size_t>(min(i + 5, last)) * kRow];
                swait(X);
#pragma unroll
                for (int g = 0; g < kNG; ++g) sload16(Y[g], rec + (static_cast<size_t>(g) * n + i + 1) * 16);
                accumulate(X, wx);
                swait(Y);
#pragma unroll
                for (int g = 0; g < kNG; ++g) sload16(X[g], rec + (static_cast<size_t>(g) * n + i2) * 16);
                accumulate(Y, wy);
                wx = wx_n, wy = wy_n;
                wx_n = wx_nn, wy_n = wy_nn;
            }
            swait(X);  // X holds ray n_even (the last ray when n is odd; else a repeat that is dropped)
            if (n & 1) accumulate(X, wx);
#pragma unroll
            for (int q = 0; q < kNG * kRecGroup; ++q)
            {
                const uint32_t pl = g0 * kRecGroup + q;
                if (pl < A.n_local_probes && valid)
                {
                    const size_t tile_off = blend_tile_slot(G, pl) * (kDepTile * kDepTile * 2);
                    float* tile = A.depth + tile_off;
                    const float* tile_old = A.depth_old + tile_off;
#pragma unroll
                    for (int t = 0; t < kDepT; ++t)
                    {
                        const float s1 = a1[q / kRecGroup][(q % kRecGroup) / 2][t][q & 1], s2 = a2[q / kRecGroup][(q % kRecGroup) / 2][t][q & 1];
                        float r1 = 0.0f, r2 = 0.0f;
                        if (sw[t] > 1e-6f) r1 = s1 / sw[t], r2 = s2 / sw[t];
                        const float2 old = *reinterpret_cast<const float2*>(tile_old + dst[t][0] * 2);
                        const float2 out{gl_mix(old.x, r1, hyst), gl_mix(old.y, r2, hyst)};
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            if (dst[t][k] >= 0) *reinterpret_cast<float2*>(tile + dst[t][k] * 2) = out;
                    }
                }
            }
        }
    }
    else
    {
        const bool valid = lane < kIrrInterior;
        const int c = valid ? kDepInterior + lane : kBlendCols - 1;  // column 255: zero weights
        int dst[4];
        blend_destinations(1 + (valid ? lane : 0) % (kIrrTile - 2), 1 + (valid ? lane : 0) / (kIrrTile - 2), kIrrTile, dst);
        const float sw = w_sum[c];
        const float* __restrict__ wcol = w_table + c;

        for (uint32_t sg = blockIdx.x; sg < n_super; sg += gridDim.x)
        {
            // the irradiance wave takes its record groups one after the other (24 scalars per ray and group)
            for (int g = 0; g < kNG; ++g)
            {
                const uint32_t grp = sg * kNG + g;
                f2v ac[3][4];  // [channel][probe pair]
#pragma unroll
                for (int ch = 0; ch < 3; ++ch)
#pragma unroll
                    for (int k = 0; k < 4; ++k) ac[ch][k] = f2v{0.0f, 0.0f};
                const float* __restrict__ rec = rad_rgb + static_cast<size_t>(grp) * n * 24;
                f16v X16, Y16;
                f8v X8, Y8;
                sload16(X16, rec);
                sload8(X8, rec + 16);
                float wx = wcol[0], wy = wcol[static_cast<size_t>(min(1, last)) * kBlendCols];
                auto accumulate = [&](const f16v& R16, const f8v& R8, float w) {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                    {
                        ac[0][k] = __builtin_elementwise_fma(f2v{R16[2 * k], R16[2 * k + 1]}, f2v{w, w}, ac[0][k]);
                        ac[1][k] = __builtin_elementwise_fma(f2v{R16[8 + 2 * k], R16[9 + 2 * k]}, f2v{w, w}, ac[1][k]);
                        ac[2][k] = __builtin_elementwise_fma(f2v{R8[2 * k], R8[2 * k + 1]}, f2v{w, w}, ac[2][k]);
                    }
                };
                for (int i = 0; i < n_even; i += 2)
                {
                    const int i2 = min(i + 2, last), i3 = min(i + 3, last);
                    const float wx_n = wcol[static_cast<size_t>(i2) * kBlendCols], wy_n = wcol[static_cast<size_t>(i3) * kBlendCols];
                    swait2(X16, X8);
                    sload16(Y16, rec + static_cast<size_t>(i + 1) * 24);
                    sload8(Y8, rec + static_cast<size_t>(i + 1) * 24 + 16);
                    accumulate(X16, X8, wx);
                    swait2(Y16, Y8);
                    sload16(X16, rec + static_cast<size_t>(i2) * 24);
                    sload8(X8, rec + static_cast<size_t>(i2) * 24 + 16);
                    accumulate(Y16, Y8, wy);
                    wx = wx_n, wy = wy_n;
                }
                swait2(X16, X8);
                if (n & 1) accumulate(X16, X8, wx);
#pragma unroll
                for (int j = 0; j < kRecGroup; ++j)
                {
                    const uint32_t pl = grp * kRecGroup + j;
                    if (pl < A.n_local_probes && valid)
                    {
                        const size_t tile_off = blend_tile_slot(G, pl) * (kIrrTile * kIrrTile * 4);
                        float* tile = A.irradiance + tile_off;
                        float res[3] = {0.0f, 0.0f, 0.0f};
                        if (sw > 1e-6f) res[0] = ac[0][j / 2][j & 1] / sw, res[1] = ac[1][j / 2][j & 1] / sw, res[2] = ac[2][j / 2][j & 1] / sw;
                        const float4 old = *reinterpret_cast<const float4*>(A.irradiance_old + tile_off + dst[0] * 4);
                        const float4 out{gl_mix(old.x, res[0], hyst), gl_mix(old.y, res[1], hyst), gl_mix(old.z, res[2], hyst), 1.0f};
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            if (dst[k] >= 0) *reinterpret_cast<float4*>(tile + dst[k] * 4) = out;
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_probe_blend — the same blend with one 256-lane workgroup per probe, lane = texel, weights
// evaluated in place and the ray records staged in LDS.  Used for ray counts whose direction table
// does not fit k_blend_weights' LDS, and as the cross-check of k_probe_blend_s
// (DDGI_BLEND_KERNEL=probe, tests/test_gpu_ddgi_mode.py).
// ------------------------------------------------------------------------------------------------
constexpr int kBlendBlock = 256;

__global__ __launch_bounds__(kBlendBlock) void k_probe_blend(const BlendArgs A)
{
    extern __shared__ __attribute__((aligned(16))) float blend_lds[];
    const GridK& G = A.grid;
    const int n = G.n;
    float4* s_rad = reinterpret_cast<float4*>(blend_lds);  // n: r, g, b, clamped distance
    float* s_dir = blend_lds + 4 * n;                      // 3 n
    float* s_irr = s_dir + 3 * n;                          // 8*8*4
    float* s_dep = s_irr + kIrrTile * kIrrTile * 4;        // 16*16*2
    const int tid = threadIdx.x;

    for (uint32_t pl = blockIdx.x; pl < A.n_local_probes; pl += gridDim.x)
    {
        const size_t slot = blend_tile_slot(G, pl);
        float* g_irr = A.irradiance + slot * (kIrrTile * kIrrTile * 4);
        float* g_dep = A.depth + slot * (kDepTile * kDepTile * 2);
        const float* rgb = A.rad_rgb + static_cast<size_t>(pl / kRecGroup) * n * 24 + (pl % kRecGroup);
        const float* dd = A.rad_dd + static_cast<size_t>(pl / kRecGroup) * n * 16 + (pl % kRecGroup);

        __syncthreads();  // previous probe's LDS fully consumed
        for (int i = tid; i < n; i += kBlendBlock)
        {
            s_rad[i] = float4{rgb[i * 24], rgb[i * 24 + 8], rgb[i * 24 + 16], dd[i * 16]};
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
                sr = fmaf(r.x, w, sr), sg = fmaf(r.y, w, sg), sb = fmaf(r.z, w, sb);
                sw += w;
            }
            float res[3] = {0.0f, 0.0f, 0.0f};
            if (sw > 1e-6f) res[0] = sr / sw, res[1] = sg / sw, res[2] = sb / sw;
            const int o = (ty * kIrrTile + tx) * 4;
            const float4 old = *reinterpret_cast<const float4*>(A.irradiance_old + slot * (kIrrTile * kIrrTile * 4) + o);
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
            float sw = 0.0f, s1 = 0.0f, s2 = 0.0f;
            for (int i = 0; i < n; ++i)
            {
                const float w = pow50(gl_max(0.0f, dot3(td, f3{s_dir[3 * i], s_dir[3 * i + 1], s_dir[3 * i + 2]})));
                const float d = s_rad[i].w;
                s1 = fmaf(d, w, s1), s2 = fmaf(d * d, w, s2);
                sw += w;
            }
            float r1 = 0.0f, r2 = 0.0f;
            if (sw > 1e-6f) r1 = s1 / sw, r2 = s2 / sw;
            const int o = (ty * kDepTile + tx) * 2;
            const float2 old = *reinterpret_cast<const float2*>(A.depth_old + slot * (kDepTile * kDepTile * 2) + o);
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

// sizes of the per-update weight table (BlendArgs::w; w_sum holds kBlendCols floats) and of the ray
// records for n_local_probes probes of n rays (rounded up to whole passes of the blend kernel)
constexpr int kBlendNG = DDGI_BLEND_NG;
size_t blend_weights_floats(int n) { return static_cast<size_t>(n) * kBlendCols; }
size_t blend_record_groups(uint32_t n_local_probes)
{
    const size_t groups = (static_cast<size_t>(n_local_probes) + kRecGroup - 1) / kRecGroup;
    return (groups + kBlendNG - 1) / kBlendNG * kBlendNG;
}

hipError_t launch_probe_blend(const BlendArgs& args, int num_cus, hipStream_t stream)
{
    const int n = args.grid.n;
    if (args.n_local_probes == 0) return hipSuccess;
    const size_t dir_lds = static_cast<size_t>(3) * n * sizeof(float);
    if (args.w && args.w_sum && dir_lds <= 64 * 1024)
    {
        hipLaunchKernelGGL(k_blend_weights, dim3(1), dim3(kBlendCols), dir_lds, stream, args);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        const uint32_t n_super = static_cast<uint32_t>(blend_record_groups(args.n_local_probes) / kBlendNG);
        const uint32_t blocks = std::min<uint32_t>(n_super, static_cast<uint32_t>(num_cus) * 16u);
        hipLaunchKernelGGL(k_probe_blend_s<kBlendNG>, dim3(blocks), dim3(128), 0, stream, args, args.rad_rgb, args.rad_dd,
                           static_cast<const float*>(args.w), static_cast<const float*>(args.w_sum));
        return hipGetLastError();
    }
    const size_t lds = (static_cast<size_t>(7) * n + kIrrTile * kIrrTile * 4 + kDepTile * kDepTile * 2) * sizeof(float);
    const uint32_t blocks = std::min<uint32_t>(args.n_local_probes, static_cast<uint32_t>(num_cus) * 8u);
    hipLaunchKernelGGL(k_probe_blend, dim3(blocks), dim3(kBlendBlock), lds, stream, args);
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
