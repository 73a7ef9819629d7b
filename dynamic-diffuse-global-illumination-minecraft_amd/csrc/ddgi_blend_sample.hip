// ddgi_blend_sample.hip — DDGI-mode kernels (the reference's dormant pieces switched on):
//   k_blend_weights + k_probe_blend_depth[_res] / k_probe_blend_irr / k_probe_blend_mfma (and the one-probe-per-workgroup k_probe_blend)
//                        octahedral irradiance (8x8 rgba f32) + depth-moment (16x16 rg f32) tile update
//                        with temporal hysteresis — the dormant line probe_pass.comp:298-299
//                        `color = mix(old, new, hysteresis)` applied to DDGI-paper tiles
//   k_probe_sample_ddgi  get_diffuse_gi (intersection.glsl:1306-1409) with its dormant Chebyshev
//                        lines (1363-1383) enabled and octahedral bilinear tile fetches
#include "ddgi_device.h"
#include "ddgi_oct.h"
#include "ddgi_sampler.h"

#include <algorithm>
#include <type_traits>

namespace ddgi {

// ------------------------------------------------------------------------------------------------
// The octahedral blend (DDGI paper; hysteresis: dormant probe_pass.comp:298-299) as a dense contraction.
//
// For every probe and interior texel:  sum_i w(texel, ray i) * value(probe, ray i)  over the frame's rays IN ORDER
// i = 0..n-1 as one fma chain per channel (the oracle's order), divided by the weight sum, mixed with the old
// texel, border texels copied from their octahedral-wrap source.
//
// A texel's weights depend on the frame's ray directions and the texel direction only — not on the probe.  So
// the sums are a matrix product  D[texel][column] = sum_i W[texel][i] * V[i][column]  with the columns running over
// (probe, channel): W is 196 x n (depth texels: weights max(0, cos)^50) and 36 x n (irradiance texels: max(0, cos)),
// V are the ray records the trace kernel left (ddgi_types.h: rec_dd_index / rec_rgb_index).  On gfx950
// v_mfma_f32_32x32x2_f32 IS a k-ordered binary32 fma chain (D = fma(a1, b1, fma(a0, b0, C)), one rounding per
// term, nothing wider inside; tests/mfma_order_check.hip), so consecutive MFMAs over ascending ray pairs give
// bit for bit the oracle's per-texel loop — at the matrix pipe's rate with all 64 lanes busy, where the VALU
// version ran its fma stream at 49/64 and 36/64 lanes.
//
//   k_blend_weights      W as MFMA A tiles: tile mt (0..6 depth rows 32 mt .. 32 mt + 31, 7..8 irradiance), ray pair
//                        q, lane l holds W[row = l & 31][ray 2q + (l >> 5)]  ->  one coalesced 256-byte load per MFMA.
//                        Rows past the last texel and rays past n are zero (fma(0, x, acc) == acc: a chain that
//                        starts at +0 never holds -0).
//   k_blend_weight_sums  every texel's weight sum, added in ray order (one lane per texel, 16 loads in flight)
//   k_probe_blend_depth  workgroup = 7 waves, one depth tile each, for the 16 probes of a group: B = the group's (d, d^2)
//                        records, 32 columns = 2 moments x 16 probes.  One accumulator tile (16 VGPRs) per wave: a
//                        32x32x2 MFMA's issue interval equals its dependent latency, so one chain per wave keeps the
//                        pipe busy; operands are prefetched 8 ray pairs ahead.
//   k_probe_blend_irr    workgroup = 2 waves, one irradiance tile each, the three colour channels one after the other, for
//                        the 32 probes of a group; the epilogue joins a texel's channels: rgba texels leave as 16-byte stores.
//                        Epilogue of both: the accumulators are transposed through LDS so that lanes = texels: divide by
//                        the weight sum, hysteresis with the old texel (read from the previous tiles: the same buffers, or
//                        the other pair of a pipelined exchange), store the texel and its octahedral-wrap border copies —
//                        runs of consecutive addresses per probe, per-texel set-up done once per task.
// Both weight kernels depend on the frame's rotation only and are launched BEFORE the trace kernel.
// HBM traffic per probe: 20 n B of ray records in, 3 KB old tiles in, 3 KB new tiles out.
// ------------------------------------------------------------------------------------------------
constexpr int kIrrInterior = (kIrrTile - 2) * (kIrrTile - 2);  // 36
constexpr int kDepInterior = (kDepTile - 2) * (kDepTile - 2);  // 196
constexpr int kBlendCols = 256;
constexpr int kDepMTiles = (kDepInterior + 31) / 32;  // 7
constexpr int kIrrMTiles = (kIrrInterior + 31) / 32;  // 2
constexpr int kBlendMTiles = kDepMTiles + kIrrMTiles;
constexpr int kBlendWaves = kDepMTiles;               // waves per workgroup of k_probe_blend_mfma

typedef float f16v __attribute__((ext_vector_type(16)));

// texel column c in [0, 232) -> (M tile, row)
DDGI_D void blend_column_tile(int c, int& mt, int& row)
{
    const int k = c < kDepInterior ? c : c - kDepInterior;
    mt = (c < kDepInterior ? 0 : kDepMTiles) + (k >> 5);
    row = k & 31;
}
// index of W[tile mt][row][ray i] in the A-operand layout (ddgi_types.h: mfma_operand_offset)
DDGI_D size_t weight_index(int mt, int row, int i, int n_pad) { return static_cast<size_t>(mt) * n_pad * 32 + mfma_operand_offset(static_cast<uint32_t>(i), static_cast<uint32_t>(row)); }

DDGI_D void blend_column_texel(int c, bool& is_dep, int& tx, int& ty)
{
    is_dep = c < kDepInterior;
    const int k = is_dep ? c : c - kDepInterior;
    const int inner = is_dep ? (kDepTile - 2) : (kIrrTile - 2);
    tx = 1 + k % inner, ty = 1 + k / inner;
}

// grid: (ray chunks of 64, kBlendMTiles); block 64 x 4: thread (l, u) -> row l & 31, ray 2 (4 blockIdx.x + u) + (l >> 5)
__global__ __launch_bounds__(256) void k_blend_weights(const BlendArgs A)
{
    const int n = A.grid.n, n_pad = static_cast<int>(rec_ray_pad(static_cast<uint32_t>(n))), q_pairs = n_pad / 2;
    const int l = threadIdx.x & 63, q = static_cast<int>(blockIdx.x) * 4 + (threadIdx.x >> 6), mt = blockIdx.y;
    if (q >= q_pairs) return;
    const int row = l & 31, i = 2 * q + (l >> 5);
    const bool is_dep = mt < kDepMTiles;
    const int k = (is_dep ? mt : mt - kDepMTiles) * 32 + row;  // interior texel number within its tile kind
    float w = 0.0f;
    if (i < n && k < (is_dep ? kDepInterior : kIrrInterior))
    {
        const int inner = is_dep ? (kDepTile - 2) : (kIrrTile - 2);
        const f3 td = texel_dir(1 + k % inner, 1 + k / inner, is_dep ? kDepTile : kIrrTile);
        w = gl_max(0.0f, dot3(td, fibonacci_dir(i, n, A.rot)));
        if (is_dep) w = pow50(w);
    }
    A.w[weight_index(mt, row, i, n_pad)] = w;
}

// one lane per texel column: the weight sum in ray order (sw += w, i = 0..n-1: the oracle's order)
__global__ __launch_bounds__(kBlendCols) void k_blend_weight_sums(const BlendArgs A)
{
    const int n = A.grid.n, q_pairs = static_cast<int>(rec_ray_pad(static_cast<uint32_t>(n))) / 2;
    const int c = threadIdx.x;
    if (c >= kDepInterior + kIrrInterior)
    {
        A.w_sum[c] = 0.0f;
        return;
    }
    int mt, row;
    blend_column_tile(c, mt, row);
    const int n_pad = 2 * q_pairs;
    float sw = 0.0f;
    int i = 0;
    for (; i + 16 <= n; i += 16)
    {
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = A.w[weight_index(mt, row, i + u, n_pad)];
#pragma unroll
        for (int u = 0; u < 16; ++u) sw += v[u];
    }
    for (; i < n; ++i) sw += A.w[weight_index(mt, row, i, n_pad)];
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

// acc += A-operand stream x B-operand stream over all ray pairs, in order.  wa / vb: this lane's operand streams
// (ddgi_types.h: mfma_operand_offset), i.e. base + lane * 4: element u of float4 number k is ray pair 4k + u.
// Operands are fetched 16 ray pairs (4 float4 per stream) ahead of the MFMAs that consume them.
DDGI_D f16v blend_contract(const float* __restrict__ wa, const float* __restrict__ vb, int q_pairs)
{
    f16v acc = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    const float4* __restrict__ pa = reinterpret_cast<const float4*>(wa);
    const float4* __restrict__ pb = reinterpret_cast<const float4*>(vb);
    const int n4 = q_pairs / 4;  // float4 per stream; a multiple of 2 (kRecRayPad = 16)
    float4 a0[4], b0[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) a0[u] = pa[static_cast<size_t>(min(u, n4 - 1)) * 64], b0[u] = pb[static_cast<size_t>(min(u, n4 - 1)) * 64];
    for (int k = 0; k < n4; k += 4)
    {
        float4 a1[4], b1[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
        {
            const int kn = min(k + 4 + u, n4 - 1);  // past the end: re-read the last one (dropped)
            a1[u] = pa[static_cast<size_t>(kn) * 64], b1[u] = pb[static_cast<size_t>(kn) * 64];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (k + u < n4)  // wave-uniform
            {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[u].x, b0[u].x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[u].y, b0[u].y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[u].z, b0[u].z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[u].w, b0[u].w, acc, 0, 0, 0);
            }
#pragma unroll
        for (int u = 0; u < 4; ++u) a0[u] = a1[u], b0[u] = b1[u];
    }
    return acc;
}

// The same for three B streams (the colour channels, `b_stride` floats apart) against ONE pass over the A stream: three
// independent chains, each in ray order; the weights are read once instead of three times.
#ifdef DDGI_BLEND_LAPS  // timing build (tools/blend_laps.py): workgroup 0's waves stamp s_memtime at the start, and before / after every stage's barrier
__device__ unsigned long long g_blend_laps[14][24];  // rows 0-11: k_probe_blend_depth_res' waves; 12, 13: k_probe_blend_irr's
#define BLEND_LAP(i) do { if (blockIdx.x == 0 && lane == 0 && (i) < 24) g_blend_laps[wave][(i)] = __builtin_readcyclecounter(); } while (0)
#else
#define BLEND_LAP(i) do { } while (0)
#endif
#ifndef DDGI_IRR_EXP
#define DDGI_IRR_EXP 0
#endif
#ifndef DDGI_IRR_ORDER
#define DDGI_IRR_ORDER 1
#endif
// The same for 256 or 512 rays (n4 = 32 or 64 float4 per lane and stream), written out: with the loop above the compiler rotates the prefetch ring
// through ~190 register moves at the end of every trip — and waits for EVERY outstanding load first (in-kernel clocks: 36 000
// cycles for 384 MFMAs that take 24 576).  Unrolled, ring slot u simply is a set of registers, and each wait names the load it needs.
template <int kDepth, int n4, class AfterFill, class AtTail>
DDGI_D void blend_contract3_unrolled(const float* __restrict__ wa, const float* __restrict__ vb, size_t b_stride, f16v (&acc)[3], AfterFill&& after_fill, AtTail&& at_tail)
{
    static_assert(n4 % kDepth == 0, "the ring goes round a whole number of times");
#pragma unroll
    for (int c = 0; c < 3; ++c) acc[c] = f16v{0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    const float4* __restrict__ pa = reinterpret_cast<const float4*>(wa);
    const float4* __restrict__ pb[3] = {reinterpret_cast<const float4*>(vb), reinterpret_cast<const float4*>(vb + b_stride), reinterpret_cast<const float4*>(vb + 2 * b_stride)};
    float4 ab[kDepth], bb[3][kDepth];
#pragma unroll
    for (int u = 0; u < kDepth; ++u)
    {
        ab[u] = pa[static_cast<size_t>(u) * 64];
#pragma unroll
        for (int c = 0; c < 3; ++c) bb[c][u] = pb[c][static_cast<size_t>(u) * 64];
    }
#ifdef DDGI_BLEND_LAPS
    const int lane = threadIdx.x & 63, wave = 12 + (threadIdx.x >> 6);
#endif
    after_fill();  // (the caller's own requests: behind the first operands — loads return in order —, with kDepth steps to arrive)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < n4; ++k)
    {
        const int u = k % kDepth;
        if (k % 8 == 1) BLEND_LAP(5 + k / 8);  // (timing build: the second step of every eight — the first one's operands have arrived)
        const float4 a = ab[u];
        float4 b[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) b[c] = bb[c][u];
        if (k + kDepth < n4)
        {
#if DDGI_IRR_EXP != 2  // (timing experiments, wrong tiles: 1 = one colour channel's records for all three chains, 2 = the weights of the first steps for all)
            ab[u] = pa[static_cast<size_t>(k + kDepth) * 64];
#endif
#pragma unroll
            for (int c = 0; c < (DDGI_IRR_EXP == 1 ? 1 : 3); ++c) bb[c][u] = pb[c][static_cast<size_t>(k + kDepth) * 64];
        }
        __builtin_amdgcn_sched_barrier(0);  // (the requests stay kDepth steps ahead of their use: the scheduler would sink them to save registers)
#if DDGI_IRR_ORDER == 0
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b[c].x, acc[c], 0, 0, 0);
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b[c].y, acc[c], 0, 0, 0);
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b[c].z, acc[c], 0, 0, 0);
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b[c].w, acc[c], 0, 0, 0);
#else
        // a chain's four links of this step back to back: a dependent MFMA takes its accumulator from the one before it; three
        // chains taking turns read theirs from the register file every time
#pragma unroll
        for (int c = 0; c < 3; ++c)
        {
            acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b[c].x, acc[c], 0, 0, 0);
            acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b[c].y, acc[c], 0, 0, 0);
            acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b[c].z, acc[c], 0, 0, 0);
            acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b[c].w, acc[c], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);  // (the scheduler would interleave the chains again)
        }
#endif
        __builtin_amdgcn_sched_barrier(0);
        if (k == n4 - kDepth)  // the last operands have been requested: the ring's registers come free from here on
        {
            at_tail();
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

template <int kDepth, class AfterFill, class AtTail>
DDGI_D void blend_contract3(const float* __restrict__ wa, const float* __restrict__ vb, size_t b_stride, int q_pairs, f16v (&acc)[3], AfterFill&& after_fill, AtTail&& at_tail)
{
    // (the stand-alone irradiance kernel only: the merged small-grid kernel shares its CUs with depth workgroups and keeps the loop's
    // smaller register footprint)
    if (kDepth >= 8 && q_pairs == 128)  // (wave-uniform: 256 rays per probe, the common case)
    {
        blend_contract3_unrolled<kDepth, 32>(wa, vb, b_stride, acc, after_fill, at_tail);
        return;
    }
    if (kDepth >= 8 && q_pairs == 256)  // (512 rays: BASELINE's C4)
    {
        blend_contract3_unrolled<kDepth, 64>(wa, vb, b_stride, acc, after_fill, at_tail);
        return;
    }
    after_fill();
#pragma unroll
    for (int c = 0; c < 3; ++c) acc[c] = f16v{0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    const float4* __restrict__ pa = reinterpret_cast<const float4*>(wa);
    const float4* __restrict__ pb[3] = {reinterpret_cast<const float4*>(vb), reinterpret_cast<const float4*>(vb + b_stride), reinterpret_cast<const float4*>(vb + 2 * b_stride)};
    const int n4 = q_pairs / 4;
    // Operands are requested kDepth float4 (4 ray pairs each = 12 MFMAs = 768 matrix-pipe cycles) ahead of the MFMAs that consume
    // them: an irradiance workgroup's two waves sit alone on their SIMDs, so the prefetch distance is all that hides the
    // records' way from HBM (two float4 ahead, the kernel waited a microsecond per 8 ray pairs: 38 us, of which 10 are MFMAs).
    // (kDepth = 8 costs 128 VGPRs: for the stand-alone irradiance kernel; the merged small-grid kernel, whose workgroups
    // share a CU with depth workgroups, keeps 2)
    float4 ab[kDepth], bb[3][kDepth];
#pragma unroll
    for (int u = 0; u < kDepth; ++u)
    {
        const size_t at = static_cast<size_t>(min(u, n4 - 1)) * 64;
        ab[u] = pa[at];
#pragma unroll
        for (int c = 0; c < 3; ++c) bb[c][u] = pb[c][at];
    }
    for (int k = 0; k < n4; k += kDepth)
    {
#pragma unroll
        for (int u = 0; u < kDepth; ++u)
        {
            const float4 a = ab[u];
            float4 b[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) b[c] = bb[c][u];
            {
                const size_t at = static_cast<size_t>(min(k + kDepth + u, n4 - 1)) * 64;  // past the end: re-read the last one (dropped)
                ab[u] = pa[at];
#pragma unroll
                for (int c = 0; c < 3; ++c) bb[c][u] = pb[c][at];
            }
            if (k + u < n4)  // wave-uniform
            {
#pragma unroll
                for (int c = 0; c < 3; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b[c].x, acc[c], 0, 0, 0);
#pragma unroll
                for (int c = 0; c < 3; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b[c].y, acc[c], 0, 0, 0);
#pragma unroll
                for (int c = 0; c < 3; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b[c].z, acc[c], 0, 0, 0);
#pragma unroll
                for (int c = 0; c < 3; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b[c].w, acc[c], 0, 0, 0);
            }
        }
    }
    at_tail();
}

// Epilogue staging: a wave parks its accumulator tile in LDS as [texel row][column] (row stride 33 words: conflict-free
// for a fixed column) and then walks it with lanes = texels, so that a probe's texels are written as runs of
// consecutive addresses and everything that depends on the texel only (wrap destinations, weight sum) is set up once.
constexpr int kStageStride = 33;
template <int kStride = kStageStride>
DDGI_D void stage_tile(float* stage, const f16v& acc, int col, int half)
{
#pragma unroll
    for (int r = 0; r < 16; ++r) stage[((r & 3) + 8 * (r >> 2) + 4 * half) * kStride + col] = acc[r];  // C/D map of the 32x32 MFMA
}

// Does any of the wave's sums lie outside pm::div_prepared's numerator domain (zero, or 2^-100 .. 2^60)?  Sums are >= +0 (weights
// and records are); anything else — records the engine did not write — counts as outside.  Wave-uniform result.
DDGI_D bool sums_outside_div_domain(const f16v& acc)
{
    pm::DivDomainCheck c;
#pragma unroll
    for (int r = 0; r < 16; ++r) c.add(acc[r]);
    return __builtin_amdgcn_ballot_w64(c.outside()) != 0ull;
}

// depth: workgroup = 7 waves, wave m = depth tile m of the 16 probes of a group; B columns = moment * 16 + probe
struct DepthShared
{
    float4 b_stage[2][256];  // the group's records, 16 ray pairs (256 float4) at a time, double buffered: all seven waves
                             // consume the same B operand — fetched from L2 once per workgroup, not once per wave
    float stage_all[kBlendWaves][32 * kStageStride];
    uint32_t slot_sh[16];  // tile slots of the group's probes (through LDS, not readlane: the epilogue runs under a partial exec mask)
};
constexpr int kIrrStageStride = 97;  // [texel row][3 probe + channel]: a texel's r, g, b side by side; odd: conflict-free for a fixed column
struct IrrShared
{
    float stage_all[kIrrMTiles][32 * kIrrStageStride];
    uint32_t slot_sh[2][32];  // by task parity: written before the contraction, while the other wave may still read the previous task's
    uint32_t unsafe[2];       // per wave: a sum of the group lies outside pm::div_prepared's domain
};
union BlendShared
{
    DepthShared dep;
    IrrShared irr;
};

// depth groups: block = first_block + k * n_blocks handles groups k
DDGI_D void blend_depth_role(const BlendArgs& A, const float* __restrict__ rad_dd, const float* __restrict__ w_tiles, const float* __restrict__ w_sum, DepthShared& sh,
                             uint32_t first_task, uint32_t task_stride)
{
    static_assert(kRecRayPad % 32 == 0, "the contraction consumes 16 ray pairs per chunk");
    auto& stage_all = sh.stage_all;
    auto& slot_sh = sh.slot_sh;
    auto& b_stage = sh.b_stage;
    const GridK& G = A.grid;
    const int n_pad = static_cast<int>(rec_ray_pad(static_cast<uint32_t>(G.n))), q_pairs = n_pad / 2;
    const int mt = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float* stage = stage_all[mt];
    const float hyst = G.hysteresis;
    const uint32_t n_tasks = (A.n_local_probes + 15u) / 16u;
    // epilogue role of this lane: texel c = 32 mt + (lane >> 1), moment lane & 1
    const int trow = lane >> 1, ch = lane & 1, c = mt * 32 + trow;
    const bool texel_valid = c < kDepInterior;
    int dst[4] = {0, -1, -1, -1};
    float sw = 0.0f;
    if (texel_valid)
    {
        blend_destinations(1 + c % (kDepTile - 2), 1 + c / (kDepTile - 2), kDepTile, dst);
        sw = w_sum[c];
    }
    for (uint32_t task = first_task; task < n_tasks; task += task_stride)
    {
        // ---- contraction: acc = W tile x records, ray pairs in order ----
        f16v acc = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        {
            const float4* __restrict__ pa = reinterpret_cast<const float4*>(w_tiles + static_cast<size_t>(mt) * n_pad * 32) + lane;
            const float4* __restrict__ gb = reinterpret_cast<const float4*>(rad_dd + static_cast<size_t>(task) * n_pad * 32);
            const int n_chunks = q_pairs / 16, n4 = q_pairs / 4;  // 16 ray pairs = 4 float4 per lane per chunk
            const bool loader = threadIdx.x < 256;
            float4 b_next = loader ? gb[threadIdx.x] : float4{0.0f, 0.0f, 0.0f, 0.0f};
            float4 a_cur[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) a_cur[u] = pa[static_cast<size_t>(min(u, n4 - 1)) * 64];
            for (int ck = 0; ck < n_chunks; ++ck)
            {
                if (loader) b_stage[ck & 1][threadIdx.x] = b_next;
                if (loader && ck + 1 < n_chunks) b_next = gb[static_cast<size_t>(ck + 1) * 256 + threadIdx.x];
                float4 a_next[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) a_next[u] = pa[static_cast<size_t>(min(4 * (ck + 1) + u, n4 - 1)) * 64];
                __syncthreads();  // chunk ck is staged; (buffer (ck+1)&1 is free: every wave has passed this barrier since it read it)
#pragma unroll
                for (int u = 0; u < 4; ++u)
                {
                    const float4 bv = b_stage[ck & 1][u * 64 + lane];
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[u].x, bv.x, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[u].y, bv.y, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[u].z, bv.z, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[u].w, bv.w, acc, 0, 0, 0);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) a_cur[u] = a_next[u];
            }
        }
        __syncthreads();  // (the previous task's staging has been read; the last chunk's too)
        stage_tile(stage, acc, lane & 31, lane >> 5);
        // tile slot of every probe of the group, once per task (two integer divisions each)
        if (threadIdx.x < 16) slot_sh[threadIdx.x] = static_cast<uint32_t>(blend_tile_slot(G, min(task * 16u + threadIdx.x, A.n_local_probes - 1u)));
        __syncthreads();
        const uint32_t np = min(16u, A.n_local_probes - task * 16u);
        if (texel_valid)
        {
            float old[16];  // all 16 probes' old texels in flight at once
#pragma unroll
            for (uint32_t p = 0; p < 16u; ++p)
                old[p] = p < np ? A.depth_old[static_cast<size_t>(slot_sh[p]) * (kDepTile * kDepTile * 2) + dst[0] * 2 + ch] : 0.0f;
#pragma unroll
            for (uint32_t p = 0; p < 16u; ++p)
                if (p < np)
                {
                    float* tile = A.depth + static_cast<size_t>(slot_sh[p]) * (kDepTile * kDepTile * 2);
                    const float s = stage[trow * kStageStride + ch * 16 + static_cast<int>(p)];
                    float res = 0.0f;
                    if (sw > 1e-6f) res = s / sw;
                    const float out = gl_mix(old[p], res, hyst);
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (dst[k] >= 0) tile[dst[k] * 2 + ch] = out;
                }
        }
    }
}

// depth, 256 rays per probe: ONE persistent workgroup per CU.  Its seven contraction waves keep their whole weight tile —
// 32 texel rows x 256 rays = 128 VGPRs — in registers across the groups they handle, so a group costs its own records and
// nothing else (the kernel above re-reads the 7 x 32 KB of tiles from L2 for every 32 KB of records).  With one workgroup
// per CU nothing else hides latency, so the work is pipelined by hand, one barrier per group:
//   contraction waves   contract this group out of LDS (two whole-group record buffers) and park the accumulators in LDS
//                       (two staging sets)
//   five service waves  meanwhile fetch the NEXT group's records (32 KB) into the other buffer, and four of them turn the
//                       PREVIOUS group's accumulators into texels: lanes = 64 consecutive texels of a tile, border texels
//                       computed from their octahedral-wrap source's sums (the same arithmetic on the same values), every
//                       load / store instruction moves 512 consecutive bytes.  (Storing from the contraction waves —
//                       4-byte stores, lanes = interior texels, borders as extra partial stores — took longer than the
//                       contraction itself: 40 us of the kernel's 80 on 16 384 probes.)
// A texel is mixed with the old value AT ITS OWN PLACE (the update may run in place, and a border's source is another lane's
// output): for a border texel that is its source's old value because every tile this engine writes has its borders equal
// to their sources, and fresh tiles are zero — tiles brought in through ddgi_bind_textures must keep that (ddgi_probe.h).
// One step of a wave's MFMA chain, followed by wait states.  A wave whose next instruction is an MFMA that waits — for the
// previous link of its chain, or for the matrix pipe while the SIMD's other chain has it — holds the SIMD's vector issue port while
// it waits: a third wave on that SIMD gets to issue next to nothing, whatever its priority (tools/blend_laps.py: texel waves that
// had nothing to do in a stage still reached its barrier 16 000 cycles late; tools/microbench/mfma_valu_coissue.hip: a chain written
// as a loop, whose branch opens a gap after every MFMA, lets a third wave issue every 22 cycles).  So the chain steps aside by
// itself: after an MFMA the wave idles in s_nop — which holds nothing — for most of the 128 clocks until its next MFMA can go
// (two chains share a SIMD's pipe, 64 clocks each; a wait state is 4 clocks).  The MFMA is inline assembly so that the wait states stay behind it; the last
// ones also cover the 18 wait states the hardware wants between a 16-pass MFMA and a read of its result by another unit.
#ifndef DDGI_BLEND_PACE
#define DDGI_BLEND_PACE 18  // wait states (4 clocks each) behind an MFMA of a paced chain; 0: no chain is paced
#endif
#define DDGI_STR2(x) #x
#define DDGI_STR(x) DDGI_STR2(x)
template <bool kPaced>
DDGI_D f16v mfma_step(float a, float b, f16v acc)
{
#if DDGI_BLEND_PACE == 0
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
#else
    if (!kPaced) return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    static_assert(DDGI_BLEND_PACE >= 18 && DDGI_BLEND_PACE <= 32, "at least the 18 wait states between a 16-pass MFMA and a read of its result; two s_nop");
    // (the compiler does not know that this is an MFMA: the two wait states in front are the ones it would put between a VALU write
    // of an operand — the accumulator's zeroes — and the MFMA that reads it)
    asm volatile("s_nop 1\n"
                 "v_mfma_f32_32x32x2_f32 %0, %1, %2, %0\n"
                 "s_nop 15\n"
                 "s_nop " DDGI_STR(DDGI_BLEND_PACE - 17)
                 : "+v"(acc)
                 : "v"(a), "v"(b));
    return acc;
#endif
}

// the same with kWaits wait states (4 clocks each, 18 .. 32) behind the MFMA; 0: a plain MFMA
#ifndef DDGI_IRR12_PACE
#define DDGI_IRR12_PACE 0  // (measured: 18 .. 32 wait states behind every MFMA of the 12-wave irradiance kernel cost 8 %: its waves' loads are prefetches, nothing waits for them)
#endif
template <int kWaits>
DDGI_D f16v mfma_step_waits(float a, float b, f16v acc)
{
    if constexpr (kWaits == 0)
        return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    else
    {
        static_assert(kWaits >= 18 && kWaits <= 32, "at least the 18 wait states between a 16-pass MFMA and a read of its result; two s_nop");
        asm volatile("s_nop 1\n"
                     "v_mfma_f32_32x32x2_f32 %0, %1, %2, %0\n"
                     "s_nop 15\n"
                     "s_nop %3"
                     : "+v"(acc)
                     : "v"(a), "v"(b), "n"(kWaits - 17));
        return acc;
    }
}

constexpr int kResN4 = 32;                          // float4 per lane of a resident tile: 128 ray pairs
constexpr int kResGroupF4 = kResN4 * 64;            // float4 of one group's records (16 probes x 2 moments x 256 rays)
constexpr int kResServiceWaves = 5;
constexpr int kResWaves = kBlendWaves + kResServiceWaves;
struct DepthResShared
{
    float4 b_all[2][kResGroupF4];
    float stage_all[2][kBlendWaves][32 * kStageStride];
    uint32_t slots[2][16];    // tile slots of a group's probes (two integer divisions each: computed once, by the service wave with time to spare)
    uint32_t unsafe[2][8];    // per contraction wave: a sum of the group lies outside pm::div_prepared's domain
};
DDGI_D void blend_depth_resident(const BlendArgs& A, const float* __restrict__ rad_dd, const float* __restrict__ w_tiles, const float* __restrict__ w_sum, DepthResShared& sh,
                                 uint32_t first_task, uint32_t task_stride)
{
    const GridK& G = A.grid;
    constexpr int n_pad = 256;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t n_tasks = (A.n_local_probes + 15u) / 16u;
    const uint32_t my_tasks = first_task < n_tasks ? (n_tasks - first_task + task_stride - 1u) / task_stride : 0u;
    BLEND_LAP(22);  // (entry)
    if (wave < kBlendWaves)
    {
        // ================= contraction waves: wave = depth tile =================
        float4 a_res[kResN4];
        {
            const float4* __restrict__ pa = reinterpret_cast<const float4*>(w_tiles + static_cast<size_t>(wave) * n_pad * 32) + lane;
#pragma unroll
            for (int k = 0; k < kResN4; ++k) a_res[k] = pa[static_cast<size_t>(k) * 64];
        }
        BLEND_LAP(0);
        __syncthreads();
        BLEND_LAP(1);
        for (uint32_t it = 0; it <= my_tasks; ++it)
        {
            if (it < my_tasks)
            {
                const int cur = static_cast<int>(it & 1u);
                f16v acc = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
                auto contract = [&](auto paced) {
#pragma unroll
                    for (int k = 0; k < kResN4; ++k)
                    {
                        const float4 bv = sh.b_all[cur][k * 64 + lane];
                        acc = mfma_step<decltype(paced)::value>(a_res[k].x, bv.x, acc);
                        acc = mfma_step<decltype(paced)::value>(a_res[k].y, bv.y, acc);
                        acc = mfma_step<decltype(paced)::value>(a_res[k].z, bv.z, acc);
                        acc = mfma_step<decltype(paced)::value>(a_res[k].w, bv.w, acc);
                    }
                };
#ifndef DDGI_BLEND_PACE_ALL
#define DDGI_BLEND_PACE_ALL 0
#endif
                if (DDGI_BLEND_PACE_ALL || wave == 3)  // the one contraction wave of the texel waves' SIMD
                    contract(std::true_type{});
                else
                    contract(std::false_type{});
                stage_tile(sh.stage_all[cur][wave], acc, ((lane & 15) << 1) | ((lane >> 4) & 1), lane >> 5);  // column = 2 probe + moment: a texel's pair is adjacent
                const bool outside = sums_outside_div_domain(acc);
                if (lane == 0) sh.unsafe[cur][wave] = outside ? 1u : 0u;
            }
            BLEND_LAP(2 + 2 * it);
            __syncthreads();
            BLEND_LAP(3 + 2 * it);
        }
    }
    else
    {
        // ================= service waves =================
        // What shapes this (tools/blend_laps.py, tools/microbench/mfma_valu_coissue.hip): a wave whose next instruction is an MFMA
        // that waits — for the previous link of its chain, or for the matrix pipe — holds its SIMD's vector issue port while it
        // waits.  Beside the TWO contraction waves of SIMDs 0-2 a third wave issues next to nothing until they are done, whatever
        // its priority (a wave with nothing to do in a stage reached the barrier 16 000 cycles late); beside the ONE contraction
        // wave of SIMD 3 (waves go round the SIMDs: tile 3, service waves 0 and 4) it runs at half speed, and that SIMD's matrix
        // pipe is idle for half of every stage anyway.  So:
        //   service waves 0, 4   (SIMD 3) do everything: they turn the previous group's sums into texels, two quarters of the 16x16
        //                        tile each; compute tile slots (integer divisions); and fetch records TWO groups ahead — at the top
        //                        of a stage they park the group they hold in registers (requested at the end of the previous
        //                        stage), at its end they request the one after it, so no stage waits for HBM.  Their SIMD's contraction wave (tile 3) runs
        //                        its chain with wait states behind every MFMA (mfma_step) so that they get to issue all along.
        //   service waves 1-3    (SIMDs 0-2) idle; in the last stage, with no contraction running, 1 and 2 take a quarter each.
        const int sw_id = wave - kBlendWaves;
        const float hyst = G.hysteresis;
        // a quarter of the 16x16 tile: lane's output texel e, the staging offset of the interior texel it takes its sums from, and
        // that texel's weight sum (or +inf where that is ~0: quotient +0, the sums are >= +0)
        struct Quarter
        {
            int e2, stage_off;
            pm::DivBy by;
        };
        auto quarter_of = [&](int q) {
            const int e = 64 * q + lane, tx = e & (kDepTile - 1), ty = e / kDepTile;
            int sx = tx, sy = ty;
            if (tx == 0 || ty == 0 || tx == kDepTile - 1 || ty == kDepTile - 1) border_source(tx, ty, kDepTile, sx, sy);
            const int c = (sy - 1) * (kDepTile - 2) + (sx - 1);
            const float sw = w_sum[c];
            return Quarter{e * 2, (c >> 5) * (32 * kStageStride) + (c & 31) * kStageStride, pm::div_by(sw > 1e-6f ? sw : __builtin_inff())};
        };
        // the texels of group (stage - 1), quarter Q
        auto texels_of = [&](uint32_t it, const Quarter& Q) {
            const uint32_t prev = (it - 1u) & 1u;
            const uint32_t task = first_task + (it - 1u) * task_stride;
            const float* __restrict__ st = &sh.stage_all[prev][0][0];
            const uint32_t np = min(16u, A.n_local_probes - task * 16u);
            const uint32_t my_slot = sh.slots[prev][lane & 15];
            uint32_t outside = A.force_division;
#pragma unroll
            for (int w = 0; w < kBlendWaves; ++w) outside |= sh.unsafe[prev][w];
            // A full group (all but possibly the last) runs as straight-line code: with a branch per probe the compiler no longer
            // knows how many memory operations are in flight and waits for ALL of them — each probe's store included — before the
            // next probe's texel.  Few instructions: the tile slots come from a fetch wave, a texel's two sums sit side by side in
            // the staging, the quotients are pm::div_prepared2's (the contraction waves checked the group's sums against its domain).
            auto texels = [&](auto full, auto prepared) {
                constexpr bool kFull = decltype(full)::value, kPrepared = decltype(prepared)::value;
                pm::f2v old[16];
#pragma unroll
                for (uint32_t p = 0; p < 16u; ++p)
                {
                    const float* tile = A.depth_old + static_cast<size_t>(__builtin_amdgcn_readlane(my_slot, p)) * (kDepTile * kDepTile * 2);
                    old[p] = (kFull || p < np) ? *reinterpret_cast<const pm::f2v*>(tile + Q.e2) : pm::f2v{0.0f, 0.0f};
                }
#pragma unroll
                for (uint32_t p = 0; p < 16u; ++p)
                    if (kFull || p < np)  // (wave-uniform)
                    {
                        float* tile = A.depth + static_cast<size_t>(__builtin_amdgcn_readlane(my_slot, p)) * (kDepTile * kDepTile * 2);
                        const pm::f2v s2 = {st[Q.stage_off + 2 * static_cast<int>(p)], st[Q.stage_off + 2 * static_cast<int>(p) + 1]};
                        const pm::f2v r = kPrepared ? pm::div_prepared2(s2, Q.by) : pm::f2v{s2.x / Q.by.d, s2.y / Q.by.d};
                        *reinterpret_cast<pm::f2v*>(tile + Q.e2) = pm::f2v{gl_mix(old[p].x, r.x, hyst), gl_mix(old[p].y, r.y, hyst)};
                    }
            };
            if (np == 16u && __builtin_amdgcn_readfirstlane(outside) == 0u)
                texels(std::true_type{}, std::true_type{});
            else if (np == 16u)
                texels(std::true_type{}, std::false_type{});
            else
                texels(std::false_type{}, std::false_type{});
        };
        if (sw_id == 0 || sw_id == 4)
        {
            // ---- texel waves: 128 threads, 16 float4 of a group's records each (2048 per group) ----
            __builtin_amdgcn_s_setprio(2);
            constexpr int kFetchLoads = kResGroupF4 / 128;
            const int ft = (sw_id >> 2) * 64 + lane;
            typedef float f4v __attribute__((ext_vector_type(4)));  // (a plain vector type, initialised: as an array of float4 the registers-across-stages end up in scratch)
            f4v r[kFetchLoads];
#pragma unroll
            for (int k = 0; k < kFetchLoads; ++k) r[k] = f4v{0.0f, 0.0f, 0.0f, 0.0f};
            auto group_records = [&](uint32_t g) { return reinterpret_cast<const f4v*>(rad_dd + static_cast<size_t>(first_task + g * task_stride) * n_pad * 32) + ft; };
#define DDGI_FETCH(g)                                                                       \
    do {                                                                                    \
        const f4v* __restrict__ gb_ = group_records(g);                                     \
        _Pragma("unroll") for (int k = 0; k < kFetchLoads; ++k) r[k] = gb_[k * 128];         \
    } while (0)
#define DDGI_PARK(buf)                                                                      \
    do {                                                                                    \
        f4v* park_ = reinterpret_cast<f4v*>(&sh.b_all[buf][0]) + ft;                        \
        _Pragma("unroll") for (int k = 0; k < kFetchLoads; ++k) park_[k * 128] = r[k];       \
    } while (0)
            if (my_tasks > 0u) DDGI_FETCH(0u);  // (first of all: the quarters' set-up below waits for a load of its own)
            const Quarter Qa = quarter_of(sw_id == 0 ? 0 : 2), Qb = quarter_of(sw_id == 0 ? 1 : 3);
            if (my_tasks > 0u) DDGI_PARK(0);
            if (my_tasks > 1u) DDGI_FETCH(1u);
            BLEND_LAP(0);
            __syncthreads();
            BLEND_LAP(1);
            for (uint32_t it = 0; it <= my_tasks; ++it)
            {
                if (it + 1u < my_tasks) DDGI_PARK((it + 1u) & 1u);  // group it + 1, requested in the previous stage; its buffer was read last in stage it - 1
                if (sw_id == 0 && it < my_tasks && lane < 16)  // the group being contracted now: its texels are written in the next stage
                    sh.slots[it & 1u][lane] = static_cast<uint32_t>(blend_tile_slot(G, min((first_task + it * task_stride) * 16u + static_cast<uint32_t>(lane), A.n_local_probes - 1u)));
                if (it >= 1u)
                {
                    texels_of(it, Qa);
                    if (it < my_tasks) texels_of(it, Qb);  // (in the last stage: an idle wave's)
                }
                if (it + 2u < my_tasks) DDGI_FETCH(it + 2u);  // (after the texels: the registers are free again; it travels while the stage ends and the next begins)
                BLEND_LAP(2 + 2 * it);
                __syncthreads();
                BLEND_LAP(3 + 2 * it);
            }
#undef DDGI_FETCH
#undef DDGI_PARK
            return;
        }
        // ---- service waves 1-3: nothing to do beside two contraction waves; in the last stage 1 and 2 take a quarter each ----
        const Quarter Ql = quarter_of(sw_id == 1 ? 1 : 3);
        BLEND_LAP(0);
        __syncthreads();
        BLEND_LAP(1);
        for (uint32_t it = 0; it <= my_tasks; ++it)
        {
            if (it == my_tasks && it >= 1u && sw_id <= 2) texels_of(it, Ql);
            BLEND_LAP(2 + 2 * it);
            __syncthreads();
            BLEND_LAP(3 + 2 * it);
        }
    }
}

// irradiance: workgroup = 2 waves, wave m = irradiance tile m, the three colour channels one after the other, for the
// 32 probes of a group; B columns = probe.  The epilogue joins a texel's three channels: rgba texels leave as 16-byte
// stores.  (Six waves per group — one per (channel, tile) — were slower: 55 us against 38 us on 16 384 probes; the
// contraction is bound by requests in flight to L2 / HBM, not by the matrix pipe.)
constexpr int kIrrWaves = kIrrMTiles;
template <int kDepth>
DDGI_D void blend_irr_role(const BlendArgs& A, const float* __restrict__ rad_rgb, const float* __restrict__ w_tiles, const float* __restrict__ w_sum, IrrShared& sh,
                           uint32_t first_task, uint32_t task_stride)
{
    auto& stage_all = sh.stage_all;
    auto& slot_sh = sh.slot_sh;
    const GridK& G = A.grid;
    const int n_pad = static_cast<int>(rec_ray_pad(static_cast<uint32_t>(G.n))), q_pairs = n_pad / 2;
    const int mi = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float hyst = G.hysteresis;
    const uint32_t n_tasks = (A.n_local_probes + 31u) / 32u;
#ifdef DDGI_BLEND_LAPS
    const int wave = 12 + mi;
#endif
    BLEND_LAP(22);
    // epilogue role: lane = output texel e of the 8x8 tile (borders included: computed from their octahedral-wrap source's
    // sums — the same arithmetic on the same values), wave mi takes the group's probes of parity mi: a probe's tile is one
    // 1 KB load and one 1 KB store.  (Old value at the texel's own place: see blend_depth_resident.)
    const int e = lane, tx = e & (kIrrTile - 1), ty = e / kIrrTile;
    int sx = tx, sy = ty;
    if (tx == 0 || ty == 0 || tx == kIrrTile - 1 || ty == kIrrTile - 1) border_source(tx, ty, kIrrTile, sx, sy);
    const int c = (sy - 1) * (kIrrTile - 2) + (sx - 1);
    const float sw = w_sum[kDepInterior + c];  // (requested here, used in the epilogue: nothing before the first operands waits for it)
    const float* stage_src = &stage_all[c >> 5][(c & 31) * kIrrStageStride];
    constexpr uint32_t kBatch = 8;        // old tiles in flight per wave
    constexpr bool kEarly = kDepth >= 8;  // the stand-alone kernel has the registers to fetch a task's first old tiles BEFORE its contraction.  (All
                                          // sixteen: the epilogue shrinks from 6 500 to 4 000 cycles and the contraction grows from 36 000 to 43 000 —
                                          // it is bound by what the memory system delivers, and the old tiles queue up in front of its operands.)
    uint32_t par = 0u;
    for (uint32_t task = first_task; task < n_tasks; task += task_stride, par ^= 1u)
    {
        const float* wa = w_tiles + static_cast<size_t>(kDepMTiles + mi) * n_pad * 32 + lane * 4;
#ifdef DDGI_IRR_ALIAS  // timing experiment (wrong tiles): every group reads the first 8 groups' records — they stay in L2
        const float* vb = rad_rgb + static_cast<size_t>(task & 7u) * 3 * n_pad * 32 + lane * 4;
#else
        const float* vb = rad_rgb + static_cast<size_t>(task) * 3 * n_pad * 32 + lane * 4;
#endif
        const uint32_t* slots = slot_sh[par];
        if (threadIdx.x < 32) slot_sh[par][threadIdx.x] = static_cast<uint32_t>(blend_tile_slot(G, min(task * 32u + threadIdx.x, A.n_local_probes - 1u)));
        BLEND_LAP(0);
        __syncthreads();  // (and: the previous task's staging has been read)
        BLEND_LAP(1);
        auto load_old = [&](uint32_t p0, float4 (&old)[kBatch]) {
#pragma unroll
            for (uint32_t b = 0; b < kBatch; ++b)
                old[b] = *reinterpret_cast<const float4*>(A.irradiance_old + static_cast<size_t>(slots[min(p0 + 2u * b, 31u)]) * (kIrrTile * kIrrTile * 4) + e * 4);
        };
        float4 old[kBatch], old_late[kBatch];  // the wave's first and second eight probes
        f16v acc[3];
        blend_contract3<kDepth>(
            wa, vb, static_cast<size_t>(n_pad) * 32, q_pairs, acc, [&] { if (kEarly) load_old(static_cast<uint32_t>(mi), old); },
            [&] { if (kEarly) load_old(static_cast<uint32_t>(mi) + 2u * kBatch, old_late); });  // (while the last steps' MFMAs run)
        BLEND_LAP(2);
        bool outside = false;
#pragma unroll
        for (int k = 0; k < 3; ++k)
        {
            stage_tile<kIrrStageStride>(stage_all[mi], acc[k], (lane & 31) * 3 + k, lane >> 5);
            outside |= sums_outside_div_domain(acc[k]);
        }
        if (lane == 0) sh.unsafe[mi] = outside ? 1u : 0u;
        __syncthreads();
        BLEND_LAP(3);
        const uint32_t np = min(32u, A.n_local_probes - task * 32u);
        const bool prepared = __builtin_amdgcn_readfirstlane(sh.unsafe[0] | sh.unsafe[1] | A.force_division) == 0u;
        const pm::DivBy by = pm::div_by(sw > 1e-6f ? sw : __builtin_inff());  // (+inf: quotient +0 where the weight sum is ~0; the sums are >= +0)
        // (full groups as straight-line code, pm::div_prepared quotients: see blend_depth_resident)
        auto tiles = [&](auto full, auto prep) {
            constexpr bool kFull = decltype(full)::value, kPrepared = decltype(prep)::value;
#pragma unroll
            for (int half = 0; half < 2; ++half)
            {
                const uint32_t p0 = static_cast<uint32_t>(mi) + static_cast<uint32_t>(half) * 2u * kBatch;
                if (!kFull && p0 >= np) break;
                float4 (&o)[kBatch] = (kEarly && half == 1) ? old_late : old;  // (not early: one set of registers, loaded twice)
                if (!kEarly) load_old(p0, o);
#pragma unroll
                for (uint32_t b = 0; b < kBatch; ++b)
                {
                    const uint32_t p = p0 + 2u * b;
                    if (kFull || p < np)  // (wave-uniform)
                    {
                        const float* sp = stage_src + 3 * static_cast<int>(p);
                        const pm::f2v rg = kPrepared ? pm::div_prepared2(pm::f2v{sp[0], sp[1]}, by) : pm::f2v{sp[0] / by.d, sp[1] / by.d};
                        const float bl = kPrepared ? pm::div_prepared(sp[2], by) : sp[2] / by.d;
                        *reinterpret_cast<float4*>(A.irradiance + static_cast<size_t>(slots[p]) * (kIrrTile * kIrrTile * 4) + e * 4) =
                            float4{gl_mix(o[b].x, rg.x, hyst), gl_mix(o[b].y, rg.y, hyst), gl_mix(o[b].z, bl, hyst), 1.0f};
                    }
                }
            }
        };
        if (np == 32u && prepared)
            tiles(std::true_type{}, std::true_type{});
        else if (np == 32u)
            tiles(std::true_type{}, std::false_type{});
        else
            tiles(std::false_type{}, std::false_type{});
        BLEND_LAP(4);
    }
}

// Many probes: one launch each (the irradiance workgroups are two waves).
__global__ __launch_bounds__(kBlendWaves * 64) void k_probe_blend_depth(const BlendArgs A, const float* __restrict__ rad_dd, const float* __restrict__ w_tiles,
                                                                        const float* __restrict__ w_sum)
{
    __shared__ DepthShared sh;
    blend_depth_role(A, rad_dd, w_tiles, w_sum, sh, blockIdx.x, gridDim.x);
}
__global__ __launch_bounds__(kResWaves * 64) void k_probe_blend_depth_res(const BlendArgs A, const float* __restrict__ rad_dd, const float* __restrict__ w_tiles,
                                                                            const float* __restrict__ w_sum)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char blend_dyn_lds[];
    blend_depth_resident(A, rad_dd, w_tiles, w_sum, *reinterpret_cast<DepthResShared*>(blend_dyn_lds), blockIdx.x, gridDim.x);
}
__global__ __launch_bounds__(kIrrWaves * 64) void k_probe_blend_irr(const BlendArgs A, const float* __restrict__ rad_rgb, const float* __restrict__ w_tiles,
                                                                    const float* __restrict__ w_sum)
{
    __shared__ IrrShared sh;
#ifndef DDGI_IRR_DEPTH
#define DDGI_IRR_DEPTH 8
#endif
    blend_irr_role<DDGI_IRR_DEPTH>(A, rad_rgb, w_tiles, w_sum, sh, blockIdx.x, gridDim.x);
}

// irradiance, 256 rays per probe: ONE persistent 12-wave workgroup per CU (round 5).  The two-wave kernel above leaves a SIMD to one wave — three
// chains in turn, with nothing but its own prefetch ring to hide the records' way from HBM, and every wave streams its weight tile from L2 again
// for every group.  Here a task is a PAIR of groups (64 probes) and a wave is ONE chain: (group of the pair, tile, colour channel) — three waves per
// SIMD take turns at its matrix pipe, each with its own requests in flight; both irradiance weight tiles (64 KB) are resident in LDS for the life
// of the workgroup (the A operand is a conflict-free ds_read_b128 per four MFMAs), so the only global traffic of a chain is its own records — the
// two tiles' waves of a (group, channel) ask for the same lines at the same time, the second finds them in the vector cache.  The epilogue is the
// two-wave kernel's: lanes = the 64 texels of a tile, a probe's tile is one 1 KB load and one 1 KB store; the 12 waves share a pair's 64 probes.
constexpr int kIrr12Waves = 12;
constexpr int kIrr12N4 = 32;  // float4 per lane and stream: 128 ray pairs
struct Irr12Shared
{
    float4 a_tiles[kIrrMTiles][kIrr12N4 * 64];                // the irradiance weight tiles as MFMA A operands (k_blend_weights' layout)
    float stage_all[2][kIrrMTiles][32 * kIrrStageStride];     // [group of the pair][tile][texel row][3 probe + channel]
    uint32_t slot_sh[2][2][32];                               // [task parity][group of the pair][probe]
    uint32_t unsafe[kIrr12Waves];                             // per wave: a sum of its chain lies outside pm::div_prepared's domain
};
DDGI_D void blend_irr12_role(const BlendArgs& A, const float* __restrict__ rad_rgb, const float* __restrict__ w_tiles, const float* __restrict__ w_sum, Irr12Shared& sh,
                             uint32_t first_task, uint32_t task_stride)
{
    const GridK& G = A.grid;
    constexpr int n_pad = 256;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int g = wave / 6, sub = wave % 6, tile = sub / 3, ch = sub % 3;
    const float hyst = G.hysteresis;
    const uint32_t n_groups = (A.n_local_probes + 31u) / 32u, n_tasks = (n_groups + 1u) / 2u;
    BLEND_LAP(22);  // (timing build: entry)
    // the weight tiles, once per workgroup
    {
        // (all of a thread's loads in flight at once: as a loop of load / store pairs every trip waited for its own load — 6 600 cycles from
        // entry to the first barrier in the timing build, tools/irr12_laps.py)
        const float4* __restrict__ src = reinterpret_cast<const float4*>(w_tiles + static_cast<size_t>(kDepMTiles) * n_pad * 32);
        float4* dst = &sh.a_tiles[0][0];
        constexpr int kTotal = kIrrMTiles * kIrr12N4 * 64, kThreads = kIrr12Waves * 64, kTrips = (kTotal + kThreads - 1) / kThreads;
        float4 v[kTrips];
#pragma unroll
        for (int t = 0; t < kTrips; ++t) v[t] = src[min(static_cast<int>(threadIdx.x) + t * kThreads, kTotal - 1)];
#pragma unroll
        for (int t = 0; t < kTrips; ++t)
            if (static_cast<int>(threadIdx.x) + t * kThreads < kTotal) dst[threadIdx.x + t * kThreads] = v[t];
    }
    // epilogue role (blend_irr_role's): lane = output texel e of the 8x8 tile, borders from their octahedral-wrap source's sums
    const int e = lane, tx = e & (kIrrTile - 1), ty = e / kIrrTile;
    int sx = tx, sy = ty;
    if (tx == 0 || ty == 0 || tx == kIrrTile - 1 || ty == kIrrTile - 1) border_source(tx, ty, kIrrTile, sx, sy);
    const int c = (sy - 1) * (kIrrTile - 2) + (sx - 1);
    const float sw = w_sum[kDepInterior + c];
    const float* stage_src = &sh.stage_all[g][c >> 5][(c & 31) * kIrrStageStride];
    constexpr uint32_t kMine = 6;  // probes of the group this wave writes: sub, sub + 6, ... (< 32)
    uint32_t par = 0u;
    for (uint32_t task = first_task; task < n_tasks; task += task_stride, par ^= 1u)
    {
        const uint32_t group = 2u * task + static_cast<uint32_t>(g);
        const bool group_valid = group < n_groups;  // (wave-uniform; the last pair of an odd number of groups has one)
        // the chain's first records: requested before the slots' divisions and the barrier — they need neither
        constexpr int kDepth = 8;
        const float4* __restrict__ pb = reinterpret_cast<const float4*>(rad_rgb + (static_cast<size_t>(group_valid ? group : 0u) * 3 + ch) * n_pad * 32) + lane;
        float4 bb[kDepth];
#pragma unroll
        for (int u = 0; u < kDepth; ++u) bb[u] = pb[static_cast<size_t>(u) * 64];
        if (threadIdx.x < 64)
        {
            const uint32_t gg = 2u * task + (threadIdx.x >> 5);
            sh.slot_sh[par][threadIdx.x >> 5][threadIdx.x & 31] = static_cast<uint32_t>(blend_tile_slot(G, min(gg * 32u + (threadIdx.x & 31u), A.n_local_probes - 1u)));
        }
        BLEND_LAP(0);
        __syncthreads();  // (and: the previous task's staging has been read; first trip: the weight tiles are in place)
        BLEND_LAP(1);
        const uint32_t* slots = sh.slot_sh[par][g];
        float4 old[kMine];
        f16v acc = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        if (group_valid)
        {
            const float4* pa = &sh.a_tiles[tile][lane];
            // this wave's old tiles: behind the first operands (loads return in order), with the whole contraction to arrive
#pragma unroll
            for (uint32_t j = 0; j < kMine; ++j)
                old[j] = *reinterpret_cast<const float4*>(A.irradiance_old + static_cast<size_t>(slots[min(static_cast<uint32_t>(sub) + 6u * j, 31u)]) * (kIrrTile * kIrrTile * 4) + e * 4);
            __builtin_amdgcn_sched_barrier(0);
            float4 a_next = pa[0];
#pragma unroll
            for (int k = 0; k < kIrr12N4; ++k)
            {
                const int u = k % kDepth;
                if (k % 8 == 1) BLEND_LAP(5 + k / 8);  // (timing build: the second step of every eight)
                const float4 a = a_next, b = bb[u];
                if (k + 1 < kIrr12N4) a_next = pa[(k + 1) * 64];
                if (k + kDepth < kIrr12N4) bb[u] = pb[static_cast<size_t>(k + kDepth) * 64];
                __builtin_amdgcn_sched_barrier(0);  // (the requests stay kDepth steps ahead of their use)
                // (three chains share a SIMD's matrix pipe: a wave that stands at an MFMA while the pipe is busy with another wave's holds the
                // SIMD's issue port — the third wave's loads wait behind it; so a chain steps aside in s_nop for the other two's turns: mfma_step)
                acc = mfma_step_waits<DDGI_IRR12_PACE>(a.x, b.x, acc);
                acc = mfma_step_waits<DDGI_IRR12_PACE>(a.y, b.y, acc);
                acc = mfma_step_waits<DDGI_IRR12_PACE>(a.z, b.z, acc);
                acc = mfma_step_waits<DDGI_IRR12_PACE>(a.w, b.w, acc);
                __builtin_amdgcn_sched_barrier(0);
            }
            BLEND_LAP(2);
            stage_tile<kIrrStageStride>(sh.stage_all[g][tile], acc, (lane & 31) * 3 + ch, lane >> 5);
        }
        {
            const bool outside = group_valid && sums_outside_div_domain(acc);
            if (lane == 0) sh.unsafe[wave] = outside ? 1u : 0u;
        }
        __syncthreads();
        BLEND_LAP(3);
        if (!group_valid) continue;  // (wave-uniform; the barrier at the top of the next trip is reached by everyone)
        const uint32_t np = min(32u, A.n_local_probes - group * 32u);
        uint32_t any_unsafe = A.force_division;
#pragma unroll
        for (int w = 0; w < 6; ++w) any_unsafe |= sh.unsafe[6 * g + w];
        const bool prepared = __builtin_amdgcn_readfirstlane(any_unsafe) == 0u;
        const pm::DivBy by = pm::div_by(sw > 1e-6f ? sw : __builtin_inff());  // (+inf: quotient +0 where the weight sum is ~0; the sums are >= +0)
        auto tiles = [&](auto prep) {
            constexpr bool kPrepared = decltype(prep)::value;
#pragma unroll
            for (uint32_t j = 0; j < kMine; ++j)
            {
                const uint32_t p = static_cast<uint32_t>(sub) + 6u * j;
                if (p < np)  // (wave-uniform)
                {
                    const float* sp = stage_src + 3 * static_cast<int>(p);
                    const pm::f2v rg = kPrepared ? pm::div_prepared2(pm::f2v{sp[0], sp[1]}, by) : pm::f2v{sp[0] / by.d, sp[1] / by.d};
                    const float bl = kPrepared ? pm::div_prepared(sp[2], by) : sp[2] / by.d;
                    *reinterpret_cast<float4*>(A.irradiance + static_cast<size_t>(slots[p]) * (kIrrTile * kIrrTile * 4) + e * 4) =
                        float4{gl_mix(old[j].x, rg.x, hyst), gl_mix(old[j].y, rg.y, hyst), gl_mix(old[j].z, bl, hyst), 1.0f};
                }
            }
        };
        if (prepared)
            tiles(std::true_type{});
        else
            tiles(std::false_type{});
        BLEND_LAP(4);
    }
}
__global__ __launch_bounds__(kIrr12Waves * 64) void k_probe_blend_irr12(const BlendArgs A, const float* __restrict__ rad_rgb, const float* __restrict__ w_tiles,
                                                                        const float* __restrict__ w_sum)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char blend_dyn_lds[];
    blend_irr12_role(A, rad_rgb, w_tiles, w_sum, *reinterpret_cast<Irr12Shared*>(blend_dyn_lds), blockIdx.x, gridDim.x);
}

// 256 rays per probe, many probes: ONE launch for the whole blend — a persistent 12-wave workgroup per CU runs its depth groups
// (blend_depth_resident) and then its pairs of irradiance groups (blend_irr12_role) out of the same dynamic LDS; the second phase starts on a CU
// as soon as that CU's depth groups are done (no launch boundary, no drain of the whole chip in between).
union BlendOneShared
{
    DepthResShared dep;
    Irr12Shared irr;
};
__global__ __launch_bounds__(kResWaves * 64) void k_probe_blend_one(const BlendArgs A, const float* __restrict__ rad_rgb, const float* __restrict__ rad_dd,
                                                                      const float* __restrict__ w_tiles, const float* __restrict__ w_sum)
{
    static_assert(kResWaves == kIrr12Waves, "both roles are written for 12 waves");
    extern __shared__ __attribute__((aligned(16))) unsigned char blend_dyn_lds[];
    BlendOneShared& sh = *reinterpret_cast<BlendOneShared*>(blend_dyn_lds);
    blend_depth_resident(A, rad_dd, w_tiles, w_sum, sh.dep, blockIdx.x, gridDim.x);
    __syncthreads();  // (the depth role's last reads of its staging; every wave comes through here: the role's `return`s are its own)
    blend_irr12_role(A, rad_rgb, w_tiles, w_sum, sh.irr, blockIdx.x, gridDim.x);
}

// Few probes (one rank's slab of a sharded grid: fewer depth groups than half the CUs): one launch for both — blocks
// [0, irr_blocks) take irradiance groups with their first two waves (the other five leave at once), the rest take depth
// groups: the two contractions are bound by their own latency there and overlap instead of running one after the other.
// Measured on slabs of C3 (tools/blend_merge_sweep.sh; one launch / two launches): 2 048 probes 27 / 35 us, 4 096 probes 43 / 36 us,
// 8 192 probes 65 / 46 us — since round 3's persistent depth kernel the two launches win from one group per CU on.
__global__ __launch_bounds__(kBlendWaves * 64) void k_probe_blend_mfma(const BlendArgs A, const float* __restrict__ rad_rgb, const float* __restrict__ rad_dd,
                                                                       const float* __restrict__ w_tiles, const float* __restrict__ w_sum, const uint32_t irr_blocks)
{
    __shared__ BlendShared sh;
    if (blockIdx.x < irr_blocks)
    {
        if (threadIdx.x < kIrrWaves * 64) blend_irr_role<2>(A, rad_rgb, w_tiles, w_sum, sh.irr, blockIdx.x, irr_blocks);
    }
    else
        blend_depth_role(A, rad_dd, w_tiles, w_sum, sh.dep, blockIdx.x - irr_blocks, gridDim.x - irr_blocks);
}

// ------------------------------------------------------------------------------------------------
// k_probe_blend — the same blend with one 256-lane workgroup per probe, lane = texel, weights
// evaluated in place and the ray records staged in LDS.  Used for ray counts whose direction table
// does not fit k_blend_weights' LDS, and as the cross-check of the MFMA blend kernels
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
        const uint32_t n_pad = rec_ray_pad(static_cast<uint32_t>(n));

        __syncthreads();  // previous probe's LDS fully consumed
        for (int i = tid; i < n; i += kBlendBlock)
        {
            s_rad[i] = float4{A.rad_rgb[rec_rgb_index(pl, i, n_pad, 0)], A.rad_rgb[rec_rgb_index(pl, i, n_pad, 1)], A.rad_rgb[rec_rgb_index(pl, i, n_pad, 2)],
                              A.rad_dd[rec_dd_index(pl, i, n_pad, 0)]};
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

template <int kMode>  // 0: a texel's floats one by one, 1: one load per texel (aligned buffers), 2: and 32-bit offsets from the buffers' bases (both below 4 GiB)
#ifndef DDGI_SAMPLE_WAVES
#define DDGI_SAMPLE_WAVES 1  // waves per SIMD the register allocation must leave room for.  The pipelined corner loop holds two corners' texels: 162 VGPRs = 3 waves per
                            // SIMD left alone, 7.41 G points/s (one corner at a time: 94 VGPRs, 5 waves, 7.21); forced to 128 VGPRs it spills 38 registers: 4.6 G
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(DDGI_SAMPLE_WAVES, 8))) void k_probe_sample_ddgi(const SampleArgs A)
{
    const uint32_t k = xcd_block(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x;  // (ddgi_device.h: consecutive points of a sorted batch behind ONE L2)
    if (k >= A.n) return;
    const uint32_t i = (A.perm && !(DDGI_SAMPLE_COHERENCE && A.perm_off && *A.perm_off)) ? A.perm[k] : k;
    int cage[8];
    const f3 p{A.pos[3 * i], A.pos[3 * i + 1], A.pos[3 * i + 2]}, nr{A.nrm[3 * i], A.nrm[3 * i + 1], A.nrm[3 * i + 2]};
    f3 out;
    if constexpr (kMode == 2)
        out = diffuse_gi_ddgi_from(A.grid, TilesGlobal32{A.irradiance, A.depth}, p, nr, cage);
    else
        out = diffuse_gi_ddgi<kMode == 1>(A.grid, A.irradiance, A.depth, p, nr, cage);
    A.rgb[3 * i] = out.x, A.rgb[3 * i + 1] = out.y, A.rgb[3 * i + 2] = out.z;
    if (A.cage)
    {
        int32_t* q1 = A.cage + 8 * static_cast<size_t>(i);  // 32 bytes per point: two 16-byte stores where the caller's buffer is 16-byte aligned
        if ((reinterpret_cast<uintptr_t>(A.cage) & 15u) == 0u)
        {
            int4* q = reinterpret_cast<int4*>(q1);
            q[0] = int4{cage[0], cage[1], cage[2], cage[3]}, q[1] = int4{cage[4], cage[5], cage[6], cage[7]};
        }
        else
            for (int c = 0; c < 8; ++c) q1[c] = cage[c];
    }
}

// ---- launchers -----------------------------------------------------------------------------------

// size of the per-update weight tiles (BlendArgs::w; w_sum holds kBlendCols floats)
size_t blend_weights_floats(int n) { return static_cast<size_t>(kBlendMTiles) * rec_ray_pad(static_cast<uint32_t>(n)) * 32; }

// k_blend_weights + k_blend_weight_sums: depend on the frame's rotation only — the engine launches them BEFORE the trace
hipError_t launch_blend_weights(const BlendArgs& args, hipStream_t stream)
{
    if (!args.w || !args.w_sum) return hipSuccess;
    const int q_pairs = static_cast<int>(rec_ray_pad(static_cast<uint32_t>(args.grid.n))) / 2;
    hipLaunchKernelGGL(k_blend_weights, dim3((q_pairs + 3) / 4, kBlendMTiles), dim3(256), 0, stream, args);
    hipLaunchKernelGGL(k_blend_weight_sums, dim3(1), dim3(kBlendCols), 0, stream, args);
    return hipGetLastError();
}

hipError_t launch_probe_blend(const BlendArgs& args, int num_cus, hipStream_t stream)
{
    const int n = args.grid.n;
    if (args.n_local_probes == 0) return hipSuccess;
    if (args.w && args.w_sum)
    {
        const uint32_t dep_tasks = (args.n_local_probes + 15u) / 16u, irr_tasks = (args.n_local_probes + 31u) / 32u;
        const uint32_t dep_blocks = std::min<uint32_t>(dep_tasks, static_cast<uint32_t>(num_cus) * 8u), irr_blocks = std::min<uint32_t>(irr_tasks, static_cast<uint32_t>(num_cus) * 16u);
        if (dep_tasks * 2u <= static_cast<uint32_t>(num_cus) * args.merge_below)  // (default: fewer depth groups than half the CUs)
            hipLaunchKernelGGL(k_probe_blend_mfma, dim3(irr_blocks + dep_blocks), dim3(kBlendWaves * 64), 0, stream, args, args.rad_rgb, args.rad_dd, static_cast<const float*>(args.w),
                               static_cast<const float*>(args.w_sum), irr_blocks);
        else
        {
#ifndef DDGI_IRR12
#define DDGI_IRR12 1  // 256 rays per probe: the 12-wave persistent irradiance kernel (0: the two-wave kernel everywhere)
#endif
#ifndef DDGI_BLEND_ONE
#define DDGI_BLEND_ONE 0  // 1: depth and irradiance of a 256-ray grid in ONE launch (k_probe_blend_one).  Measured (profiles/r05_i_blend_one_ab.txt, C3): the one kernel
                          // takes 64.0 us against 42.6 + 22.4 — and the UPDATE gets slower, 1.558 against 1.543 ms: next frame's k_blend_weights runs on the
                          // preparation stream beside the blend, between two launches it finds CUs at once (28 us), beside one persistent kernel that holds
                          // every CU for the whole blend it waits (53 us), and the next update waits for it.  Off.
#endif
            if (DDGI_BLEND_ONE && DDGI_IRR12 && rec_ray_pad(static_cast<uint32_t>(n)) == 256u)
            {
                hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(k_probe_blend_one), sizeof(BlendOneShared));
                if (e != hipSuccess) return e;
                hipLaunchKernelGGL(k_probe_blend_one, dim3(std::min<uint32_t>(dep_tasks, static_cast<uint32_t>(num_cus))), dim3(kResWaves * 64), sizeof(BlendOneShared), stream, args,
                                   args.rad_rgb, args.rad_dd, static_cast<const float*>(args.w), static_cast<const float*>(args.w_sum));
                return hipGetLastError();
            }
            if (rec_ray_pad(static_cast<uint32_t>(n)) == 256u)  // the weight tiles fit the register file: persistent workgroups, one per CU
            {
                hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(k_probe_blend_depth_res), sizeof(DepthResShared));
                if (e != hipSuccess) return e;
                hipLaunchKernelGGL(k_probe_blend_depth_res, dim3(std::min<uint32_t>(dep_tasks, static_cast<uint32_t>(num_cus))), dim3(kResWaves * 64), sizeof(DepthResShared), stream, args,
                                   args.rad_dd, static_cast<const float*>(args.w), static_cast<const float*>(args.w_sum));
            }
            else
                hipLaunchKernelGGL(k_probe_blend_depth, dim3(dep_blocks), dim3(kBlendWaves * 64), 0, stream, args, args.rad_dd, static_cast<const float*>(args.w),
                                   static_cast<const float*>(args.w_sum));
            if (DDGI_IRR12 && rec_ray_pad(static_cast<uint32_t>(n)) == 256u)
            {
                hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(k_probe_blend_irr12), sizeof(Irr12Shared));
                if (e != hipSuccess) return e;
                hipLaunchKernelGGL(k_probe_blend_irr12, dim3(std::min<uint32_t>((irr_tasks + 1u) / 2u, static_cast<uint32_t>(num_cus))), dim3(kIrr12Waves * 64), sizeof(Irr12Shared), stream, args,
                                   args.rad_rgb, static_cast<const float*>(args.w), static_cast<const float*>(args.w_sum));
            }
            else
                hipLaunchKernelGGL(k_probe_blend_irr, dim3(irr_blocks), dim3(kIrrWaves * 64), 0, stream, args, args.rad_rgb, static_cast<const float*>(args.w),
                                   static_cast<const float*>(args.w_sum));
        }
        return hipGetLastError();
    }
    const size_t lds = (static_cast<size_t>(7) * n + kIrrTile * kIrrTile * 4 + kDepTile * kDepTile * 2) * sizeof(float);
    const uint32_t blocks = std::min<uint32_t>(args.n_local_probes, static_cast<uint32_t>(num_cus) * 8u);
    hipLaunchKernelGGL(k_probe_blend, dim3(blocks), dim3(kBlendBlock), lds, stream, args);
    return hipGetLastError();
}

#ifdef DDGI_BLEND_LAPS
extern "C" int ddgi_debug_blend_laps(unsigned long long* out) { return static_cast<int>(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_blend_laps), sizeof(g_blend_laps))); }
#endif

hipError_t launch_probe_sample_ddgi(const SampleArgs& args, hipStream_t stream)
{
    const unsigned blocks = (args.n + 255u) / 256u;
    if (blocks == 0) return hipSuccess;
    // (a texel as one load needs the tiles' base aligned to a texel: the engine's buffers are; textures bound by the host may not be)
    const bool vec = (reinterpret_cast<uintptr_t>(args.irradiance) & 15u) == 0u && (reinterpret_cast<uintptr_t>(args.depth) & 7u) == 0u;
#ifndef DDGI_SAMPLE_OFF32
#define DDGI_SAMPLE_OFF32 0
#endif
    // (TilesGlobal32 — a texel's address as the buffer's base in scalar registers + a 32-bit offset per lane, the trace kernel's remedy for 64-bit address
    // arithmetic — takes 16 of a corner's 320 instructions away and makes the batch 7 % SLOWER: 0.214 against 0.200 ms per 1.44 M points; off)
    const size_t n_slots = static_cast<size_t>(args.grid.cx) * args.grid.cy * args.grid.cz;
    const bool off32 = DDGI_SAMPLE_OFF32 && vec && n_slots * (kDepTile * kDepTile * 2 * 4) <= 0xffffffffull;  // (the depth tiles are the larger buffer)
    if (off32) hipLaunchKernelGGL(k_probe_sample_ddgi<2>, dim3(blocks), dim3(256), 0, stream, args);
    else if (vec) hipLaunchKernelGGL(k_probe_sample_ddgi<1>, dim3(blocks), dim3(256), 0, stream, args);
    else hipLaunchKernelGGL(k_probe_sample_ddgi<0>, dim3(blocks), dim3(256), 0, stream, args);
    return hipGetLastError();
}

}  // namespace ddgi
