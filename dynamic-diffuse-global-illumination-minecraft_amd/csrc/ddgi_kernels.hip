// ddgi_kernels.hip — hand-written gfx950 (CDNA4, wave64) kernels of the DDGI probe path.
//
//   k_probe_trace_ref   REF mode probe update: one lane = one probe ray, multi-bounce direct light,
//                       one rgba8 texel per ray.  Replaces assets/shaders/probe_pass.comp:main
//                       (253-303) and everything it calls in intersection.glsl.
//   k_probe_sample_ref  REF mode 8-probe-cage sampler.  Replaces get_diffuse_gi / sample_probe
//                       (intersection.glsl:1176-1240, 1306-1409).
//
// All arithmetic is the engine's pinned binary32 arithmetic (DESIGN.md "Arithmetic pinning");
// this file is compiled with -ffp-contract=off so that only the fmaf() written in the source fuse.
#include "ddgi_device.h"
#include "ddgi_sampler.h"

#include <algorithm>

namespace ddgi {

enum : int
{
    kPrimary = 0,
    kFeeler = 1
};

// ------------------------------------------------------------------------------------------------
// k_probe_trace_ref
//
// Mapping: persistent workgroups of 256 lanes; each takes chunks of 256 consecutive local rays
// (for s = 16 one chunk = one probe).  The occupancy bitmap of the baked scene is staged into LDS
// once per workgroup; block types are fetched from global memory (L2) only on a hit.
//
// Control flow: every lane runs a small state machine {primary march, feeler march} over ONE
// shared march loop, so lanes in different bounces / light feelers of their paths still execute
// the same march_step instructions.  A lane whose march ended parks ("waiting") until
// A.wait_threshold lanes of the wave are parked or none is marching; then all parked lanes resolve
// their hit (shading, next ray) together.  Results do not depend on this scheduling.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kTraceBlock) void k_probe_trace_ref(const TraceArgs A)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t s_bits[];
    for (int i = threadIdx.x; i < A.scene.nwords; i += kTraceBlock) s_bits[i] = A.scene.bits[i];
    __syncthreads();

    const GridK& G = A.grid;
    const int rays_per_probe = G.n;
    const uint32_t n_chunks = (A.n_rays + kTraceBlock - 1) / kTraceBlock;
    const float inf = __builtin_inff();

    unsigned long long st_trips = 0, st_lane_steps = 0, st_rounds = 0, st_lane_events = 0;
    for (uint32_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x)
    {
        const uint32_t r = chunk * kTraceBlock + threadIdx.x;  // local ray index
        const bool in_range = r < A.n_rays;

        // local (y, zl, x) probe enumeration -> reference probe index p -> global ray index
        uint32_t global_ray = 0;
        f3 o = mk3(0, 0, 0), d = mk3(1, 0, 0);
        int dst_probe = 0, tile_x = 0, tile_y = 0;
        if (in_range)
        {
            const int pl = static_cast<int>(r / static_cast<uint32_t>(rays_per_probe));
            const int i = static_cast<int>(r) - pl * rays_per_probe;
            const int slab_row = G.czl * G.cx;
            const int y = pl / slab_row;
            const int rem = pl - y * slab_row;
            const int p = y * G.cx * G.cz + G.z0 * G.cx + rem;
            global_ray = static_cast<uint32_t>(p) * static_cast<uint32_t>(rays_per_probe) + static_cast<uint32_t>(i);
            const float4* rec = A.rays + 3 * static_cast<size_t>(r);
            const float4 a = rec[0], b = rec[1], c = rec[2];
            o = mk3(a.x, a.y, a.z);
            d = mk3(b.x, b.y, b.z);
            dst_probe = static_cast<int>(c.x);  // int(probe_info.x), probe_pass.comp:269
            tile_x = static_cast<int>(c.y);
            tile_y = static_cast<int>(c.z);
        }

        uint32_t rng = wang_hash(global_ray);  // p_idx == buffer index (probe_pass.comp:55-57)
        f3 color = mk3(0, 0, 0);
        int bounce = 0;
        int phase = kPrimary;
        f3 hpos = mk3(0, 0, 0), hnrm = mk3(0, 0, 0), hcol = mk3(0, 0, 0), direct = mk3(0, 0, 0);
        int li = 0, nvis = 0;

        March m;
        bool marching = in_range && A.max_bounces > 0;
        bool waiting = false;
        bool hit_block = false;
        if (marching) start_march(m, o, d, A);
        else
        {
            m.ro = m.rd = m.dn = m.inv = m.cc = m.p = mk3(0, 0, 0);
            m.t = 0.0f, m.tl = inf, m.it = 0, m.lid = -1;
        }

        for (;;)
        {
            // ---- shared march loop ----
            int trips = 0;
            for (;;)
            {
                const unsigned long long mb = __ballot(marching);
                if (mb == 0ull) break;
                if (__popcll(__ballot(waiting)) >= A.wait_threshold) break;
                st_trips += 1;
                st_lane_steps += __popcll(mb);
                if (marching)
                {
                    const bool occ = march_step(m, A.scene, s_bits);
                    bool fin = occ | (m.t >= m.tl) | (m.it >= kMarchIters);
                    if (!fin && ((trips & 7) == 7)) fin = march_escaped(m, A.scene);
                    if (fin)
                    {
                        hit_block = occ;
                        marching = false;
                        waiting = true;
                    }
                }
                ++trips;
            }
            if (__ballot(waiting) == 0ull) break;  // nobody marching, nobody waiting: chunk done

            // ---- resolve finished marches (intersect_scene's tail + caller) ----
            st_rounds += 1;
            st_lane_events += __popcll(__ballot(waiting));
            if (waiting)
            {
                waiting = false;
                const bool block_wins = hit_block && (m.t < m.tl);   // temp_isect.t < closest_t
                const bool any_hit = block_wins || (m.tl < inf);    // closest_t < INF
                bool lighting_done = false;
                f3 contribution = mk3(0, 0, 0);

                if (phase == kPrimary)
                {
                    if (!any_hit)
                    {
                        bounce = A.max_bounces;  // probe_pass.comp:288-290 break
                    }
                    else
                    {
                        f3 nraw;
                        float th;
                        if (block_wins)
                        {
                            th = m.t;
                            const f3 cell = cell_id(m.p);
                            const f3 centre = f3{cell.x - 0.5f, cell.y - 0.5f, cell.z - 0.5f};
                            const f3 diff = normalize3(m.p - centre);
                            // axis of the largest |component|, first wins on ties (:1075-1086)
                            f3 n = mk3(0, 0, 0);
                            float best = 0.0f;
                            if (fabsf(diff.x) > best) { best = fabsf(diff.x); n = mk3(gl_sign(diff.x), 0, 0); }
                            if (fabsf(diff.y) > best) { best = fabsf(diff.y); n = mk3(0, gl_sign(diff.y), 0); }
                            if (fabsf(diff.z) > best) { best = fabsf(diff.z); n = mk3(0, 0, gl_sign(diff.z)); }
                            const f3 nn = normalize3(n);
                            const int idx = cell_index(A.scene, static_cast<int>(cell.x), static_cast<int>(cell.y), static_cast<int>(cell.z));
                            const int type = hit_block_type(A.scene, A.scene_id, cell, idx);
                            hcol = block_albedo(m.p, type, nn, A.noise);
                            nraw = nn;
                        }
                        else
                        {
                            th = m.tl;
                            const LightK& L = A.lights[m.lid];
                            const f3 lp{L.pos[0], L.pos[1], L.pos[2]};
                            nraw = ray_at((m.ro - lp) * 10.0f, m.rd * 10.0f, th);  // sphere-space position
                            hcol = mk3(0, 0, 0);  // Q12: unassigned Material, pinned to zero
                        }
                        hnrm = normalize3(nraw);
                        hpos = ray_at(m.ro, m.rd, th) + hnrm * 0.001f;
                        li = 0;
                        nvis = 0;
                        direct = mk3(0, 0, 0);
                        if (A.nl > 0)
                        {
                            phase = kFeeler;
                            const LightK& L = A.lights[0];
                            start_march(m, hpos, normalize3(f3{L.pos[0], L.pos[1], L.pos[2]} - hpos), A);
                            marching = true;
                        }
                        else
                            lighting_done = true;
                    }
                }
                else  // kFeeler: get_direct_lighting's loop body (probe_pass.comp:186-207)
                {
                    const LightK& L = A.lights[li];
                    const f3 lp{L.pos[0], L.pos[1], L.pos[2]};
                    bool early = false;
                    if (any_hit)
                    {
                        const float lambert = gl_clamp(dot3(normalize3(hnrm), normalize3(lp - hpos)), 0.0f, 1.0f);
                        if (!block_wins)
                        {
                            const float dist = length3(lp - hpos);
                            const f3 lc{L.col[0], L.col[1], L.col[2]};
                            direct = direct + div3((lc * lambert) * L.intensity, dist);
                            nvis += 1;
                        }
                        else
                        {
                            contribution = (hcol * 0.2f) * lambert;  // Q10 early return
                            early = true;
                        }
                    }
                    li += 1;
                    if (early) lighting_done = true;
                    else if (li < A.nl)
                    {
                        const LightK& Ln = A.lights[li];
                        start_march(m, hpos, normalize3(f3{Ln.pos[0], Ln.pos[1], Ln.pos[2]} - hpos), A);
                        marching = true;
                    }
                    else
                    {
                        if (nvis != 0) contribution = div3(hcol * direct, static_cast<float>(nvis));
                        lighting_done = true;
                    }
                }

                if (lighting_done)
                {
                    color = color + contribution;
                    bounce += 1;
                    if (bounce < A.max_bounces)
                    {
                        phase = kPrimary;
                        const f3 no = hpos + hnrm * 0.0001f;
                        const f3 nd = hemisphere_dir(hnrm, rng);
                        start_march(m, no, nd, A);
                        marching = true;
                    }
                }
            }
        }

        if (in_range)
        {
            const f3 c = div3(color, static_cast<float>(A.max_bounces));  // Q14: always /max_bounces
            const uint32_t texel = unorm8(c.x) | (unorm8(c.y) << 8) | (unorm8(c.z) << 16) | (255u << 24);
            const size_t dst = static_cast<size_t>(slab_slot(G, dst_probe)) * rays_per_probe + tile_y * G.sx + tile_x;
            A.albedo[dst] = texel;
            A.distance[dst] = 0u;  // `distances` is never assigned (probe_pass.comp:276,302)
        }
    }
    if (A.stats && (threadIdx.x & 63) == 0)
    {
        atomicAdd(&A.stats[0], st_trips);
        atomicAdd(&A.stats[1], st_lane_steps);
        atomicAdd(&A.stats[2], st_rounds);
        atomicAdd(&A.stats[3], st_lane_events);
        atomicAdd(&A.stats[4], 1ull);
    }
}

// ------------------------------------------------------------------------------------------------
// k_probe_sample_ref — one lane per shading point
// ------------------------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void k_probe_sample_ref(const SampleArgs A)
{
    __shared__ float s_unorm[256];
    s_unorm[threadIdx.x] = static_cast<float>(threadIdx.x) / 255.0f;  // blockDim.x == 256
    __syncthreads();
    const uint32_t k = xcd_block(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x;  // (ddgi_device.h: consecutive points of a sorted batch behind ONE L2)
    if (k >= A.n) return;
    const uint32_t i = (A.perm && !(DDGI_SAMPLE_COHERENCE && A.perm_off && *A.perm_off)) ? A.perm[k] : k;
    int cage[8];
    const f3 p{A.pos[3 * i], A.pos[3 * i + 1], A.pos[3 * i + 2]}, nr{A.nrm[3 * i], A.nrm[3 * i + 1], A.nrm[3 * i + 2]};
    const f3 out = A.box ? diffuse_gi_ref<true>(A.grid, A.albedo, p, nr, s_unorm, cage, A.box) : diffuse_gi_ref<false>(A.grid, A.albedo, p, nr, s_unorm, cage, nullptr);
    A.rgb[3 * i] = out.x;
    A.rgb[3 * i + 1] = out.y;
    A.rgb[3 * i + 2] = out.z;
    if (A.cage)  // (8 indices = 32 bytes per point: two 16-byte stores instead of eight 4-byte ones at a 32-byte stride —
    {            //  when the caller's buffer is 16-byte aligned, include/ddgi_probe.h: ddgi_sample_device)
        int32_t* q1 = A.cage + 8 * static_cast<size_t>(i);
        if ((reinterpret_cast<uintptr_t>(A.cage) & 15u) == 0u)
        {
            int4* q = reinterpret_cast<int4*>(q1);
            q[0] = int4{cage[0], cage[1], cage[2], cage[3]}, q[1] = int4{cage[4], cage[5], cage[6], cage[7]};
        }
        else
            for (int c = 0; c < 8; ++c) q1[c] = cage[c];
    }
}

// ------------------------------------------------------------------------------------------------
// k_sample_box_filter — sample_probe for every texel of every tile, once per probe update.
// What sample_probe returns depends on the probe and on the texel (rx, ry) the direction lands in, nothing else: the centre texel
// plus its clipped 5x5 box, divided by the count (intersection.glsl:1213-1239).  get_diffuse_gi asks for it 8 times per shaded
// point — 208 gathers and 624 rgba8 conversions per point, the same few thousand sums over and over.  This kernel evaluates it
// ONCE per texel, in the reference's order of additions (so every value has the bits the per-point evaluation would give), into
// a float4 table (layout below); the sampler then loads one table entry per cage corner (diffuse_gi_ref's `box`).
// One workgroup per tile: the tile's texels are converted once into three float planes in LDS, every lane sums its own box.
// The engine rebuilds the table lazily — the first large sample batch after a probe update pays for it (ddgi_engine.cpp).
// ------------------------------------------------------------------------------------------------
// TABLE LAYOUT (round 4): texel-major, [ry][rx][slab slot].  The 8 corners of a cage are asked for the SAME texel, and two of
// them are x-neighbours — consecutive slab slots: their entries are 32 contiguous bytes, so a point's 8 entries cost about 5
// 64-byte sectors through L2 instead of 8 (tile-major, a corner's entry shared its sector with other TEXELS of its tile, which
// this point never reads).  The build keeps its stores sector-sized: a workgroup takes kTiles consecutive slots and every group
// of kTiles lanes writes the kTiles x 16 bytes of one texel side by side.
template <int kTiles>
__global__ __launch_bounds__(256) void k_sample_box_filter(const GridK G, const uint32_t* __restrict__ albedo, float4* __restrict__ box, uint32_t n_probes)
{
    extern __shared__ __attribute__((aligned(16))) float box_lds[];
    const int n = G.n, s = G.sx, sh = G.sy;
    float* unorm = box_lds;
    float* planes = box_lds + 256;  // [tile][r, g, b][n], tiles kPad words apart
    // (kTiles consecutive lanes sum the SAME texel of kTiles tiles: with the tiles' planes 3 n words apart — a multiple of 32 for any
    // n the engine makes — the four lanes of a texel sat on one LDS bank, a 4-way conflict on every one of the 78 reads per lane;
    // 8 words of padding per tile put the 8 texels x 4 tiles of a half wave on 32 different banks)
    constexpr int kPad = kTiles > 1 ? 32 / kTiles : 0;  // (4 tiles: 8 words, 8 tiles: 4 — a half wave's kTiles x (32 / kTiles) texels on 32 different banks)
    const size_t tile_words = static_cast<size_t>(3) * n + kPad;
    unorm[threadIdx.x] = static_cast<float>(threadIdx.x) / 255.0f;  // blockDim.x == 256
    // a workgroup takes kTiles consecutive TABLE slots (bricks: the four probes of one z-layer of a brick — wherever their tiles lie)
    const uint32_t n_slots = box_slots(G.cx, G.cy, G.cz);
    const uint32_t n_groups = (n_slots + kTiles - 1) / kTiles;
    for (uint32_t g = blockIdx.x; g < n_groups; g += gridDim.x)
    {
        __syncthreads();  // the previous group's planes are no longer read (and the table is written)
        const uint32_t slot0 = g * kTiles;
        int src_slot[kTiles];  // the tiles' slab slots; -1: no such probe (padding of an odd count, or past the end)
#pragma unroll
        for (int w = 0; w < kTiles; ++w) src_slot[w] = slot0 + w < n_slots ? box_slot_to_slab_slot(G.cx, G.cy, G.cz, slot0 + w) : -1;
        for (uint32_t t = threadIdx.x; t < static_cast<uint32_t>(kTiles) * static_cast<uint32_t>(n); t += 256)
        {
            const uint32_t w = t / n, tt = t % n;
            int from = src_slot[0];
#pragma unroll
            for (int q = 1; q < kTiles; ++q) from = w == static_cast<uint32_t>(q) ? src_slot[q] : from;
            if (from < 0) continue;
            const uint32_t v = albedo[static_cast<size_t>(from) * n + tt];
            float* pl = planes + static_cast<size_t>(w) * tile_words;
            pl[tt] = unorm[v & 255u], pl[n + tt] = unorm[(v >> 8) & 255u], pl[2 * n + tt] = unorm[(v >> 16) & 255u];
        }
        __syncthreads();
        for (uint32_t o = threadIdx.x; o < static_cast<uint32_t>(kTiles) * n; o += 256)
        {
            const uint32_t which = o % kTiles, t = o / kTiles;  // kTiles consecutive lanes: one texel of kTiles consecutive slots
            int from = src_slot[0];
#pragma unroll
            for (int q = 1; q < kTiles; ++q) from = which == static_cast<uint32_t>(q) ? src_slot[q] : from;
            if (from < 0) continue;
            const float *pr = planes + static_cast<size_t>(which) * tile_words, *pg = pr + n, *pb = pg + n;
            const f3 v = sample_box_ref(s, sh, static_cast<int>(t % s), static_cast<int>(t / s), [&](int off) { return f3{pr[off], pg[off], pb[off]}; });
            box[box_index(t, slot0 + which, static_cast<uint32_t>(n), n_slots)] = float4{v.x, v.y, v.z, 0.0f};
        }
    }
}

hipError_t launch_sample_box_filter(const GridK& grid, const uint32_t* albedo, float4* box, int num_cus, hipStream_t stream)
{
    const uint32_t n_probes = static_cast<uint32_t>(grid.cx) * grid.cy * grid.cz;
    // tiles per workgroup = consecutive table slots it writes per texel: 8 = a whole brick, one full 128-byte line per texel (up to 512 texels per
    // tile: 50 KB of planes in LDS); 4 = a brick's z-layer, half a line — on a table beyond the Infinity Cache (C4: 1.07 GB) half lines cost the build
    // its write bandwidth
#ifndef DDGI_BOX_BUILD_TILES
#define DDGI_BOX_BUILD_TILES 4  // (8 — a whole brick, full 128-byte lines per texel — measured SLOWER: C3 49.2 against 45.4 us, C4 2.12 against 2.05 ms)
#endif
    const int tiles = (DDGI_BOX_BUILD_TILES == 8 && DDGI_BOX_LAYOUT >= 2 && grid.n <= 512) ? 8 : (grid.n <= 1024 ? 4 : 1);  // (4 tiles' planes: 48 KB at 1024 texels per tile)
    const size_t lds = (256 + (static_cast<size_t>(3) * grid.n + 8) * tiles) * sizeof(float);
    const void* fn = tiles == 8 ? reinterpret_cast<const void*>(k_sample_box_filter<8>) : tiles == 4 ? reinterpret_cast<const void*>(k_sample_box_filter<4>) : reinterpret_cast<const void*>(k_sample_box_filter<1>);
    hipError_t e = ensure_dynamic_lds(fn, static_cast<int>(lds));
    if (e != hipSuccess) return e;
    const uint32_t n_slots = box_slots(grid.cx, grid.cy, grid.cz);
    const uint32_t groups = (n_slots + static_cast<uint32_t>(tiles) - 1u) / static_cast<uint32_t>(tiles);
    const dim3 grid_dim(std::min<uint32_t>(groups, static_cast<uint32_t>(num_cus) * 8u));
    if (tiles == 8) hipLaunchKernelGGL(k_sample_box_filter<8>, grid_dim, dim3(256), lds, stream, grid, albedo, box, n_probes);
    else if (tiles == 4) hipLaunchKernelGGL(k_sample_box_filter<4>, grid_dim, dim3(256), lds, stream, grid, albedo, box, n_probes);
    else hipLaunchKernelGGL(k_sample_box_filter<1>, grid_dim, dim3(256), lds, stream, grid, albedo, box, n_probes);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Grouping a batch of shading points by probe cage (a counting sort by the cage's first probe; order inside and
// between the groups is arbitrary).  Points scattered over the grid share nothing between neighbouring lanes:
// every lane pulls its own 8 tiles — 40 64-byte sectors per point in REF mode — through L2.  Handled cage by cage,
// a group's lanes read the same 8 tiles out of L1.  Results do not depend on the order: each point's output is
// written to its own index.
//   k_sample_keys     key = slab-major slot of the cage's corner-0 probe (n_probes: outside the field); histogram
//   k_sample_bins     every bin reserves a contiguous range of the permutation (one atomic per bin — no scan needed)
//   k_sample_scatter  perm[range start + arrival number] = point
// ------------------------------------------------------------------------------------------------
//
// Round 4: no global atomics.  Device-scope atomics on MI355X are resolved beyond the XCDs' L2s, at ~15 G per second: the
// histogram's and the scatter's 1.44 M atomic adds took 98 + 72 us — more than the sample kernel they were to speed up (141 us).
// Now every workgroup counts its own contiguous run of points in LDS (an LDS atomic returns the point's rank inside its run and
// bin for free), the counts go out as one coalesced row per workgroup, one small kernel turns the table of rows into offsets, and
// the scatter is a plain store.  Bins are COARSE — the cage's slot index shifted so that at most kSampleBins fit the LDS
// histogram (C3: 8 consecutive cages along x per bin): locality is all the grouping is for.
constexpr uint32_t kSampleBins = 8192;   // bins (LDS entries per workgroup: 16-bit counters in k_sample_count — a run holds fewer than 65 536 points — and the 32-bit starts in k_sample_place)
constexpr uint32_t kSampleRuns = 256;    // workgroups = contiguous runs of the batch

DDGI_D uint32_t sample_key(const GridK& G, const float* __restrict__ pos, uint32_t i, uint32_t shift, uint32_t n_bins)
{
    const float side = static_cast<float>(G.side);
    // (grouping only: any point lands in SOME bin; the sample kernels redo get_diffuse_gi's arithmetic exactly)
    const int bx = gl_int(floorf((pos[3 * i] - G.origin[0]) / side)) + G.cx / 2;
    const int by = gl_int(floorf((pos[3 * i + 1] - G.origin[1]) / side)) + G.cy / 2;
    const int bz = gl_int(floorf((pos[3 * i + 2] - G.origin[2]) / side)) + G.cz / 2;
    if (bx >= 0 && bx < G.cx && by >= 0 && by < G.cy && bz >= 0 && bz < G.cz) return static_cast<uint32_t>((bz * G.cy + by) * G.cx + bx) >> shift;
    return n_bins - 1u;  // outside the field: the last bin, on its own
}
// run r = points [r * per_run, (r + 1) * per_run): keys[i], rank[i] (arrival number inside its run and bin), counts[r][bin]
__global__ __launch_bounds__(1024) void k_sample_count(const GridK G, const float* __restrict__ pos, uint32_t n, uint32_t per_run, uint32_t shift, uint32_t n_probes, uint32_t n_bins,
                                                       uint32_t* __restrict__ keys, uint32_t* __restrict__ rank, uint32_t* __restrict__ counts, uint32_t* __restrict__ run_nz)
{
    // 16-bit counters, two to a word (LDS has 32-bit atomics): the add returns the word, the bin's half of it is the rank
    extern __shared__ __attribute__((aligned(16))) uint32_t hist[];
    const uint32_t n_words = (n_bins + 1u) / 2u;
    for (uint32_t b = threadIdx.x; b < n_words; b += 1024) hist[b] = 0u;
    __syncthreads();
    const uint32_t lo = blockIdx.x * per_run, hi = min(n, lo + per_run);
    for (uint32_t i = lo + threadIdx.x; i < hi; i += 1024)
    {
        const uint32_t key = sample_key(G, pos, i, shift, n_bins);
        keys[i] = key;
        const uint32_t sh = (key & 1u) * 16u;
        rank[i] = (atomicAdd(&hist[key >> 1], 1u << sh) >> sh) & 0xffffu;
    }
    __syncthreads();
    uint32_t nz = 0;  // bins of this run that hold a point: few of them = the run's points came cage by cage already (k_sample_place)
    for (uint32_t b = threadIdx.x; b < n_bins; b += 1024)
    {
        const uint32_t c = (hist[b >> 1] >> ((b & 1u) * 16u)) & 0xffffu;
        counts[static_cast<size_t>(blockIdx.x) * n_bins + b] = c;
        nz += c != 0u ? 1u : 0u;
    }
    __syncthreads();
    if (threadIdx.x == 0) hist[0] = 0u;
    __syncthreads();
    for (int m = 32; m >= 1; m >>= 1) nz += __shfl_xor(nz, m);  // (a wave's sum first: sixteen LDS atomics per run, not a thousand on one word)
    if ((threadIdx.x & 63u) == 0u && nz) atomicAdd(&hist[0], nz);
    __syncthreads();
    if (threadIdx.x == 0) run_nz[blockIdx.x] = hist[0];
}
// counts[r][bin] -> the exclusive sum over the runs before r, per bin (in place); totals[bin] = the bin's points.
// A workgroup = 16 bins x 16 lanes per bin, each lane scanning kSampleRuns / 16 consecutive runs (a chain of 16 instead of 256).
// the batch's points per occupied (run, bin): at 32 or more it came cage by cage already (k_sample_place)
DDGI_D bool sample_batch_in_order(uint32_t occupied, uint32_t n) { return DDGI_SAMPLE_COHERENCE && static_cast<unsigned long long>(occupied) * 32ull <= static_cast<unsigned long long>(n); }

__global__ __launch_bounds__(256) void k_sample_scan_runs(uint32_t n_runs, uint32_t n_bins, uint32_t* __restrict__ counts, uint32_t* __restrict__ totals,
                                                          const uint32_t* __restrict__ run_nz, uint32_t n)
{
    constexpr uint32_t kPer = kSampleRuns / 16;
    __shared__ uint32_t part[16][17];
    static_assert(kSampleRuns <= 256, "one thread per run adds up the runs' occupied bins");
    {
        // (a batch in cage order needs no offsets: k_sample_place comes to the same verdict from the same numbers and writes no permutation)
        __shared__ uint32_t occ[4];
        uint32_t v = threadIdx.x < n_runs ? run_nz[threadIdx.x] : 0u;
        for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
        if ((threadIdx.x & 63u) == 0u) occ[threadIdx.x >> 6] = v;
        __syncthreads();
        if (sample_batch_in_order(occ[0] + occ[1] + occ[2] + occ[3], n)) return;
    }
    const uint32_t bl = threadIdx.x & 15u, chunk = threadIdx.x >> 4, b = blockIdx.x * 16u + bl;
    uint32_t c[kPer];
    uint32_t acc = 0;
#pragma unroll
    for (uint32_t k = 0; k < kPer; ++k)
    {
        const uint32_t r = chunk * kPer + k;
        c[k] = (b < n_bins && r < n_runs) ? counts[static_cast<size_t>(r) * n_bins + b] : 0u;
    }
#pragma unroll
    for (uint32_t k = 0; k < kPer; ++k)
    {
        const uint32_t v = c[k];
        c[k] = acc;
        acc += v;
    }
    part[chunk][bl] = acc;
    __syncthreads();
    uint32_t before = 0;
    for (uint32_t q = 0; q < chunk; ++q) before += part[q][bl];
#pragma unroll
    for (uint32_t k = 0; k < kPer; ++k)
    {
        const uint32_t r = chunk * kPer + k;
        if (b < n_bins && r < n_runs) counts[static_cast<size_t>(r) * n_bins + b] = c[k] + before;
    }
    if (chunk == 15u && b < n_bins) totals[b] = before + acc;
}
// perm[base[key] + before[run][key] + rank[i]] = i, run by run (the workgroups of k_sample_count again).  base[bin] = the exclusive
// sum of totals[] over the bins before it: every workgroup scans the totals itself (at most kSampleBins words, in LDS) and keeps
// its run's row of `before` beside them — the bins' bases need no kernel of their own, and a point's two table look-ups are LDS reads.
__global__ __launch_bounds__(1024) void k_sample_place(uint32_t n, uint32_t per_run, uint32_t n_bins, const uint32_t* __restrict__ keys, const uint32_t* __restrict__ rank,
                                                       const uint32_t* __restrict__ before, const uint32_t* __restrict__ totals, uint32_t* __restrict__ perm,
                                                       const uint32_t* __restrict__ run_nz, uint32_t n_runs, uint32_t* __restrict__ perm_off)
{
    __shared__ uint32_t start[kSampleBins];  // base[bin] + before[run][bin]
    __shared__ uint32_t scan[1024];
    // A BATCH THAT CAME IN CAGE ORDER — a frame's pixels, a G-buffer — gains nothing from the permutation: its runs each touch a few bins.
    // Every workgroup adds up the runs' occupied bins (k_sample_count); at 32 points or more per occupied (run, bin) the batch goes as it
    // came: no permutation is written, the sample kernels are told so (*perm_off), and the scatter's 25 us of a 1.44 M-point batch stay unspent.
    {
        scan[threadIdx.x] = threadIdx.x < n_runs ? run_nz[threadIdx.x] : 0u;  // (n_runs <= kSampleRuns <= 1024)
        __syncthreads();
        for (uint32_t off = 512; off >= 1; off >>= 1)
        {
            if (threadIdx.x < off) scan[threadIdx.x] += scan[threadIdx.x + off];
            __syncthreads();
        }
        const bool in_order = sample_batch_in_order(scan[0], n);
        __syncthreads();  // (scan[] is written again below)
        if (blockIdx.x == 0 && threadIdx.x == 0) *perm_off = in_order ? 1u : 0u;
        if (in_order) return;
    }
    constexpr uint32_t kPer = kSampleBins / 1024;
    uint32_t v[kPer], mine = 0;
#pragma unroll
    for (uint32_t k = 0; k < kPer; ++k)
    {
        const uint32_t b = threadIdx.x * kPer + k;
        v[k] = b < n_bins ? totals[b] : 0u;
        mine += v[k];
    }
    scan[threadIdx.x] = mine;
    __syncthreads();
    for (uint32_t off = 1; off < 1024; off <<= 1)
    {
        const uint32_t add = threadIdx.x >= off ? scan[threadIdx.x - off] : 0u;
        __syncthreads();
        scan[threadIdx.x] += add;
        __syncthreads();
    }
    uint32_t base = scan[threadIdx.x] - mine;
    const uint32_t* __restrict__ row = before + static_cast<size_t>(blockIdx.x) * n_bins;
#pragma unroll
    for (uint32_t k = 0; k < kPer; ++k)
    {
        const uint32_t b = threadIdx.x * kPer + k;
        if (b < n_bins) start[b] = base + row[b];
        base += v[k];
    }
    __syncthreads();
    const uint32_t lo = blockIdx.x * per_run, hi = min(n, lo + per_run);
    for (uint32_t i = lo + threadIdx.x; i < hi; i += 1024) perm[start[keys[i]] + rank[i]] = i;
}

// Bins are as fine as kSampleBins allows: the sample kernels' waves then hold points of one cage (C3: bins of 4 cages), whose 8
// probes' tiles they share in the vector cache (bins of 8 cages: k_probe_sample_ddgi 147 -> 207 us per 1.44 M points)
static void sample_bins(uint32_t n_probes, uint32_t& shift, uint32_t& n_bins)
{
    shift = 0;
    while ((n_probes >> shift) + 2u > kSampleBins) ++shift;  // (+ the bin of the points outside the field)
    n_bins = ((n_probes + (1u << shift) - 1u) >> shift) + 1u;
}
// scratch: keys[n] | rank[n] | perm[n] | counts[kSampleRuns x n_bins] | totals[n_bins]   (uint32 each)
size_t sample_group_scratch_words(uint32_t n, uint32_t n_probes)
{
    uint32_t shift, n_bins;
    sample_bins(n_probes, shift, n_bins);
    return 3 * static_cast<size_t>(n) + static_cast<size_t>(kSampleRuns + 1) * n_bins + kSampleRuns + 4;  // (+ run_nz[kSampleRuns] | perm_off)
}

hipError_t launch_sample_grouping(const GridK& grid, const float* pos, uint32_t n, uint32_t* scratch, const uint32_t** perm_out, const uint32_t** perm_off_out, hipStream_t stream)
{
    uint32_t shift, n_bins;
    const uint32_t n_probes = static_cast<uint32_t>(grid.cx) * grid.cy * grid.cz;
    sample_bins(n_probes, shift, n_bins);
    // (a run's 16-bit counters: fewer than 65 536 points per run — more runs than kSampleRuns for batches beyond 16 M points)
    const uint32_t n_runs = std::max<uint32_t>(std::min<uint32_t>(kSampleRuns, (n + 1023u) / 1024u), (n + 65534u) / 65535u);
    if (n_runs > kSampleRuns) return hipErrorInvalidValue;  // (ddgi_engine.cpp splits batches of more than 2^24 points)
    const uint32_t per_run = (n + n_runs - 1u) / n_runs;
    uint32_t *keys = scratch, *rank = keys + n, *perm = rank + n, *counts = perm + n, *totals = counts + static_cast<size_t>(kSampleRuns) * n_bins;
    const size_t lds = static_cast<size_t>((n_bins + 1u) / 2u) * sizeof(uint32_t);
    uint32_t *run_nz = totals + n_bins, *perm_off = run_nz + kSampleRuns;
    hipLaunchKernelGGL(k_sample_count, dim3(n_runs), dim3(1024), lds, stream, grid, pos, n, per_run, shift, n_probes, n_bins, keys, rank, counts, run_nz);
    hipLaunchKernelGGL(k_sample_scan_runs, dim3((n_bins + 15u) / 16u), dim3(256), 0, stream, n_runs, n_bins, counts, totals, run_nz, n);
    hipLaunchKernelGGL(k_sample_place, dim3(n_runs), dim3(1024), 0, stream, n, per_run, n_bins, keys, rank, counts, totals, perm, run_nz, n_runs, perm_off);
    *perm_out = perm;
    *perm_off_out = perm_off;
    return hipGetLastError();
}

// ---- launchers (called from ddgi_engine.cpp) -----------------------------------------------------

hipError_t launch_probe_trace_ref(const TraceArgs& args, int grid_blocks, hipStream_t stream)
{
    const size_t lds = static_cast<size_t>(args.scene.nwords) * sizeof(uint32_t);
    hipLaunchKernelGGL(k_probe_trace_ref, dim3(grid_blocks), dim3(kTraceBlock), lds, stream, args);
    return hipGetLastError();
}

hipError_t launch_probe_sample_ref(const SampleArgs& args, hipStream_t stream)
{
    const unsigned blocks = (args.n + 255u) / 256u;
    if (blocks == 0) return hipSuccess;
    hipLaunchKernelGGL(k_probe_sample_ref, dim3(blocks), dim3(256), 0, stream, args);
    return hipGetLastError();
}

hipError_t trace_kernel_occupancy(int* blocks_per_cu, size_t lds_bytes)
{
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, k_probe_trace_ref, kTraceBlock, lds_bytes);
}

// ------------------------------------------------------------------------------------------------
// k_carry_tiles — reconfiguration with carry-over: the tile of every new probe
// that stands exactly where an old probe stood is copied from the old textures; map[new slot] = old
// slot or -1.  One 64-lane wave per probe tile, 4-byte words.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_carry_tiles(uint32_t* __restrict__ dst, const uint32_t* __restrict__ src, const int32_t* __restrict__ map, uint32_t n_probes,
                                                     uint32_t words_per_tile)
{
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63u;
    const uint32_t n_waves = (gridDim.x * blockDim.x) >> 6;
    for (uint32_t p = wave; p < n_probes; p += n_waves)
    {
        const int32_t from = map[p];
        if (from < 0) continue;
        const uint32_t* s = src + static_cast<size_t>(from) * words_per_tile;
        uint32_t* d = dst + static_cast<size_t>(p) * words_per_tile;
        for (uint32_t w = lane; w < words_per_tile; w += 64u) d[w] = s[w];
    }
}

hipError_t launch_carry_tiles(void* dst, const void* src, const int32_t* map, uint32_t n_probes, uint32_t words_per_tile, hipStream_t stream)
{
    if (n_probes == 0) return hipSuccess;
    const unsigned blocks = std::min<unsigned>((n_probes + 3u) / 4u, 4096u);
    hipLaunchKernelGGL(k_carry_tiles, dim3(blocks), dim3(256), 0, stream, static_cast<uint32_t*>(dst), static_cast<const uint32_t*>(src), map, n_probes, words_per_tile);
    return hipGetLastError();
}

}  // namespace ddgi
