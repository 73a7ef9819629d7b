// ddgi_sampler.h — the 8-probe-cage sample for ONE shading point, REF and DDGI mode (device).
// Used by the batched sample kernels (ddgi_kernels.hip, ddgi_blend_sample.hip) and by the
// primary-visibility integrators (ddgi_render.hip).
#pragma once

#include "ddgi_device.h"
#include "ddgi_oct.h"

#include <type_traits>

namespace ddgi {

// rgba8 UNORM -> float: c / 255 (IEEE division).  The 256 possible quotients are tabulated in LDS
// once per workgroup (s_unorm[c] == float(c) / 255.0f exactly), replacing ~10 VALU instructions per
// channel by one ds_read.
DDGI_D f3 load_rgb(const uint32_t* tex, size_t i, const float* s_unorm)
{
    const uint32_t v = tex[i];
    return f3{s_unorm[v & 255u], s_unorm[(v >> 8) & 255u], s_unorm[(v >> 16) & 255u]};
}

// Where a direction lands in a probe's tile: sample_probe's inversion of generate_samples (intersection.glsl:1180-1211).  The
// same for all 8 probes of a cage — they are all asked for the shading normal.
DDGI_D void sample_texel_ref(const GridK& G, f3 dir, int& rx, int& ry)
{
    const int s = G.sx, sh = G.sy;
    const f3 id = normalize3(dir);
    rx = gl_int(((-1.0f * (id.z - 1.0f)) / 2.0f) * static_cast<float>(s));
    if (rx == s) rx = 0;
    const float sqrt_z = sqrtf(1.0f - (id.z * id.z));
    const float kPi = 3.1415926535897932384626433832795f;
    ry = gl_int((pm::acosf_pinned(id.x / sqrt_z) / (2.0f * kPi)) * static_cast<float>(sh));
}

// The rest of sample_probe (:1213-1239) for tile texel (rx, ry): (albedo[centre] + the 5x5 box around it clipped to the tile,
// dx outer, dy inner) / count.  fetch(off) = texel `off` of the tile as floats c / 255.
template <class Fetch>
DDGI_D f3 sample_box_ref(int s, int sh, int rx, int ry, const Fetch& fetch)
{
    f3 result = fetch(ry * s + rx);  // (`albedo` and `which` are the same image, intersection.glsl:1226)
    int count = 0;
    for (int dx = -2; dx <= 2; ++dx)
    {
        const int x = rx + dx;
        if (x < 0 || x >= s) continue;
        for (int dy = -2; dy <= 2; ++dy)
        {
            const int y = ry + dy;
            if (y < 0 || y >= sh) continue;
            count += 1;
            result = result + fetch(y * s + x);
        }
    }
    return div3(result, static_cast<float>(count));
}

// sample_probe (intersection.glsl:1176-1240) against the slab-major texel buffer
DDGI_D f3 sample_probe_ref(const GridK& G, const uint32_t* albedo, const uint32_t* tex, int probe, f3 dir, const float* s_unorm)
{
    const int cxz = G.cx * G.cz;
    // get_text_coord_from_probe_number (:1152-1174): out of range -> magenta
    if (probe >= cxz * G.cy || probe < 0) return mk3(1, 0, 1);
    int rx, ry;
    sample_texel_ref(G, dir, rx, ry);
    const size_t base = static_cast<size_t>(slab_slot(G, probe)) * G.n;
    return sample_box_ref(G.sx, G.sy, rx, ry, [&](int off) { return load_rgb(tex, base + static_cast<size_t>(off), s_unorm); });
}

// get_diffuse_gi (intersection.glsl:1306-1409), REF mode: rgb for one shading point; cage[8] receives
// the probe_index_1d of the 8 cage corners, or -1 everywhere when the shader returns magenta.
// box (optional): sample_probe's value for EVERY texel of every tile, tabulated by k_sample_box_filter — [ry][rx][slab slot] float4;
// then a corner costs one 16-byte load instead of 26 gathers and 78 conversions.
// kBatch: ask for the 8 table entries up front (k_probe_sample_ref: the batch is bound by the entries' latency); the pixel kernel
// (k_render_primary), whose registers are spoken for by the camera ray's march, takes them corner by corner.
// The table's layout (ddgi_types.h: DDGI_BOX_LAYOUT): 0 = tile-major [slab slot][ry][rx] (a tile's entries are one contiguous run), 1 = texel-major
// [ry][rx][slab slot] (a cage's two x-neighbours share 32 contiguous bytes), 2 = texel-major in 2x2x2 bricks of probes (a brick's 8 entries
// are one 128-byte line: a cage's entries come from 3.4 lines on average instead of 4.5)
DDGI_D size_t box_index(uint32_t texel, uint32_t slot, uint32_t texels_per_tile, uint32_t n_slots)
{
    if (DDGI_BOX_LAYOUT == 3)  // brick-major: [brick][texel][the brick's 8 probes] — a brick's lines of all texels are one contiguous run (the build streams)
        return (static_cast<size_t>(slot >> 3) * texels_per_tile + texel) * 8u + (slot & 7u);
    return DDGI_BOX_LAYOUT != 0 ? static_cast<size_t>(texel) * n_slots + slot : static_cast<size_t>(slot) * texels_per_tile + texel;
}
// the table slot of reference probe index p = y*cx*cz + z*cx + x (decoded: two integer divisions)
DDGI_D uint32_t box_slot_of_probe(const GridK& G, int p)
{
    const int cxz = G.cx * G.cz;
    const int y = p / cxz;
    const int rem = p - y * cxz;
    const int z = rem / G.cx;
    return box_slot_xyz(G.cx, G.cy, rem - z * G.cx, y, z);
}
// ... of the index get_diffuse_gi makes of a cage corner's grid coordinates: inside the grid no index has wrapped (ddgi_device.h: slab_slot_of_corner)
DDGI_D uint32_t box_slot_of_corner(const GridK& G, int sx, int sy, int sz, int idx)
{
    const bool inside = static_cast<unsigned>(sx) < static_cast<unsigned>(G.cx) && static_cast<unsigned>(sy) < static_cast<unsigned>(G.cy) && static_cast<unsigned>(sz) < static_cast<unsigned>(G.cz);
    if (__builtin_expect(!inside, 0)) return box_slot_of_probe(G, idx);
    return box_slot_xyz(G.cx, G.cy, sx, sy, sz);
}

template <bool kBatch = false>
DDGI_D f3 diffuse_gi_ref(const GridK& G, const uint32_t* albedo, f3 pos, f3 nrm_raw, const float* s_unorm, int* cage, const float4* box = nullptr)
{
    const f3 N = normalize3(nrm_raw);
    const f3 origin{G.origin[0], G.origin[1], G.origin[2]};
    const float side = static_cast<float>(G.side);
    for (int k = 0; k < 8; ++k) cage[k] = -1;
    f3 out = mk3(1, 0, 1);
    bool ok = true;

    const f3 rel = div3(pos - origin, side);
    const int bx = gl_int(floorf(rel.x)), by = gl_int(floorf(rel.y)), bz = gl_int(floorf(rel.z));
    // Q6: every axis is bounds-checked against probe_count.x
    const int lo = gl_int(-floorf(static_cast<float>(G.cx) / 2.0f));
    const int hi = gl_int(floorf(static_cast<float>(G.cx) / 2.0f) - 1.0f);
    if (bx < lo || bx > hi || by < lo || by > hi || bz < lo || bz > hi) ok = false;

    if (ok)
    {
        const f3 base_world = f3{static_cast<float>(bx * G.side), static_cast<float>(by * G.side), static_cast<float>(bz * G.side)} + origin;
        const f3 a = div3(pos - base_world, side);
        const f3 alpha{gl_clamp(a.x, 0.0f, 1.0f), gl_clamp(a.y, 0.0f, 1.0f), gl_clamp(a.z, 0.0f, 1.0f)};
        f3 irradiance = mk3(0, 0, 0);
        float sum_weight = 0.0f;
        const int n_probes = G.cx * G.cy * G.cz;
        const uint32_t n_box_slots = box_slots(G.cx, G.cy, G.cz);
        int box_off = 0;
        if (box)
        {
            int rx, ry;
            sample_texel_ref(G, N, rx, ry);
            box_off = ry * G.sx + rx;
        }
        // one corner of the cage (intersection.glsl:1331-1404): its weight, and its sample_probe value into the sum
        auto corner = [&](int k, int idx, f3 smp) {
            const int ox = (k >> 2) & 1, oy = (k >> 1) & 1, oz = k & 1;  // Q7 corner order
            cage[k] = idx;
            const f3 tri{ox ? alpha.x : 1.0f - alpha.x, oy ? alpha.y : 1.0f - alpha.y, oz ? alpha.z : 1.0f - alpha.z};
            const f3 probe_pos = base_world + f3{static_cast<float>(ox * G.side), static_cast<float>(oy * G.side), static_cast<float>(oz * G.side)};
            const f3 dir = normalize3(probe_pos - pos);
            const float tmp = gl_max(0.0001f, (dot3(dir, N) + 1.0f) * 0.5f);
            float weight = tmp * tmp + 0.2f;
            weight = gl_max(0.000001f, weight);
            const float crush = 0.2f;
            if (weight < crush) weight *= weight * weight * (1.f / (crush * crush));  // unreachable (Q11)
            weight *= tri.x * tri.y * tri.z;
            irradiance = irradiance + smp * weight;
            sum_weight += weight;
        };
        auto corner_index = [&](int k) {
            const int sx = bx + ((k >> 2) & 1) + G.cx / 2, sy = by + ((k >> 1) & 1) + G.cy / 2, sz = bz + (k & 1) + G.cz / 2;  // Q4
            return sy * G.cx * G.cz + sz * G.cx + sx;
        };
        auto corner_slot = [&](int k, int idx) {  // the table slot of a corner in range
            return box_slot_of_corner(G, bx + ((k >> 2) & 1) + G.cx / 2, by + ((k >> 1) & 1) + G.cy / 2, bz + (k & 1) + G.cz / 2, idx);
        };
        if constexpr (kBatch)
        {
            // (with the table only.)  ALL EIGHT entries are asked for before the first is used: taken corner by corner each load
            // stands behind the previous corner's range check (a possible `break`), i.e. eight trips to memory one after the other —
            // and the batch is bound by exactly that latency, whatever the table's layout (7.5 -> 10 G points/s on scattered points).
            // A cage with a corner out of range reads nothing and returns magenta.
            int idx[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) idx[k] = corner_index(k), ok = ok && idx[k] >= 0 && idx[k] < n_probes;
            if (ok)
            {
                // the corners' table slots: with the whole cage inside the grid (ONE test in front of the loads: bx, by, bz and their
                // successors in range) no index has wrapped and a slot comes from the coordinates — per axis one term for either
                // corner, a slot is their sum; else every index is decoded
                const int sx0 = bx + G.cx / 2, sy0 = by + G.cy / 2, sz0 = bz + G.cz / 2;
                const bool cage_inside = sx0 >= 0 && sx0 + 1 < G.cx && sy0 >= 0 && sy0 + 1 < G.cy && sz0 >= 0 && sz0 + 1 < G.cz;
                uint32_t slot[8];
                if (cage_inside)
                {
                    uint32_t tx[2], ty[2], tz[2];
#pragma unroll
                    for (int o = 0; o < 2; ++o)
                        tx[o] = box_slot_xyz(G.cx, G.cy, sx0 + o, 0, 0), ty[o] = box_slot_xyz(G.cx, G.cy, 0, sy0 + o, 0), tz[o] = box_slot_xyz(G.cx, G.cy, 0, 0, sz0 + o);
#pragma unroll
                    for (int k = 0; k < 8; ++k) slot[k] = tz[k & 1] + ty[(k >> 1) & 1] + tx[(k >> 2) & 1];
                }
                else
                {
#pragma unroll
                    for (int k = 0; k < 8; ++k) slot[k] = box_slot_of_probe(G, idx[k]);
                }
                float4 tab[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) tab[k] = box[box_index(static_cast<uint32_t>(box_off), slot[k], static_cast<uint32_t>(G.n), n_box_slots)];
#pragma unroll
                for (int k = 0; k < 8; ++k) corner(k, idx[k], f3{tab[k].x, tab[k].y, tab[k].z});
            }
        }
        else
        {
            for (int k = 0; k < 8 && ok; ++k)
            {
                const int idx = corner_index(k);
                if (idx < 0 || idx >= n_probes)
                {
                    ok = false;
                    break;
                }
                f3 smp;
                if (box)
                {
                    const float4 v = box[box_index(static_cast<uint32_t>(box_off), corner_slot(k, idx), static_cast<uint32_t>(G.n), n_box_slots)];  // (idx is a valid probe here)
                    smp = f3{v.x, v.y, v.z};
                }
                else
                    smp = sample_probe_ref(G, albedo, albedo, idx, N, s_unorm);
                corner(k, idx, smp);
            }
        }
        if (ok) out = div3(irradiance, sum_weight);
    }
    if (!ok)
    {
        out = mk3(1, 0, 1);
        for (int k = 0; k < 8; ++k) cage[k] = -1;
    }
    return out;
}

// Where a direction lands in a bordered kSide x kSide octahedral tile: the four texels of the bilinear fetch (offsets in texels)
// and the two weights.  The same for every tile that is asked for the same direction — all 8 irradiance tiles of a cage are
// asked for the shading normal — so a caller works it out once.
struct TileCoords
{
    int o00, o01, o10, o11;  // (y0, x0), (y0, x1), (y1, x0), (y1, x1)
    float tx, ty;
};
template <int kSide>
DDGI_D TileCoords tile_coords(f3 dir)
{
    const f2 uv = oct_encode(normalize3(dir));
    const float inner = static_cast<float>(kSide - 2);
    const float fx = (uv.x * 0.5f + 0.5f) * inner + 0.5f;  // texel-centre coordinate inside the bordered tile
    const float fy = (uv.y * 0.5f + 0.5f) * inner + 0.5f;
    const float bx = floorf(fx), by = floorf(fy);
    int x0 = gl_int(bx), y0 = gl_int(by);
    int x1 = x0 + 1, y1 = y0 + 1;
    x0 = max(x0, 0), y0 = max(y0, 0);
    x1 = min(x1, kSide - 1), y1 = min(y1, kSide - 1);
    return TileCoords{y0 * kSide + x0, y0 * kSide + x1, y1 * kSide + x0, y1 * kSide + x1, fx - bx, fy - by};
}
// kVec: a texel's kCh floats come as ONE load (the tile's base is aligned to kCh floats: the engine's own buffers are; textures
// bound by the host are checked, launch_probe_sample_ddgi) — 4 loads per fetch instead of 4 kCh
template <int kCh, bool kVec = false>
DDGI_D void tile_gather(const float* tile, const TileCoords& c, float* out)
{
    float a[kCh], b[kCh], cc[kCh], d[kCh];
    if constexpr (kVec && kCh == 4)
    {
        const float4 va = *reinterpret_cast<const float4*>(tile + c.o00 * 4), vb = *reinterpret_cast<const float4*>(tile + c.o01 * 4);
        const float4 vc = *reinterpret_cast<const float4*>(tile + c.o10 * 4), vd = *reinterpret_cast<const float4*>(tile + c.o11 * 4);
        a[0] = va.x, a[1] = va.y, a[2] = va.z, a[3] = va.w, b[0] = vb.x, b[1] = vb.y, b[2] = vb.z, b[3] = vb.w;
        cc[0] = vc.x, cc[1] = vc.y, cc[2] = vc.z, cc[3] = vc.w, d[0] = vd.x, d[1] = vd.y, d[2] = vd.z, d[3] = vd.w;
    }
    else if constexpr (kVec && kCh == 2)
    {
        const float2 va = *reinterpret_cast<const float2*>(tile + c.o00 * 2), vb = *reinterpret_cast<const float2*>(tile + c.o01 * 2);
        const float2 vc = *reinterpret_cast<const float2*>(tile + c.o10 * 2), vd = *reinterpret_cast<const float2*>(tile + c.o11 * 2);
        a[0] = va.x, a[1] = va.y, b[0] = vb.x, b[1] = vb.y, cc[0] = vc.x, cc[1] = vc.y, d[0] = vd.x, d[1] = vd.y;
    }
    else
    {
        for (int ch = 0; ch < kCh; ++ch)
            a[ch] = tile[c.o00 * kCh + ch], b[ch] = tile[c.o01 * kCh + ch], cc[ch] = tile[c.o10 * kCh + ch], d[ch] = tile[c.o11 * kCh + ch];
    }
    for (int ch = 0; ch < kCh; ++ch) out[ch] = gl_mix(gl_mix(a[ch], b[ch], c.tx), gl_mix(cc[ch], d[ch], c.tx), c.ty);
}
template <int kSide, int kCh>
DDGI_D void tile_fetch(const float* tile, f3 dir, float* out)
{
    tile_gather<kCh>(tile, tile_coords<kSide>(dir), out);
}

// get_diffuse_gi with the dormant Chebyshev lines (1363-1383) enabled and octahedral bilinear tile
// fetches, for one shading point (DDGI mode).
// Where the sampler's tiles come from: the slab-major tile buffers in HBM.  (A policy, so that another source can stand in: a kernel
// that staged a bin of cages' 36 tiles in LDS first was built on it in round 5 and measured slower — docs/LAB_NOTES.md.)
template <bool kVec>
struct TilesGlobal
{
    const float* irradiance;
    const float* depth;
    DDGI_D f2 gather_depth(uint32_t slot, const TileCoords& c) const
    {
        float o[2];
        tile_gather<2, kVec>(depth + static_cast<size_t>(slot) * (kDepTile * kDepTile * 2), c, o);
        return f2{o[0], o[1]};
    }
    DDGI_D f3 gather_irradiance(uint32_t slot, const TileCoords& c) const  // (rgb: the sampler does not read alpha)
    {
        float o[4];
        tile_gather<4, kVec>(irradiance + static_cast<size_t>(slot) * (kIrrTile * kIrrTile * 4), c, o);
        return f3{o[0], o[1], o[2]};
    }
    // the same in two halves — the four texels asked for, and mixed later (tile_gather's order of operations) — for the sampler's pipelined
    // corner loop (diffuse_gi_ddgi_from): a corner's texels travel while the corner before it is weighed
    static constexpr bool kSplit = kVec;
    struct DepthTexels
    {
        float2 a, b, c, d;
    };
    struct IrrTexels
    {
        float4 a, b, c, d;
    };
    DDGI_D DepthTexels load_depth(uint32_t slot, const TileCoords& c) const
    {
        const float* tile = depth + static_cast<size_t>(slot) * (kDepTile * kDepTile * 2);
        return DepthTexels{*reinterpret_cast<const float2*>(tile + c.o00 * 2), *reinterpret_cast<const float2*>(tile + c.o01 * 2), *reinterpret_cast<const float2*>(tile + c.o10 * 2),
                           *reinterpret_cast<const float2*>(tile + c.o11 * 2)};
    }
    DDGI_D IrrTexels load_irradiance(uint32_t slot, const TileCoords& c) const
    {
        const float* tile = irradiance + static_cast<size_t>(slot) * (kIrrTile * kIrrTile * 4);
        return IrrTexels{*reinterpret_cast<const float4*>(tile + c.o00 * 4), *reinterpret_cast<const float4*>(tile + c.o01 * 4), *reinterpret_cast<const float4*>(tile + c.o10 * 4),
                         *reinterpret_cast<const float4*>(tile + c.o11 * 4)};
    }
    static DDGI_D f2 mix_depth(const DepthTexels& t, const TileCoords& c)
    {
        return f2{gl_mix(gl_mix(t.a.x, t.b.x, c.tx), gl_mix(t.c.x, t.d.x, c.tx), c.ty), gl_mix(gl_mix(t.a.y, t.b.y, c.tx), gl_mix(t.c.y, t.d.y, c.tx), c.ty)};
    }
    static DDGI_D f3 mix_irradiance(const IrrTexels& t, const TileCoords& c)
    {
        return f3{gl_mix(gl_mix(t.a.x, t.b.x, c.tx), gl_mix(t.c.x, t.d.x, c.tx), c.ty), gl_mix(gl_mix(t.a.y, t.b.y, c.tx), gl_mix(t.c.y, t.d.y, c.tx), c.ty),
                  gl_mix(gl_mix(t.a.z, t.b.z, c.tx), gl_mix(t.c.z, t.d.z, c.tx), c.ty)};
    }
};

// The same for tile buffers below 4 GiB each (every grid up to 2 M probes; the engine's own, aligned buffers): a texel's address is the buffer's base —
// wave-uniform, scalar registers — plus a 32-bit byte offset per lane (global_load ... v, s[base:base+1]) instead of a 64-bit address per lane
// (v_lshl_add_u64 / v_mad_u64_u32: 16 of a corner's 320 instructions).
struct TilesGlobal32
{
    const float* irradiance;
    const float* depth;
    template <class V>
    static DDGI_D V at(const float* base, uint32_t byte_offset) { return *reinterpret_cast<const V*>(reinterpret_cast<const char*>(base) + byte_offset); }
    DDGI_D f2 gather_depth(uint32_t slot, const TileCoords& c) const
    {
        const uint32_t t = slot * static_cast<uint32_t>(kDepTile * kDepTile * 2 * 4);
        const float2 a = at<float2>(depth, t + static_cast<uint32_t>(c.o00) * 8u), b = at<float2>(depth, t + static_cast<uint32_t>(c.o01) * 8u);
        const float2 cc = at<float2>(depth, t + static_cast<uint32_t>(c.o10) * 8u), d = at<float2>(depth, t + static_cast<uint32_t>(c.o11) * 8u);
        return f2{gl_mix(gl_mix(a.x, b.x, c.tx), gl_mix(cc.x, d.x, c.tx), c.ty), gl_mix(gl_mix(a.y, b.y, c.tx), gl_mix(cc.y, d.y, c.tx), c.ty)};
    }
    DDGI_D f3 gather_irradiance(uint32_t slot, const TileCoords& c) const
    {
        const uint32_t t = slot * static_cast<uint32_t>(kIrrTile * kIrrTile * 4 * 4);
        const float4 a = at<float4>(irradiance, t + static_cast<uint32_t>(c.o00) * 16u), b = at<float4>(irradiance, t + static_cast<uint32_t>(c.o01) * 16u);
        const float4 cc = at<float4>(irradiance, t + static_cast<uint32_t>(c.o10) * 16u), d = at<float4>(irradiance, t + static_cast<uint32_t>(c.o11) * 16u);
        return f3{gl_mix(gl_mix(a.x, b.x, c.tx), gl_mix(cc.x, d.x, c.tx), c.ty), gl_mix(gl_mix(a.y, b.y, c.tx), gl_mix(cc.y, d.y, c.tx), c.ty),
                  gl_mix(gl_mix(a.z, b.z, c.tx), gl_mix(cc.z, d.z, c.tx), c.ty)};
    }
};

// does a tile source offer its fetches in two halves (load_*, mix_*)?
template <class T, class = void>
struct tiles_split : std::false_type
{
};
template <class T>
struct tiles_split<T, std::enable_if_t<T::kSplit>> : std::true_type
{
};

template <class Tiles>
DDGI_D f3 diffuse_gi_ddgi_from(const GridK& G, const Tiles& tiles, f3 pos, f3 nrm_raw, int* cage);

template <bool kVec = false>
DDGI_D f3 diffuse_gi_ddgi(const GridK& G, const float* irradiance, const float* depth, f3 pos, f3 nrm_raw, int* cage)
{
    return diffuse_gi_ddgi_from(G, TilesGlobal<kVec>{irradiance, depth}, pos, nrm_raw, cage);
}

template <class Tiles>
DDGI_D f3 diffuse_gi_ddgi_from(const GridK& G, const Tiles& tiles, f3 pos, f3 nrm_raw, int* cage)
{
    const f3 N = normalize3(nrm_raw);
    const f3 origin{G.origin[0], G.origin[1], G.origin[2]};
    const float side = static_cast<float>(G.side);
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
        const TileCoords irr_at = tile_coords<kIrrTile>(N);  // (all 8 corners are asked for the shading normal)
        // (the corners' range check comes first, for all eight: with a possible `break` inside the loop below every corner's
        // tile fetches would have to wait for the corner before — eight round trips to memory in a row)
        // (and the cage indices are set here, with constant subscripts: written inside the loop below — which is not unrolled — `cage[k]`
        // is a chain of ten compares and selects per corner; a cage with a corner out of range is reset to -1 at the end)
#pragma unroll
        for (int k = 0; k < 8; ++k)
        {
            const int sx = bx + ((k >> 2) & 1) + G.cx / 2, sy = by + ((k >> 1) & 1) + G.cy / 2, sz = bz + (k & 1) + G.cz / 2;
            const int idx = sy * G.cx * G.cz + sz * G.cx + sx;
            ok = ok && idx >= 0 && idx < n_probes;
            cage[k] = idx;
        }
#ifndef DDGI_SAMPLE_PIPELINED
#define DDGI_SAMPLE_PIPELINED 1
#endif
        if constexpr (DDGI_SAMPLE_PIPELINED && tiles_split<Tiles>::value)
        {
            // THE PIPELINED CORNER LOOP.  A corner is a round trip to memory — its tile slot and direction, eight texels, then the weight that
            // needs them — and corner by corner a lane makes eight of them in a row (the counters: 60 % of the kernel's wave cycles at a wait
            // for memory).  Here a corner's texels are asked for BEFORE the corner in front of it is weighed: two corners' requests in flight
            // per lane, the arithmetic of one under the latency of the other.  Same operations in the same order per corner.
            if (ok)
            {
                struct Corner
                {
                    f3 tri;
                    float weight0, dist;
                    TileCoords dep_at;
                    typename Tiles::DepthTexels dt;
                    typename Tiles::IrrTexels it;
                };
                auto ask = [&](int k, Corner& c) {
                    const int ox = (k >> 2) & 1, oy = (k >> 1) & 1, oz = k & 1;                            // Q7
                    const int sx = bx + ox + G.cx / 2, sy = by + oy + G.cy / 2, sz = bz + oz + G.cz / 2;  // Q4
                    const int idx = sy * G.cx * G.cz + sz * G.cx + sx;
                    c.tri = f3{ox ? alpha.x : 1.0f - alpha.x, oy ? alpha.y : 1.0f - alpha.y, oz ? alpha.z : 1.0f - alpha.z};
                    const f3 probe_pos = base_world + f3{static_cast<float>(ox * G.side), static_cast<float>(oy * G.side), static_cast<float>(oz * G.side)};
                    const f3 dir = normalize3(probe_pos - pos);
                    const float tmp = gl_max(0.0001f, (dot3(dir, N) + 1.0f) * 0.5f);
                    c.weight0 = tmp * tmp + 0.2f;
                    const uint32_t slot = static_cast<uint32_t>(slab_slot_of_corner(G, sx, sy, sz, idx));
                    c.dist = length3(pos - probe_pos);
                    c.dep_at = tile_coords<kDepTile>(f3{-dir.x, -dir.y, -dir.z});
                    c.dt = tiles.load_depth(slot, c.dep_at);
                    c.it = tiles.load_irradiance(slot, irr_at);
                };
                auto weigh = [&](const Corner& c) {
                    float weight = c.weight0;
                    const f2 mms = Tiles::mix_depth(c.dt, c.dep_at);  // moment visibility test (intersection.glsl:1363-1383, enabled)
                    const float mean = mms.x;
                    const float variance = fabsf(mean * mean - mms.y);
                    const float tmp = gl_max(c.dist - mean, 0.0f);
                    float cheb = variance / (variance + tmp * tmp);
                    cheb = gl_max(cheb * cheb * cheb, 0.0f);
                    if (!(c.dist <= mean)) weight *= cheb;
                    weight = gl_max(0.000001f, weight);
                    const float crush = 0.2f;
                    if (weight < crush) weight *= weight * weight * (1.f / (crush * crush));
                    weight *= c.tri.x * c.tri.y * c.tri.z;
                    irr = irr + Tiles::mix_irradiance(c.it, irr_at) * weight;
                    sum_w += weight;
                };
                Corner c0, c1;
                ask(0, c0);
#pragma unroll 1  // (two corners per trip: each of the two sets of registers is asked for while the other is weighed)
                for (int k = 0; k < 8; k += 2)
                {
                    ask(k + 1, c1);
                    weigh(c0);
                    if (k + 2 < 8) ask(k + 2, c0);
                    weigh(c1);
                }
            }
        }
        else
#pragma unroll 1  // (eight copies of the corner's body and its two tile fetches do not fit the register file)
        for (int k = 0; k < 8 && ok; ++k)
        {
            const int ox = (k >> 2) & 1, oy = (k >> 1) & 1, oz = k & 1;                            // Q7
            const int sx = bx + ox + G.cx / 2, sy = by + oy + G.cy / 2, sz = bz + oz + G.cz / 2;  // Q4
            const int idx = sy * G.cx * G.cz + sz * G.cx + sx;
            const f3 tri{ox ? alpha.x : 1.0f - alpha.x, oy ? alpha.y : 1.0f - alpha.y, oz ? alpha.z : 1.0f - alpha.z};
            const f3 probe_pos = base_world + f3{static_cast<float>(ox * G.side), static_cast<float>(oy * G.side), static_cast<float>(oz * G.side)};
            const f3 dir = normalize3(probe_pos - pos);
            float tmp = gl_max(0.0001f, (dot3(dir, N) + 1.0f) * 0.5f);
            float weight = tmp * tmp + 0.2f;
            const uint32_t slot = static_cast<uint32_t>(slab_slot_of_corner(G, sx, sy, sz, idx));
            // moment visibility test (intersection.glsl:1363-1383, enabled)
            const float dist = length3(pos - probe_pos);
            const f2 mms = tiles.gather_depth(slot, tile_coords<kDepTile>(f3{-dir.x, -dir.y, -dir.z}));
            const float mean = mms.x;
            const float variance = fabsf(mean * mean - mms.y);
            tmp = gl_max(dist - mean, 0.0f);
            float cheb = variance / (variance + tmp * tmp);
            cheb = gl_max(cheb * cheb * cheb, 0.0f);
            if (!(dist <= mean)) weight *= cheb;
            weight = gl_max(0.000001f, weight);
            const float crush = 0.2f;
            if (weight < crush) weight *= weight * weight * (1.f / (crush * crush));
            weight *= tri.x * tri.y * tri.z;
            irr = irr + tiles.gather_irradiance(slot, irr_at) * weight;
            sum_w += weight;
        }
        if (ok) out = div3(irr, sum_w);
    }
    if (!ok)
    {
        out = mk3(1, 0, 1);
        for (int k = 0; k < 8; ++k) cage[k] = -1;
    }
    return out;
}

}  // namespace ddgi
