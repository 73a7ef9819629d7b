// ddgi_host.cpp — host half of the probe path: scene bake and probe-ray generation.
// (compiled as HIP only so that it can share ddgi_scene.h with the device code; no kernels here)
#include "ddgi_host.h"

#include <array>
#include <cmath>
#include <cstring>
#include <algorithm>
#include <mutex>
#include <thread>
#include <vector>

#include "ddgi_scene.h"

namespace ddgi {

// ---- scene bake -----------------------------------------------------------------------------------
//
// The reference evaluates getBlockAt (intersection.glsl:699-826: SDF spheres, 8-octave fbm, if-chains)
// at EVERY voxel step of EVERY ray.  The engine evaluates it once per voxel on the host and the
// kernels traverse the result.  The bake covers an inclusive voxel-id box chosen per scene such
// that the world outside is exactly the axis-wise extrusion of the box's outermost layer, so a
// clamped lookup reproduces the procedural function for every voxel a ray can visit:
//   cave    hollow = union of 4 spheres, cut by y > 17 (empty above) and the fbm floor (solid
//           for y <= -21): box x[-42,32] y[-21,18] z[-38,31]; sides/bottom layers solid (for
//           y <= 17), top layer empty.
//   Cornell walls at |x|,|y| = 10, z = 25, open front: box x,y[-11,11] z[4,26]; all layers empty.
//   house   walls |x| = 25, |z| = 15, roof y = 5, unbounded floor plane y = -5:
//           box x[-26,26] y[-6,6] z[-16,16]; the floor row of each side layer extrudes to the plane.
// tests/test_host_parity.py checks the clamped lookup against the oracle's procedural getBlockAt on
// a box far larger than the bake.

static void scene_box(int scene, int lo[3], int hi[3])
{
    static const int boxes[3][6] = {
        {-42, -21, -38, 32, 18, 31},
        {-11, -11, 4, 11, 11, 26},
        {-26, -6, -16, 26, 6, 16},
    };
    for (int a = 0; a < 3; ++a)
    {
        lo[a] = boxes[scene][a];
        hi[a] = boxes[scene][3 + a];
    }
}

static SceneBake bake_scene(int scene)
{
    SceneBake b;
    b.scene = scene;
    scene_box(scene, b.lo, b.hi);
    for (int a = 0; a < 3; ++a) b.dim[a] = b.hi[a] - b.lo[a] + 1;
    const size_t n = static_cast<size_t>(b.dim[0]) * b.dim[1] * b.dim[2];
    b.types.assign(n, 0);
    b.bits.assign((n + 31) / 32, 0u);
    size_t i = 0;
    for (int z = b.lo[2]; z <= b.hi[2]; ++z)
        for (int y = b.lo[1]; y <= b.hi[1]; ++y)
            for (int x = b.lo[0]; x <= b.hi[0]; ++x, ++i)
            {
                const int t = block_at(mk3(static_cast<float>(x), static_cast<float>(y), static_cast<float>(z)), scene);
                b.types[i] = static_cast<uint8_t>(t);
                if (t > 0) b.bits[i >> 5] |= (1u << (i & 31));
            }
    // which border layers are entirely empty (march_escaped in the trace kernel)
    unsigned fe = 0;
    for (int axis = 0; axis < 3; ++axis)
        for (int side = 0; side < 2; ++side)
        {
            bool empty = true;
            int c[3];
            const int a1 = (axis + 1) % 3, a2 = (axis + 2) % 3;
            c[axis] = side ? b.hi[axis] : b.lo[axis];
            for (c[a1] = b.lo[a1]; c[a1] <= b.hi[a1] && empty; ++c[a1])
                for (c[a2] = b.lo[a2]; c[a2] <= b.hi[a2]; ++c[a2])
                    if (b.block_at(c[0], c[1], c[2]) > 0)
                    {
                        empty = false;
                        break;
                    }
            if (empty) fe |= 1u << (2 * axis + side);
        }
    b.face_empty = fe;
    return b;
}

int SceneBake::block_at(int x, int y, int z) const
{
    // the cave's fbm floor band continues outside the box (see hit_block_type in ddgi_device.h)
    if (scene == 0 && y < -15 && (x < lo[0] || x > hi[0] || z < lo[2] || z > hi[2]))
        return ddgi::block_at(mk3(static_cast<float>(x), static_cast<float>(y), static_cast<float>(z)), 0);
    x = x < lo[0] ? lo[0] : (x > hi[0] ? hi[0] : x);
    y = y < lo[1] ? lo[1] : (y > hi[1] ? hi[1] : y);
    z = z < lo[2] ? lo[2] : (z > hi[2] ? hi[2] : z);
    const size_t i = (static_cast<size_t>(z - lo[2]) * dim[1] + (y - lo[1])) * dim[0] + (x - lo[0]);
    return types[i];
}

void build_skip_field(const SceneBake& b, int shift, std::vector<uint32_t>& words)
{
    const int nx = b.dim[0], ny = b.dim[1], nz = b.dim[2];
    const size_t n = b.types.size();
    // occupancy dilated by Chebyshev radius 1 and 2: three separable 3-wide maxima per radius, edges replicated (a lookup
    // outside the box reads the border layer, so the neighbour beyond an edge is the edge voxel itself)
    std::vector<uint8_t> lvl[3];
    lvl[0].resize(n);
    for (size_t i = 0; i < n; ++i) lvl[0][i] = b.types[i] ? 1 : 0;
    std::vector<uint8_t> tmp(n);
    auto dilate_axis = [&](const std::vector<uint8_t>& src, std::vector<uint8_t>& dst, int axis) {
        const int dim[3] = {nx, ny, nz};
        const size_t stride[3] = {1, static_cast<size_t>(nx), static_cast<size_t>(nx) * ny};
        for (int z = 0; z < nz; ++z)
            for (int y = 0; y < ny; ++y)
                for (int x = 0; x < nx; ++x)
                {
                    const int c[3] = {x, y, z};
                    const size_t i = (static_cast<size_t>(z) * ny + y) * nx + x;
                    uint8_t v = src[i];
                    if (c[axis] > 0) v |= src[i - stride[axis]];
                    if (c[axis] + 1 < dim[axis]) v |= src[i + stride[axis]];
                    dst[i] = v;
                }
    };
    for (int r = 1; r <= 2; ++r)
    {
        lvl[r].resize(n);
        dilate_axis(lvl[r - 1], lvl[r], 0);
        dilate_axis(lvl[r], tmp, 1);
        dilate_axis(tmp, lvl[r], 2);
    }
    words.assign((n + static_cast<size_t>(shift) + 15) / 16, 0u);
    for (size_t i = 0; i < n; ++i)
    {
        const uint32_t code = lvl[0][i] ? 0u : (lvl[1][i] ? 1u : (lvl[2][i] ? 2u : 3u));
        const size_t k = i + static_cast<size_t>(shift);
        words[k >> 4] |= code << ((k & 15) * 2);
    }
}

const SceneBake& baked_scene(int scene)
{
    static std::array<SceneBake, 3> cache;
    static std::array<std::once_flag, 3> once;
    std::call_once(once[scene], [scene] { cache[scene] = bake_scene(scene); });
    return cache[scene];
}

// ---- memoised lattice hashes ----------------------------------------------------------------------
// Rectangles chosen from the lattice points the three scenes can touch (ddgi_scene.h call sites):
//   noise2D : cave wall   fbm2(0.05, (uv.y+p.y)*0.3): ix in [0,13],  iy in [-1700, 1500]
//             cave ground fbm2(uv*2)                 : ix, iy in [0, 513]
//             moss/mold   interp_noise2D(axis)       : ix, iy in [-1, 2]
//             stem        fbm2(uv.x*5, p.z)          : ix in [0, 1281], |iy| <= 22 * 256 + 1 (stems stand at |z| <= 22);
//                         0.07 stem hits per ray of the C3 workload used to compute their 64 binary64 sines: 9 % of the kernel
//             -> ix in [-2, 1286), iy in [-5888, 5888)   (60 MB; the rows the other block types touch stay L2 resident)
//   noise1  : stem fbm1(p.x): i in [-42*128, 32*128] -> [-8192, 8192)
//   worley  : cell = floor(pixel/5) +- 1 with pixel in [-50, 45] -> cells [-16, 16)
static NoiseLutHost build_noise_lut()
{
    NoiseLutHost t;
    t.n2_x0 = lut::kN2X0, t.n2_nx = lut::kN2NX, t.n2_y0 = lut::kN2Y0, t.n2_ny = lut::kN2NY;
    t.n1_i0 = lut::kN1I0, t.n1_n = lut::kN1N;
    t.wp_c0 = lut::kWpC0, t.wp_n = lut::kWpN;
    t.n2.resize(static_cast<size_t>(t.n2_nx) * t.n2_ny);
    {
        // 15 M binary64 sines: spread over the host's threads (rows are independent)
        const unsigned nthreads = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
        std::vector<std::thread> pool;
        for (unsigned w = 0; w < nthreads; ++w)
            pool.emplace_back([&t, w, nthreads] {
                for (int ix = static_cast<int>(w); ix < t.n2_nx; ix += static_cast<int>(nthreads))
                    for (int iy = 0; iy < t.n2_ny; ++iy)
                        t.n2[static_cast<size_t>(ix) * t.n2_ny + iy] = noise2D(static_cast<float>(ix + t.n2_x0), static_cast<float>(iy + t.n2_y0));
            });
        for (auto& th : pool) th.join();
    }
    t.wall.resize(static_cast<size_t>(8) * t.n2_ny);
    {
        float freq = 1.0f;
        for (int o = 0; o < 8; ++o)
        {
            freq *= 2.0f;
            const float xo = kWallFbmX * freq;
            const size_t ux = static_cast<size_t>(gl_int(floorf(xo)) - t.n2_x0);
            const float tx = gl_fract(xo);
            for (int iy = 0; iy < t.n2_ny; ++iy)
                t.wall[static_cast<size_t>(o) * t.n2_ny + iy] = gl_mix(t.n2[ux * t.n2_ny + iy], t.n2[(ux + 1) * t.n2_ny + iy], tx);
        }
    }
    // random1 over the cave's bake box and a margin around it (cave wall and ground colours hash the voxel id)
    {
        const int lo[3] = {lut::kR1Lo0, lut::kR1Lo1, lut::kR1Lo2};
        const int hi[3] = {lo[0] + lut::kR1N0 - 1, lo[1] + lut::kR1N1 - 1, lo[2] + lut::kR1N2 - 1};
        for (int a = 0; a < 3; ++a) t.r1_lo[a] = lo[a], t.r1_n[a] = hi[a] - lo[a] + 1;
        t.r1.resize(static_cast<size_t>(t.r1_n[0]) * t.r1_n[1] * t.r1_n[2]);
        size_t k = 0;
        for (int z = 0; z < t.r1_n[2]; ++z)
            for (int y = 0; y < t.r1_n[1]; ++y)
                for (int x = 0; x < t.r1_n[0]; ++x)
                    t.r1[k++] = random1(mk3(static_cast<float>(x + lo[0]), static_cast<float>(y + lo[1]), static_cast<float>(z + lo[2])));
    }
    t.n1.resize(t.n1_n);
    for (int i = 0; i < t.n1_n; ++i) t.n1[i] = noise1(static_cast<float>(i + t.n1_i0));
    t.wp.resize(static_cast<size_t>(t.wp_n) * t.wp_n * 2);
    for (int cx = 0; cx < t.wp_n; ++cx)
        for (int cy = 0; cy < t.wp_n; ++cy)
        {
            const f2 q = worley_point_eval(f2{static_cast<float>(cx + t.wp_c0), static_cast<float>(cy + t.wp_c0)});
            t.wp[(static_cast<size_t>(cx) * t.wp_n + cy) * 2 + 0] = q.x;
            t.wp[(static_cast<size_t>(cx) * t.wp_n + cy) * 2 + 1] = q.y;
        }
    return t;
}

const NoiseLutHost& noise_lut_host()
{
    static NoiseLutHost lut;
    static std::once_flag once;
    std::call_once(once, [] { lut = build_noise_lut(); });
    return lut;
}

// ---- glibc rand() ---------------------------------------------------------------------------------
// random_r TYPE_3 as published in glibc stdlib/random_r.c: 31-word additive feedback generator
// x[n] = x[n-31] + x[n-3]; state seeded by the Lehmer LCG 16807 (Schrage's method), the first 310
// outputs discarded, each result shifted right by one.

void GlibcRand::seed(uint32_t s)
{
    if (s == 0) s = 1;
    int32_t word = static_cast<int32_t>(s);
    ring[0] = static_cast<uint32_t>(word);
    for (int i = 1; i < 31; ++i)
    {
        const int32_t hi = word / 127773;
        const int32_t lo = word % 127773;
        word = 16807 * lo - 2836 * hi;
        if (word < 0) word += 2147483647;
        ring[i] = static_cast<uint32_t>(word);
    }
    step = 0;
    for (int i = 0; i < 310; ++i) (void)next();
}

int32_t GlibcRand::next()
{
    const uint32_t f = (step + 3u) % 31u, r = step % 31u;
    ring[f] += ring[r];
    step += 1;
    return static_cast<int32_t>(ring[f] >> 1);
}

// ---- probe rays -------------------------------------------------------------------------------------

void generate_probe_rays(const ddgi_irradiance_field& f, int tile_x, int tile_y, GlibcRand& rng, std::vector<ddgi_probe_ray>& out)
{
    const int cx = f.probe_count[0], cy = f.probe_count[1], cz = f.probe_count[2];
    const int s = tile_x, sh = tile_y;  // the reference: s == sh == sqrt_rays_per_probe
    const int n = s * sh;
    const size_t probes = static_cast<size_t>(cx) * cy * cz;

    // generate_samples (rvpt.cpp:1147-1173): stratified jitter on [0,1)^2 warped to the unit sphere.
    // The reference's host PI is 3.1415926 (rvpt.cpp:1145) and the angle product is formed in
    // double before cosf/sinf narrow it.  Jitter draw order is g++'s: y first, then x (Q1).
    std::vector<f3> dirs(static_cast<size_t>(n));
    const float inv_s = 1.f / static_cast<float>(s), inv_sh = 1.f / static_cast<float>(sh);
    const float rand_max = static_cast<float>(2147483647);
    for (int y = 0, i = 0; y < sh; ++y)
        for (int x = 0; x < s; ++x, ++i)
        {
            const float jy = static_cast<float>(rng.next()) / rand_max;
            const float jx = static_cast<float>(rng.next()) / rand_max;
            const float u = (static_cast<float>(x) + jx) * inv_s;
            const float v = (static_cast<float>(y) + jy) * inv_sh;
            const float z = 1 - (2 * u);
            const float phi = static_cast<float>(2.0 * 3.1415926 * static_cast<double>(v));
            const float rxy = sqrtf(1 - (z * z));
            const pm::SinCos sc = pm::sincos_core(phi);
            // glm::normalize = v * inversesqrt(dot(v,v))  (rvpt.cpp:1212)
            dirs[i] = normalize3(f3{static_cast<float>(sc.c) * rxy, static_cast<float>(sc.s) * rxy, z});
        }

    out.resize(probes * static_cast<size_t>(n));
    ddgi_probe_ray* dst = out.data();
    for (size_t p = 0; p < probes; ++p)
    {
        const int pi = static_cast<int>(p);
        const int py = pi / (cx * cz);
        const int rem = pi - py * cx * cz;
        const int pz = rem / cx;
        const int px = rem - pz * cx;
        // Q4: integer (dim-1)/2, then * side_length (int -> float), then + origin (rvpt.cpp:1199-1205)
        float org[3] = {static_cast<float>(px - (cx - 1) / 2), static_cast<float>(py - (cy - 1) / 2),
                        static_cast<float>(pz - (cz - 1) / 2)};
        for (int a = 0; a < 3; ++a) org[a] = org[a] * static_cast<float>(f.side_length) + f.field_origin[a];
        for (int i = 0; i < n; ++i, ++dst)
        {
            std::memset(dst, 0, sizeof(*dst));
            dst->origin[0] = org[0], dst->origin[1] = org[1], dst->origin[2] = org[2];
            dst->direction[0] = dirs[i].x, dst->direction[1] = dirs[i].y, dst->direction[2] = dirs[i].z;
            dst->probe_info[0] = static_cast<float>(pi);
            dst->probe_info[1] = static_cast<float>(i % s);
            dst->probe_info[2] = static_cast<float>(i / s);
        }
    }
}

void shipped_lights(int scene, LightK* out, int* n)
{
    static const LightK cave[] = {{100.f, {1.f, 1.f, 1.f}, {4.f, 17.5f, 8.5f}}};
    static const LightK cornell[] = {{15.f, {1.f, 1.f, 1.f}, {0.f, 8.f, 13.f}}};
    static const LightK house[] = {{1.f, {1.f, 1.f, 1.f}, {5.f, 9.3f, 36.5f}}, {1.f, {1.f, 1.f, 1.f}, {0.f, 0.f, 0.f}}};
    const LightK* src = nullptr;
    int cnt = 0;
    if (scene == 0) src = cave, cnt = 1;
    else if (scene == 1) src = cornell, cnt = 1;
    else if (scene == 2) src = house, cnt = 2;
    for (int i = 0; i < cnt; ++i) out[i] = src[i];
    *n = cnt;
}

// ---- DDGI mode ------------------------------------------------------------------------------------

static uint32_t host_wang_hash(uint32_t seed)  // probe_pass.comp:45-53
{
    seed = (seed ^ 61u) ^ (seed >> 16);
    seed *= 9u;
    seed = seed ^ (seed >> 4);
    seed *= 0x27d4eb2du;
    seed = seed ^ (seed >> 15);
    return seed;
}

static float host_rand(uint32_t& s)  // probe_pass.comp:59-71
{
    s ^= (s << 13);
    s ^= (s >> 17);
    s ^= (s << 5);
    return static_cast<float>(s) * 0x1.0p-32f;
}

uint32_t frame_key(uint32_t frame) { return host_wang_hash(frame); }

void frame_rotation(uint32_t frame, float* m)
{
    // Shoemake's uniform random unit quaternion from three draws of the reference's own RNG
    const float kTwoPi = 6.2831853071795864769252867665590057683943f;
    uint32_t rng = host_wang_hash(0x9E3779B9u ^ frame);
    const float u1 = host_rand(rng), u2 = host_rand(rng), u3 = host_rand(rng);
    const float a = sqrtf(1.0f - u1), b = sqrtf(u1);
    const pm::SinCos s2 = pm::sincos_core(kTwoPi * u2), s3 = pm::sincos_core(kTwoPi * u3);
    const float qx = a * static_cast<float>(s2.s), qy = a * static_cast<float>(s2.c);
    const float qz = b * static_cast<float>(s3.s), qw = b * static_cast<float>(s3.c);
    m[0] = 1.0f - 2.0f * (qy * qy + qz * qz);
    m[1] = 2.0f * (qx * qy - qz * qw);
    m[2] = 2.0f * (qx * qz + qy * qw);
    m[3] = 2.0f * (qx * qy + qz * qw);
    m[4] = 1.0f - 2.0f * (qx * qx + qz * qz);
    m[5] = 2.0f * (qy * qz - qx * qw);
    m[6] = 2.0f * (qx * qz - qy * qw);
    m[7] = 2.0f * (qy * qz + qx * qw);
    m[8] = 1.0f - 2.0f * (qx * qx + qy * qy);
}

void animate_lights(int scene, float time, const LightK* base, int n, LightK* out)
{
    for (int i = 0; i < n; ++i)
    {
        out[i] = base[i];
        float x = base[i].pos[0], y = base[i].pos[1], z = base[i].pos[2];
        if (scene == 0)
        {
            const float t = 0.05f * time;
            if (i == 0)
                z = z + 10 * pm::cosf_pinned(t * 0.1f);
            else
            {
                const float sn = pm::sinf_pinned(t * 0.5f), cs = pm::cosf_pinned(t * 0.5f);
                x = x + static_cast<float>((i + 1) * 2) * sn;
                y = y + static_cast<float>((i / 2) * 4) * sn;
                z = z + static_cast<float>((i + 1) * 2) * cs;
            }
        }
        else if (scene == 1)
        {
            const float t = 0.005f * time;
            x = x + static_cast<float>(i + 1) * pm::sinf_pinned(t);
            y = y + static_cast<float>((i / 2) * 4) * pm::sinf_pinned(t);
            z = z + static_cast<float>(i + 1) * pm::cosf_pinned(t);
        }
        else if (scene == 2)
        {
            const float d = 0.00005f * time;
            x += d, y += d, z += d;
        }
        out[i].pos[0] = x, out[i].pos[1] = y, out[i].pos[2] = z;
    }
}

}  // namespace ddgi
