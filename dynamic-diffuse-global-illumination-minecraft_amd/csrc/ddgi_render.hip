// ddgi_render.hip — k_render_primary: the primary-visibility consumer of the probe field
// (SURVEY.md §8f row 1): one lane per pixel builds the camera ray (assets/shaders/camera.glsl:29-74),
// intersects the scene (intersection.glsl:1244-1301) and evaluates one of the reference's
// integrators (integrators.glsl:27-271, dispatched by compute_pass.comp:58-87): 0 DDGI (direct +
// probe-field indirect), 1 direct, 2 indirect, 3 colour, 4 normal, 5 reciprocal depth.  The
// indirect term is the same per-point cage sample the batched kernels use (ddgi_sampler.h), in
// whichever mode the engine is in.  RenderSettings::visualize_probes draws the probes as spheres
// (intersect_probes, intersection.glsl:1102-1128, for integrators 0 and 2, integrators.glsl:45-65, 180-199).
// Debug views: render_mode 6 = the whole probe texture on screen (the reference's dormant
// get_probe_image_coords blit, compute_pass.comp:116-124, 185-190), 7 = the cage's first probe index as a colour
// (README.md:89-91).
#include "ddgi_device.h"
#include "ddgi_sampler.h"

namespace ddgi {

struct Hit
{
    bool any;
    int type;  // 2 light, 3 block
    int lid;
    float t;
    f3 pos, normal, base;
};

// intersect_scene for a single ray, ray-per-lane (one march per call)
DDGI_D Hit intersect_scene_dev(f3 o, f3 d, const TraceArgs& T, const uint32_t* s_bits, bool want_albedo)
{
    March m;
    start_march(m, o, d, T);
    bool occ = false;
    for (int i = 0; i < kMarchIters; ++i)
    {
        occ = march_step(m, T.scene, s_bits);
        if (occ || m.t >= m.tl) break;
        if ((i & 7) == 7 && march_escaped(m, T.scene)) break;
    }
    Hit h;
    const float inf = __builtin_inff();
    const bool block_wins = occ && (m.t < m.tl);
    h.any = block_wins || (m.tl < inf);
    h.type = block_wins ? 3 : 2;
    h.lid = m.lid;
    h.t = inf;
    h.pos = h.normal = h.base = mk3(0, 0, 0);
    if (!h.any) return h;
    f3 nraw;
    if (block_wins)
    {
        h.t = m.t;
        const f3 cell = cell_id(m.p);
        const f3 diff = normalize3(m.p - f3{cell.x - 0.5f, cell.y - 0.5f, cell.z - 0.5f});
        f3 n = mk3(0, 0, 0);
        float best = 0.0f;
        if (fabsf(diff.x) > best) { best = fabsf(diff.x); n = mk3(gl_sign(diff.x), 0, 0); }
        if (fabsf(diff.y) > best) { best = fabsf(diff.y); n = mk3(0, gl_sign(diff.y), 0); }
        if (fabsf(diff.z) > best) { best = fabsf(diff.z); n = mk3(0, 0, gl_sign(diff.z)); }
        const f3 nn = normalize3(n);
        if (want_albedo) h.base = block_albedo(m.p, hit_block_type(T.scene, T.scene_id, cell, m.cell), nn, T.noise);
        nraw = nn;
    }
    else
    {
        h.t = m.tl;
        const LightK& L = T.lights[m.lid];
        nraw = ray_at((o - f3{L.pos[0], L.pos[1], L.pos[2]}) * 10.0f, d * 10.0f, h.t);
    }
    h.normal = normalize3(nraw);
    h.pos = ray_at(o, d, h.t) + h.normal * 0.001f;
    return h;
}

// the direct-light loop of integrator_DDGI (:77-96) / integrator_direct (:131-149)
DDGI_D int direct_light(const Hit& info, const TraceArgs& T, const uint32_t* s_bits, f3& direct)
{
    direct = mk3(0, 0, 0);
    int nvis = 0;
    for (int i = 0; i < T.nl; ++i)
    {
        const LightK& L = T.lights[i];
        const f3 lp{L.pos[0], L.pos[1], L.pos[2]};
        const Hit f = intersect_scene_dev(info.pos, normalize3(lp - info.pos), T, s_bits, false);
        if (f.any && f.type == 2)
        {
            const float lambert = gl_clamp(dot3(normalize3(info.normal), normalize3(lp - info.pos)), 0.0f, 1.0f);
            const float dist = length3(lp - info.pos);
            direct = direct + div3((f3{L.col[0], L.col[1], L.col[2]} * lambert) * L.intensity, dist);
            nvis += 1;
        }
    }
    return nvis;
}

// sceneSDF / opRepLim (intersection.glsl:333-347): distance to the nearest probe sphere (radius 0.2) of the clamped
// lattice c * clamp(round(p / c), -l, l), l = vec3(probeCount / 2) (integer division); round(): nearest-even (pinned)
DDGI_D float probes_sdf(const GridK& G, f3 point)
{
    const f3 p = point - f3{G.origin[0], G.origin[1], G.origin[2]};
    const float c = static_cast<float>(G.side);
    const float qx = p.x - c * gl_clamp(rintf(p.x / c), -static_cast<float>(G.cx / 2), static_cast<float>(G.cx / 2));
    const float qy = p.y - c * gl_clamp(rintf(p.y / c), -static_cast<float>(G.cy / 2), static_cast<float>(G.cy / 2));
    const float qz = p.z - c * gl_clamp(rintf(p.z / c), -static_cast<float>(G.cz / 2), static_cast<float>(G.cz / 2));
    return length3(f3{qx, qy, qz}) - 0.2f;
}

// implicit_surface (intersection.glsl:367-392): sphere tracing while curr_t < 100 along normalize(direction)
DDGI_D bool probes_hit(const GridK& G, f3 o, f3 d, float& t_out)
{
    const f3 dir = normalize3(d);
    float t = 0.0f;
    while (t < 100.0f)
    {
        const float dist = probes_sdf(G, ray_at(o, dir, t));
        if (dist < 0.001f)
        {
            t_out = t;
            return true;
        }
        t += dist;
    }
    return false;
}

__global__ __launch_bounds__(256) void k_render_primary(const RenderArgs A)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t render_lds[];
    float* s_unorm = reinterpret_cast<float*>(render_lds);  // 256 entries
    uint32_t* s_bits = render_lds + 256;
    const TraceArgs& T = A.trace;
    s_unorm[threadIdx.x] = static_cast<float>(threadIdx.x) / 255.0f;
    for (int i = threadIdx.x; i < T.scene.nwords; i += 256) s_bits[i] = T.scene.bits[i];
    __syncthreads();

    const uint32_t k = blockIdx.x * 256u + threadIdx.x;
    if (k >= static_cast<uint32_t>(A.width) * static_cast<uint32_t>(A.height)) return;
    const int px = static_cast<int>(k % static_cast<uint32_t>(A.width)), py = static_cast<int>(k / static_cast<uint32_t>(A.width));
    // compute_pass.comp:178-181: coord = gid / dim, flipped vertically
    const float x = static_cast<float>(px) / static_cast<float>(A.width);
    const float y = 1.0f - static_cast<float>(py) / static_cast<float>(A.height);
    const float* M = A.cam_matrix;
    const float aspect = A.cam_params[0];
    const float u = aspect * (2.0f * x - 1.0f), v = 2.0f * y - 1.0f;
    auto mul = [&](float a, float b, float c, float w) {
        return f3{((M[0] * a + M[4] * b) + M[8] * c) + M[12] * w, ((M[1] * a + M[5] * b) + M[9] * c) + M[13] * w,
                  ((M[2] * a + M[6] * b) + M[10] * c) + M[14] * w};
    };
    f3 ro, rd;
    if (A.camera_mode == 1)  // camera_ortho_ray
    {
        const float s = A.cam_params[2];
        ro = mul(s * u, s * v, 0.0f, 1.0f);
        rd = f3{M[8], M[9], M[10]};
    }
    else  // camera_pinhole_ray; w = 1/tan(hfov/2) comes from the host (pinned tan = sin/cos)
    {
        ro = f3{M[12], M[13], M[14]};
        rd = normalize3(mul(u, v, A.pinhole_w, 0.0f));
    }

    if (A.render_mode == 6)  // the probe texture on screen (REF mode; get_probe_image_coords works on the unflipped pixel)
    {
        f3 texel = mk3(0, 0, 0);
        const GridK& G = T.grid;
        const int W = G.cx * G.cz * G.sx, H = G.cy * G.sy;
        const int tx = gl_int((static_cast<float>(px) * static_cast<float>(W)) / static_cast<float>(A.width));
        const int ty = gl_int((static_cast<float>(py) * static_cast<float>(H)) / static_cast<float>(A.height));
        if (A.albedo && tx >= 0 && tx < W && ty >= 0 && ty < H)
        {
            // raster (tx, ty) -> probe p = row * cx*cz + column, texel inside its tile -> the slab-major buffer
            const int p = (ty / G.sy) * (G.cx * G.cz) + tx / G.sx;
            texel = load_rgb(A.albedo, static_cast<size_t>(slab_slot(G, p)) * G.n + (ty % G.sy) * G.sx + tx % G.sx, s_unorm);
        }
        A.rgba8[k] = unorm8(texel.x) | (unorm8(texel.y) << 8) | (unorm8(texel.z) << 16) | (255u << 24);
        if (A.rgb_f32) A.rgb_f32[3 * k] = texel.x, A.rgb_f32[3 * k + 1] = texel.y, A.rgb_f32[3 * k + 2] = texel.z;
        return;
    }
    const Hit info = intersect_scene_dev(ro, rd, T, s_bits, true);
    f3 out = mk3(0, 0, 0);
    int cage[8];
    bool probe_seen = false;
    if (A.visualize_probes && (A.render_mode == 0 || A.render_mode == 2))
    {
        float pt;
        probe_seen = probes_hit(T.grid, ro, rd, pt) && pt < info.t;  // (info.t is INF when nothing was hit)
    }
    auto gi = [&]() {
        return A.irradiance ? diffuse_gi_ddgi(T.grid, A.irradiance, A.depth, info.pos, info.normal, cage)
                            : (A.box ? diffuse_gi_ref<true>(T.grid, A.albedo, info.pos, info.normal, s_unorm, cage, A.box)
                                     : diffuse_gi_ref<false>(T.grid, A.albedo, info.pos, info.normal, s_unorm, cage, nullptr));
    };
    if (probe_seen) out = mk3(0, 1, 1);  // probe colour, integrators.glsl:65,199
    else switch (A.render_mode)
    {
        case 7:  // the cage's first probe index as a colour; magenta outside the field
            if (info.any)
            {
                (void)gi();
                if (cage[0] < 0) out = mk3(1, 0, 1);
                else
                {
                    const uint32_t h = (static_cast<uint32_t>(cage[0]) * 2654435761u) & 0xffffffu;
                    out = f3{static_cast<float>((h >> 16) & 255u) / 255.0f, static_cast<float>((h >> 8) & 255u) / 255.0f, static_cast<float>(h & 255u) / 255.0f};
                }
            }
            break;
        case 1:
        {
            f3 direct;
            const int nvis = info.any ? direct_light(info, T, s_bits, direct) : 0;
            if (nvis != 0) out = (info.base * 0.5f) * div3(direct, static_cast<float>(nvis));
            break;
        }
        case 2:
            if (info.any) out = gi() * 0.5f;
            break;
        case 3:
            if (info.any) out = info.base;
            break;
        case 4:
        {
            const float h = info.any ? 1.0f : 0.0f;
            out = f3{0.5f * info.normal.x + 0.5f * h, 0.5f * info.normal.y + 0.5f * h, 0.5f * info.normal.z + 0.5f * h};
            break;
        }
        case 5:
        {
            const float inv = 1.0f / (length3(rd) * info.t);
            out = mk3(inv, inv, inv);
            break;
        }
        default:
        {
            if (!info.any) out = mk3(0.898f, 0.968f, 1.0f);
            else if (info.type == 2)
            {
                const LightK& L = T.lights[info.lid];
                out = f3{L.col[0], L.col[1], L.col[2]};
            }
            else
            {
                const f3 indirect = gi();
                f3 direct;
                const int nvis = direct_light(info, T, s_bits, direct);
                const f3 half_base = info.base * 0.5f;
                if (nvis != 0) out = half_base * div3(direct, static_cast<float>(nvis)) + half_base * indirect;
                else out = (indirect * 0.5f) * info.base;
            }
        }
    }
    A.rgba8[k] = unorm8(out.x) | (unorm8(out.y) << 8) | (unorm8(out.z) << 16) | (255u << 24);
    if (A.rgb_f32) A.rgb_f32[3 * k] = out.x, A.rgb_f32[3 * k + 1] = out.y, A.rgb_f32[3 * k + 2] = out.z;
}

hipError_t launch_render_primary(const RenderArgs& args, hipStream_t stream)
{
    const unsigned n = static_cast<unsigned>(args.width) * static_cast<unsigned>(args.height);
    if (n == 0) return hipSuccess;
    const size_t lds = (256 + static_cast<size_t>(args.trace.scene.nwords)) * sizeof(uint32_t);
    if (lds > 64 * 1024)  // a large user scene: opt in to more dynamic LDS (ddgi_render_device has checked the 160 KB bound)
    {
        hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(k_render_primary), 160 * 1024);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k_render_primary, dim3((n + 255u) / 256u), dim3(256), lds, stream, args);
    return hipGetLastError();
}

}  // namespace ddgi
