// ddgi_oct.h — DDGI-mode building blocks (host + device): the pieces the reference leaves dormant.
//   octahedral mapping         assets/shaders/octahedral.glsl:13-34 (never #included by the reference)
//   animated lights            assets/shaders/probe_pass.comp:217-251 (update_lights; call commented out)
//   spherical Fibonacci rays   not in the reference: DDGI paper (README.md:45; Majercik et al. 2019)
//   tile texel directions, border wrap, depth sharpening: DDGI paper supplemental
// Pinned arithmetic as everywhere else (DESIGN.md "Arithmetic pinning").
#pragma once

#include "ddgi_scene.h"
#include "ddgi_types.h"

namespace ddgi {

constexpr int kIrrTile = 8;   // 6x6 interior + 1 texel border, rgba f32
constexpr int kDepTile = 16;  // 14x14 interior + border, (mean d, mean d^2) f32
constexpr float kMissDistance = 1e27f;

DDGI_HD float sign_not_zero(float v) { return v >= 0.0f ? 1.0f : -1.0f; }

DDGI_HD f2 oct_encode(f3 v)  // octahedral.glsl:16-23
{
    const float inv = 1.0f / (fabsf(v.x) + fabsf(v.y) + fabsf(v.z));
    f2 r{v.x * inv, v.y * inv};
    if (v.z < 0.0f) r = f2{(1.0f - fabsf(r.y)) * sign_not_zero(r.x), (1.0f - fabsf(r.x)) * sign_not_zero(r.y)};
    return r;
}

DDGI_HD f3 oct_decode(f2 o)  // octahedral.glsl:28-34
{
    f3 v{o.x, o.y, 1.0f - fabsf(o.x) - fabsf(o.y)};
    if (v.z < 0.0f)
    {
        const float nx = (1.0f - fabsf(v.y)) * sign_not_zero(v.x);
        const float ny = (1.0f - fabsf(v.x)) * sign_not_zero(v.y);
        v.x = nx, v.y = ny;
    }
    return normalize3(v);
}

// direction of interior texel (x, y) in [1, side-2]^2 of a bordered side x side tile
DDGI_HD f3 texel_dir(int x, int y, int side)
{
    const float inner = static_cast<float>(side - 2);
    const f2 uv{(static_cast<float>(x - 1) + 0.5f) / inner * 2.0f - 1.0f, (static_cast<float>(y - 1) + 0.5f) / inner * 2.0f - 1.0f};
    return oct_decode(uv);
}

// interior texel a border texel (x, y) copies from (octahedral wrap)
DDGI_HD void border_source(int x, int y, int side, int& sx, int& sy)
{
    const int last = side - 1;
    const bool ex = (x == 0 || x == last), ey = (y == 0 || y == last);
    if (ex && ey)
    {
        sx = x == 0 ? last - 1 : 1;
        sy = y == 0 ? last - 1 : 1;
    }
    else if (ey)
    {
        sx = last - x;
        sy = y == 0 ? 1 : last - 1;
    }
    else
    {
        sx = x == 0 ? 1 : last - 1;
        sy = last - y;
    }
}

// sphericalFibonacci(i, n) rotated by the frame's row-major 3x3 matrix
DDGI_HD f3 fibonacci_dir(int i, int n, const float* m9)
{
    const float kTwoPi = 6.2831853071795864769252867665590057683943f;
    const float kPhiM1 = 0.6180339887498948482045868343656381f;
    const float fi = static_cast<float>(i);
    const float fr = fmaf(fi, kPhiM1, -floorf(fi * kPhiM1));
    const pm::SinCos sc = pm::sincos_core(kTwoPi * fr);
    const float cos_t = 1.0f - (2.0f * fi + 1.0f) / static_cast<float>(n);
    const float sin_t = sqrtf(gl_clamp(1.0f - cos_t * cos_t, 0.0f, 1.0f));
    const f3 d{static_cast<float>(sc.c) * sin_t, static_cast<float>(sc.s) * sin_t, cos_t};
    return f3{dot3(f3{m9[0], m9[1], m9[2]}, d), dot3(f3{m9[3], m9[4], m9[5]}, d), dot3(f3{m9[6], m9[7], m9[8]}, d)};
}

DDGI_HD float pow50(float x)  // depth sharpness 50 as an exact multiplication chain
{
    const float x2 = x * x, x4 = x2 * x2, x8 = x4 * x4, x16 = x8 * x8, x32 = x16 * x16;
    return (x32 * x16) * x2;
}

// world position of probe (px, py, pz): RVPT::generate_probe_rays, src/rvpt/rvpt.cpp:1199-1205
DDGI_HD f3 probe_position(const GridK& G, int px, int py, int pz)
{
    const float side = static_cast<float>(G.side);
    return f3{static_cast<float>(px - (G.cx - 1) / 2) * side + G.origin[0], static_cast<float>(py - (G.cy - 1) / 2) * side + G.origin[1],
              static_cast<float>(pz - (G.cz - 1) / 2) * side + G.origin[2]};
}

}  // namespace ddgi
