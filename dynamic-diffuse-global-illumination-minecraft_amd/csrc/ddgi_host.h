// ddgi_host.h — host-side pieces of the probe path: scene bake, probe-ray generation.
#pragma once

#include <cstdint>
#include <vector>

#include "ddgi_types.h"

namespace ddgi {

// Baked voxel scene (host copy).  See ddgi_host.cpp: bake_scene().
struct SceneBake
{
    int scene = -1;
    int lo[3] = {0, 0, 0};
    int hi[3] = {0, 0, 0};
    int dim[3] = {0, 0, 0};
    unsigned face_empty = 0;
    std::vector<uint32_t> bits;  // 1 bit / voxel, linear index (z*ny + y)*nx + x
    std::vector<uint8_t> types;  // block type 0..13 / voxel
    int block_at(int x, int y, int z) const;  // clamped lookup = the world the kernels see
};

const SceneBake& baked_scene(int scene);  // cached, thread-safe; scene in {0,1,2}

// The fast (tolerance-mode) march's per-voxel skip field, 2 bits per voxel in the bitmap's order: 0 = occupied, else
// 1 + min(r, 2) with r the voxel's free Chebyshev radius — every voxel within r of it, in the world the kernels see
// (clamped lookups: outside the box the border layer repeats), is empty.  Stored shifted by `shift` entries like the
// occupancy bitmap (ddgi_engine.cpp: ensure_scene), 16 entries per word.
void build_skip_field(const SceneBake& b, int shift, std::vector<uint32_t>& words);

// glibc rand() (random_r TYPE_3) restated: the reference draws its ray jitter from the unseeded C
// library generator (src/rvpt/rvpt.cpp:1161-1162).
struct GlibcRand
{
    uint32_t ring[31];
    uint32_t step = 0;
    void seed(uint32_t s);
    int32_t next();
};

// generate_samples + RVPT::generate_probe_rays (src/rvpt/rvpt.cpp:1147-1224): full-grid array in
// the reference's order (probe-major p = py*cx*cz + pz*cx + px, ray i = y*s + x).
// tile_x x tile_y: the ray tile (the reference: both = sqrt_rays_per_probe; see ddgi_set_ray_tile)
void generate_probe_rays(const ddgi_irradiance_field& f, int tile_x, int tile_y, GlibcRand& rng, std::vector<ddgi_probe_ray>& out);

// Host copy of the memoised lattice hashes (ddgi_scene.h: NoiseLut).  Scene independent.
struct NoiseLutHost
{
    int n2_x0, n2_nx, n2_y0, n2_ny;
    int n1_i0, n1_n;
    int wp_c0, wp_n;
    int r1_lo[3], r1_n[3];
    std::vector<float> n2, n1, wp, wall, r1;
};
const NoiseLutHost& noise_lut_host();  // cached, thread-safe

// Default light tables (assets/shaders/structs.glsl:61-89, the shipped ones).
void shipped_lights(int scene, LightK* out, int* n);

// DDGI mode host pieces -------------------------------------------------------------------------
// update_lights (assets/shaders/probe_pass.comp:217-251, dormant): light positions as a function
// of RenderSettings::time, applied to the base table.
void animate_lights(int scene, float time, const LightK* base, int n, LightK* out);
// the frame's random rotation of the ray set (row-major 3x3) and its RNG key
void frame_rotation(uint32_t frame, float* m9);
uint32_t frame_key(uint32_t frame);

}  // namespace ddgi
