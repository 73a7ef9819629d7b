// isa_probes.hip — instruction-count probes: the building blocks of a trace-kernel event as kernels of their own.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -S -o - tools/microbench/isa_probes.hip | less
// (kept out of the library's sources: nothing here ships)
#include "../../dynamic-diffuse-global-illumination-minecraft_amd/csrc/ddgi_trace_wf.hip"

// instruction-count probes (hipcc -S -DDDGI_ISA_PROBE): the building blocks of an event as kernels of their own
namespace ddgi {
__global__ void k_isa_post_march(const TraceArgs A, float* io, uint32_t* lds_src)
{
    extern __shared__ uint32_t pl[];
    WfPool P;
    float* f = reinterpret_cast<float*>(pl);
    for (int a = 0; a < 3; ++a) P.ro[a] = f + 1536 * a, P.dn[a] = f + 1536 * (3 + a);
    P.t = f + 1536 * 6, P.tl = f + 1536 * 7, P.flags = pl + 1536 * 8;
    for (int a = 0; a < 3; ++a) P.col[a] = f + 1536 * (10 + a), P.rd[a] = f + 1536 * (17 + a);
    P.rng = pl + 1536 * 13, P.cnt = pl + 1536 * 14, P.dst = pl + 1536 * 15;
    P.cold = reinterpret_cast<WfColdGlobal*>(io);
    WfCold c = load_cold(P, threadIdx.x, true);
    const f3 o = v3of(c.hn), d = v3of(c.hc);
    const int r = wf_post_march<CfgPlain<0>>(P, threadIdx.x, c, o, d, c.cnt != 0, A, pl + 1536 * 16);
    store_cold(P, threadIdx.x, c, true);
    io[threadIdx.x] = static_cast<float>(r);
}
__global__ void k_isa_hemisphere(float* io)
{
    uint32_t rng = __float_as_uint(io[threadIdx.x + 512]);
    const f3 d = hemisphere_dir(f3{io[threadIdx.x], io[threadIdx.x + 64], io[threadIdx.x + 128]}, rng);
    io[threadIdx.x] = d.x, io[threadIdx.x + 64] = d.y, io[threadIdx.x + 128] = d.z, io[threadIdx.x + 512] = __uint_as_float(rng);
}
__global__ void k_isa_light_spheres(const TraceArgs A, float* io)
{
    float tl;
    int lid;
    light_spheres<1>(f3{io[threadIdx.x], io[threadIdx.x + 64], io[threadIdx.x + 128]}, f3{io[threadIdx.x + 192], io[threadIdx.x + 256], io[threadIdx.x + 320]}, A, tl, lid);
    io[threadIdx.x] = tl, io[threadIdx.x + 64] = static_cast<float>(lid);
}
__global__ void k_isa_normalize(float* io)
{
    const f3 d = normalize3(f3{io[threadIdx.x], io[threadIdx.x + 64], io[threadIdx.x + 128]});
    io[threadIdx.x] = d.x, io[threadIdx.x + 64] = d.y, io[threadIdx.x + 128] = d.z;
}
__global__ void k_isa_albedo_wall(const TraceArgs A, float* io)
{
    const f3 d = block_albedo(f3{io[threadIdx.x], io[threadIdx.x + 64], io[threadIdx.x + 128]}, 10, f3{1.0f, 0.0f, 0.0f}, A.noise);
    io[threadIdx.x] = d.x, io[threadIdx.x + 64] = d.y, io[threadIdx.x + 128] = d.z;
}
__global__ void k_isa_step(const TraceArgs A, float* io, int n)
{
    extern __shared__ uint32_t pl[];
    March m;
    m.ro = f3{io[threadIdx.x], io[threadIdx.x + 64], io[threadIdx.x + 128]};
    m.dn = f3{io[threadIdx.x + 192], io[threadIdx.x + 256], io[threadIdx.x + 320]};
    m.inv = f3{io[threadIdx.x + 384], io[threadIdx.x + 448], io[threadIdx.x + 512]};
    m.cc = f3{io[threadIdx.x + 576], io[threadIdx.x + 640], io[threadIdx.x + 704]};
    m.t = 0, m.tl = io[threadIdx.x + 768], m.p = m.ro, m.it = 0, m.cell = 0, m.lid = 0, m.rd = m.dn;
    f3 hi = f3{A.scene.hi_f[0], A.scene.hi_f[1], A.scene.hi_f[2]};
    asm volatile("" : "+v"(hi.x), "+v"(hi.y), "+v"(hi.z));
    bool occ = false;
    for (int i = 0; i < n; ++i)
    {
        asm volatile("; STEP BEGIN");
        occ = march_step_burst(m, A.scene, pl, hi);
        asm volatile("; STEP END");
        if (occ | (m.t >= m.tl)) break;
    }
    io[threadIdx.x] = m.t + (occ ? 1.0f : 0.0f);
}
}  // namespace ddgi
