// example_probe_loop.cpp — the reference's frame loop (src/rvpt/main.cpp:37-100) reduced to the
// probe path, on RVPTProbePath.  Build: g++ -std=c++17 example_probe_loop.cpp -L.. -lddgi_probe
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "rvpt_probe_path.h"

int main(int argc, char** argv)
{
    RVPTProbePath rvpt;
    rvpt.render_settings.scene = 1;  // Cornell box
    rvpt.ir = ddgi_irradiance_field{{2, 2, 2}, 6, 0.9f, 8, {0, 0}, {0.f, 0.f, 15.f}, 1, {0, 0, 0}};
    rvpt.generate_probe_rays();  // main.cpp:47
    if (!rvpt.initialize()) return 1;
    const int frames = argc > 1 ? std::atoi(argv[1]) : 3;
    for (int f = 0; f < frames; ++f)
        if (!rvpt.update() || !rvpt.draw()) return 1;  // main.cpp:93-95
    std::vector<uint8_t> albedo, distance;
    if (!rvpt.read_probe_textures(albedo, distance)) return 1;
    unsigned long long sum = 0;
    for (uint8_t v : albedo) sum += v;
    std::printf("frames %d, texture bytes %zu, checksum %llu\n", frames, albedo.size(), sum);
    rvpt.shutdown();
    return 0;
}
