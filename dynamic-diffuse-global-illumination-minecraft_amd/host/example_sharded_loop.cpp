// example_sharded_loop.cpp — the reference's frame loop (src/rvpt/main.cpp:37-100) reduced to the probe
// path and sharded by z-slab over the GPUs of one node, with NO Python in the loop: one process drives one
// RVPTProbePath per visible device (at most `max_gpus`, and only as many as divide the grid's z count), each
// tracing its slab; draw() issues the RCCL all-gather inside libddgi_probe.so.  Every rank then holds the
// whole field: the checksum printed for each rank must equal the single-GPU checksum of example_probe_loop.
// Build: g++ -std=c++17 example_sharded_loop.cpp -L.. -lddgi_probe -L/opt/rocm/lib -lamdhip64
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <vector>

#include "rvpt_probe_path.h"

extern "C" int hipGetDeviceCount(int*);

int main(int argc, char** argv)
{
    const int frames = argc > 1 ? std::atoi(argv[1]) : 3;
    const int max_gpus = argc > 2 ? std::atoi(argv[2]) : 8;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != 0 || ndev < 1)
    {
        std::fprintf(stderr, "no HIP device: this library has no CPU path\n");
        return 1;
    }
    const ddgi_irradiance_field field{{2, 2, 8}, 3, 0.9f, 8, {0, 0}, {0.f, 0.f, 15.f}, 1, {0, 0, 0}};
    int world = ndev < max_gpus ? ndev : max_gpus;
    while (field.probe_count[2] % world) --world;

    std::vector<int> devices(world);
    for (int r = 0; r < world; ++r) devices[r] = r;
    std::vector<void*> comms(world, nullptr);
    if (ddgi_comm_create_all(world, devices.data(), comms.data()) != DDGI_OK)  // ncclCommInitAll
    {
        std::fprintf(stderr, "communicator: %s\n", ddgi_last_error());
        return 1;
    }
    std::vector<std::unique_ptr<RVPTProbePath>> slabs;
    for (int r = 0; r < world; ++r)
    {
        slabs.emplace_back(new RVPTProbePath(devices[r], r, world, comms[r]));
        slabs[r]->render_settings.scene = 1;  // Cornell box
        slabs[r]->ir = field;
        slabs[r]->generate_probe_rays();
        if (!slabs[r]->initialize()) return 1;
    }
    for (int f = 0; f < frames; ++f)
    {
        for (auto& s : slabs)
            if (!s->update()) return 1;
        // one thread issues the collectives of every rank: group them (ncclGroupStart/End)
        if (ddgi_exchange_group_begin() != DDGI_OK) return 1;
        bool good = true;
        for (auto& s : slabs) good = s->draw() && good;
        if (ddgi_exchange_group_end() != DDGI_OK || !good) return 1;
    }
    for (int r = 0; r < world; ++r)
    {
        std::vector<uint8_t> albedo, distance;
        if (!slabs[r]->read_probe_textures(albedo, distance)) return 1;  // waits for the latest exchange
        unsigned long long sum = 0;
        for (uint8_t v : albedo) sum += v;
        std::printf("rank %d of %d: frames %d, texture bytes %zu, checksum %llu\n", r, world, frames, albedo.size(), sum);
    }
    for (auto& s : slabs) s->shutdown();
    for (void* c : comms) (void)ddgi_comm_destroy(c);
    return 0;
}
