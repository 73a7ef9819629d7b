// example_p2p_loop.cpp — the sharded frame loop with ONE PROCESS PER RANK and no RCCL: the ranks exchange their z-slabs by pushing
// them into each other's textures (ddgi_exchange_p2p_*; the textures are mapped with hipIpcOpenMemHandle, the rendezvous is a pair
// of flag words in device memory that the command processor waits on).  The parent forks `world` ranks BEFORE any HIP call, relays
// their 512-byte addresses (the one all-gather the host has to provide) and a barrier before shutdown; every rank prints the
// checksum of the WHOLE field it holds after the last frame — all of them must equal the unsharded engine's.
//   ./example_p2p_loop [frames] [world]      rank r runs on device r % (visible devices): several ranks may share one GPU
// Build: g++ -std=c++17 example_p2p_loop.cpp -L.. -lddgi_probe -L/opt/rocm/lib -lamdhip64
#include <sys/wait.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "rvpt_probe_path.h"

extern "C" int hipGetDeviceCount(int*);

namespace {

struct Wire  // a rank's two pipe ends to the parent
{
    int to_parent, from_parent;
};
bool write_all(int fd, const void* p, size_t n)
{
    const char* c = static_cast<const char*>(p);
    while (n)
    {
        const ssize_t k = write(fd, c, n);
        if (k <= 0) return false;
        c += k, n -= static_cast<size_t>(k);
    }
    return true;
}
bool read_all(int fd, void* p, size_t n)
{
    char* c = static_cast<char*>(p);
    while (n)
    {
        const ssize_t k = read(fd, c, n);
        if (k <= 0) return false;
        c += k, n -= static_cast<size_t>(k);
    }
    return true;
}
// RVPTProbePath::AddressGather over the pipes: mine up, everybody's down.  (bytes == 1: the same relay used as a barrier.)
struct GatherState
{
    Wire w;
    int world;
};
bool gather_through_parent(const uint8_t* mine, uint8_t* all, size_t bytes, void* user)
{
    GatherState* g = static_cast<GatherState*>(user);
    return write_all(g->w.to_parent, mine, bytes) && read_all(g->w.from_parent, all, bytes * static_cast<size_t>(g->world));
}

int run_rank(int rank, int world, int frames, Wire w)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != 0 || ndev < 1)
    {
        std::fprintf(stderr, "no HIP device: this library has no CPU path\n");
        return 1;
    }
    GatherState gs{w, world};
    RVPTProbePath rvpt(rank % ndev, rank, world, gather_through_parent, &gs);
    rvpt.render_settings.scene = 1;  // Cornell box
    rvpt.ir = ddgi_irradiance_field{{2, 2, 8}, 3, 0.9f, 8, {0, 0}, {0.f, 0.f, 15.f}, 1, {0, 0, 0}};
    rvpt.generate_probe_rays();
    if (!rvpt.initialize()) return 1;
    for (int f = 0; f < frames; ++f)
    {
        if (f > 0) rvpt.generate_probe_rays();  // new jitter every frame (the generator's sequence goes on): a stale slab would show
        if (!rvpt.update() || !rvpt.draw()) return 1;
    }
    std::vector<uint8_t> albedo, distance;
    if (!rvpt.read_probe_textures(albedo, distance)) return 1;  // waits for every peer's slab of the last exchange
    unsigned long long sum = 0;
    for (uint8_t v : albedo) sum += v;
    std::printf("rank %d of %d (pid %d, device %d): frames %d, texture bytes %zu, checksum %llu\n", rank, world, static_cast<int>(getpid()), rank % ndev, frames, albedo.size(), sum);
    std::fflush(stdout);
    // nobody unmaps or frees while a peer may still be pushing: a barrier through the parent, then shut down
    uint8_t token = 1;
    std::vector<uint8_t> tokens(static_cast<size_t>(world));
    if (!rvpt.wait() || !gather_through_parent(&token, tokens.data(), 1, &gs)) return 1;
    rvpt.shutdown();
    return 0;
}

}  // namespace

int main(int argc, char** argv)
{
    const int frames = argc > 1 ? std::atoi(argv[1]) : 3;
    const int world = argc > 2 ? std::atoi(argv[2]) : 2;
    if (world < 1 || 8 % world != 0)
    {
        std::fprintf(stderr, "world must divide the grid's 8 z layers\n");
        return 1;
    }
    std::vector<Wire> parent_side(static_cast<size_t>(world));
    std::vector<pid_t> pids(static_cast<size_t>(world));
    for (int r = 0; r < world; ++r)
    {
        int up[2], down[2];
        if (pipe(up) || pipe(down)) return 1;
        const pid_t pid = fork();  // (no HIP call has been made yet: every rank initialises its own runtime)
        if (pid < 0) return 1;
        if (pid == 0)
        {
            close(up[0]), close(down[1]);
            for (int q = 0; q < r; ++q) close(parent_side[static_cast<size_t>(q)].to_parent), close(parent_side[static_cast<size_t>(q)].from_parent);
            return run_rank(r, world, frames, Wire{up[1], down[0]});
        }
        close(up[1]), close(down[0]);
        parent_side[static_cast<size_t>(r)] = Wire{down[1], up[0]};  // (to the rank, from the rank)
        pids[static_cast<size_t>(r)] = pid;
    }
    // relay: the addresses (DDGI_P2P_ADDRESS_BYTES per rank), one 1-byte round per rank (the turns in which the ranks map their peers, one at a time:
    // RVPTProbePath::attach_exchange), then the shutdown barrier (1 byte per rank)
    bool good = true;
    std::vector<size_t> rounds{static_cast<size_t>(DDGI_P2P_ADDRESS_BYTES)};
    rounds.insert(rounds.end(), static_cast<size_t>(world) + 1, static_cast<size_t>(1));
    for (size_t bytes : rounds)
    {
        std::vector<uint8_t> all(bytes * static_cast<size_t>(world));
        for (int r = 0; r < world && good; ++r) good = read_all(parent_side[static_cast<size_t>(r)].from_parent, all.data() + bytes * static_cast<size_t>(r), bytes);
        for (int r = 0; r < world && good; ++r) good = write_all(parent_side[static_cast<size_t>(r)].to_parent, all.data(), all.size());
        if (!good) break;
    }
    int failures = good ? 0 : 1;
    for (int r = 0; r < world; ++r)
    {
        close(parent_side[static_cast<size_t>(r)].to_parent), close(parent_side[static_cast<size_t>(r)].from_parent);  // (a rank blocked on the relay sees EOF)
        int status = 0;
        waitpid(pids[static_cast<size_t>(r)], &status, 0);
        failures += !(WIFEXITED(status) && WEXITSTATUS(status) == 0);
    }
    return failures ? 1 : 0;
}
