// ipc_open_cost.hip — what does bringing the peer-to-peer transport up COST?  W processes on one GPU (what the one-GPU
// test box runs; on a node each process would own a GPU), each hipMalloc's one buffer of B bytes, exports it and opens
// every peer's: seconds per hipIpcOpenMemHandle against bytes and against the number of processes opening at once, the
// first 4 KB copy into a freshly mapped buffer (hipIpcMemLazyEnablePeerAccess: the mapping may be finished lazily) and a
// copy of a G-th of the buffer (a slab).  Round 5's verdict: "nobody has measured seconds per mapped GB".
//   hipcc --offload-arch=gfx950 -O2 ipc_open_cost.hip -o ipc_open_cost.bin && ./ipc_open_cost.bin [limit_s]
// The parent never touches HIP (it only forwards the handles); a size that is not done within limit_s (default 120) is
// reported as such and ends the run.
#include <hip/hip_runtime.h>
#include <poll.h>
#include <signal.h>
#include <sys/wait.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

static double now_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

struct Report
{
    double malloc_s, export_s, open_sum_s, open_max_s, first_touch_max_s, slab_copy_s, close_s;
    int rc;
};

static bool read_all(int fd, void* p, size_t n)
{
    char* c = static_cast<char*>(p);
    while (n)
    {
        ssize_t r = read(fd, c, n);
        if (r <= 0) return false;
        c += r, n -= static_cast<size_t>(r);
    }
    return true;
}
static bool write_all(int fd, const void* p, size_t n)
{
    const char* c = static_cast<const char*>(p);
    while (n)
    {
        ssize_t r = write(fd, c, n);
        if (r <= 0) return false;
        c += r, n -= static_cast<size_t>(r);
    }
    return true;
}

#define CK(x)                                                                                    \
    do                                                                                           \
    {                                                                                            \
        hipError_t e_ = (x);                                                                     \
        if (e_ != hipSuccess)                                                                    \
        {                                                                                        \
            std::fprintf(stderr, "[rank %d] %s -> %s\n", rank, #x, hipGetErrorString(e_));       \
            rep.rc = 1;                                                                          \
            goto out;                                                                            \
        }                                                                                        \
    } while (0)

static int child(int rank, int world, int rfd, int wfd, const std::vector<size_t>& sizes)
{
    if (hipSetDevice(0) != hipSuccess) return 3;
    hipStream_t s;
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return 3;
    for (size_t bytes : sizes)
    {
        Report rep;
        std::memset(&rep, 0, sizeof rep);
        void *own = nullptr, *own2 = nullptr;
        std::vector<void*> mapped(static_cast<size_t>(world), nullptr), mapped2(static_cast<size_t>(world), nullptr);
        std::vector<hipIpcMemHandle_t> all2(static_cast<size_t>(world));
        std::vector<hipIpcMemHandle_t> all(static_cast<size_t>(world));
        hipIpcMemHandle_t mine;
        double t0 = now_s();
        CK(hipMalloc(&own, bytes));
        CK(hipMemset(own, rank + 1, bytes));
        CK(hipDeviceSynchronize());
        rep.malloc_s = now_s() - t0;
        t0 = now_s();
        CK(hipIpcGetMemHandle(&mine, own));
        rep.export_s = now_s() - t0;
        if (!write_all(wfd, &mine, sizeof mine) || !read_all(rfd, all.data(), sizeof(hipIpcMemHandle_t) * all.size())) return 4;
        if (std::getenv("IPC_PAIR"))
        {
            // the engine's shape: a SECOND buffer of twice the size per process (the depth tiles' ring), exported and opened after the first
            CK(hipMalloc(&own2, 2 * bytes));
            CK(hipMemsetAsync(own2, 0, 2 * bytes, s));
            CK(hipStreamSynchronize(s));
            CK(hipIpcGetMemHandle(&mine, own2));
            if (!write_all(wfd, &mine, sizeof mine) || !read_all(rfd, all2.data(), sizeof(hipIpcMemHandle_t) * all2.size())) return 4;
        }
        for (int step = 1; step < world; ++step)
        {
            const int q = (rank + step) % world;
            t0 = now_s();
            CK(hipIpcOpenMemHandle(&mapped[static_cast<size_t>(q)], all[static_cast<size_t>(q)], hipIpcMemLazyEnablePeerAccess));
            if (own2) CK(hipIpcOpenMemHandle(&mapped2[static_cast<size_t>(q)], all2[static_cast<size_t>(q)], hipIpcMemLazyEnablePeerAccess));
            const double dt = now_s() - t0;
            rep.open_sum_s += dt;
            if (dt > rep.open_max_s) rep.open_max_s = dt;
        }
        for (int step = 1; step < world; ++step)
        {
            const int q = (rank + step) % world;
            t0 = now_s();
            CK(hipMemcpyAsync(static_cast<char*>(mapped[static_cast<size_t>(q)]) + (bytes / world) * rank, static_cast<char*>(own) + (bytes / world) * rank, 4096, hipMemcpyDeviceToDevice, s));
            CK(hipStreamSynchronize(s));
            const double dt = now_s() - t0;
            if (dt > rep.first_touch_max_s) rep.first_touch_max_s = dt;
        }
        t0 = now_s();
        for (int step = 1; step < world; ++step)
        {
            const int q = (rank + step) % world;
            CK(hipMemcpyAsync(static_cast<char*>(mapped[static_cast<size_t>(q)]) + (bytes / world) * rank, static_cast<char*>(own) + (bytes / world) * rank, bytes / world, hipMemcpyDeviceToDevice, s));
        }
        CK(hipStreamSynchronize(s));
        rep.slab_copy_s = now_s() - t0;
    out:
        // every rank must be through with its peers' memory before anybody frees
        {
            char b = 1;
            if (!write_all(wfd, &b, 1) || !read_all(rfd, &b, 1)) return 4;
        }
        t0 = now_s();
        for (void* m : mapped)
            if (m) (void)hipIpcCloseMemHandle(m);
        for (void* m : mapped2)
            if (m) (void)hipIpcCloseMemHandle(m);
        if (own2) (void)hipFree(own2);
        rep.close_s = now_s() - t0;
        if (own) (void)hipFree(own);
        if (!write_all(wfd, &rep, sizeof rep)) return 4;
        char b;
        if (!read_all(rfd, &b, 1)) return 4;
    }
    return 0;
}

int main(int argc, char** argv)
{
    const double limit_s = argc > 1 ? std::atof(argv[1]) : 120.0;
    std::vector<size_t> sizes = {16ull << 20, 128ull << 20, 805ull << 20, 2048ull << 20, 6400ull << 20};
    if (argc > 2) sizes = {static_cast<size_t>(std::atoll(argv[2])) << 20};  // one size in MB (e.g. 8600: the depth-tile ring of C5 DDGI at 4 pairs)
    std::vector<int> worlds = {2, 4, 8};
    if (argc > 3) worlds = {std::atoi(argv[3])};
    std::printf("# hipIpcOpenMemHandle cost, W processes on one GPU, each opening the W-1 peers' buffers of B bytes (seconds; maximum over the ranks)\n");
    std::printf("# %5s %9s | %9s %9s | %12s %12s %12s | %12s %10s | %9s\n", "W", "B (MB)", "malloc", "export", "open (sum)", "open (max 1)", "s per GB", "first 4 KB", "slab copy", "close");
    for (int world : worlds)
    {
        std::vector<int> to_child(static_cast<size_t>(world)), from_child(static_cast<size_t>(world));
        std::vector<pid_t> pids(static_cast<size_t>(world));
        for (int r = 0; r < world; ++r)
        {
            int a[2], b[2];
            if (pipe(a) || pipe(b)) return 2;
            pid_t pid = fork();
            if (pid == 0)
            {
                close(a[1]), close(b[0]);
                _exit(child(r, world, a[0], b[1], sizes));
            }
            close(a[0]), close(b[1]);
            to_child[static_cast<size_t>(r)] = a[1], from_child[static_cast<size_t>(r)] = b[0];
            pids[static_cast<size_t>(r)] = pid;
        }
        bool dead = false;
        auto wait_all = [&](void* dst, size_t each, const char* what, size_t bytes) {
            const double t_end = now_s() + limit_s;
            for (int r = 0; r < world && !dead; ++r)
            {
                pollfd pf{from_child[static_cast<size_t>(r)], POLLIN, 0};
                const double left = t_end - now_s();
                if (left <= 0 || poll(&pf, 1, static_cast<int>(left * 1000)) <= 0 || !read_all(pf.fd, static_cast<char*>(dst) + each * static_cast<size_t>(r), each))
                {
                    std::printf("  %5d %9zu | rank %d did not report \"%s\" within %.0f s — run ended\n", world, bytes >> 20, r, what, limit_s);
                    dead = true;
                }
            }
        };
        for (size_t bytes : sizes)
        {
            if (dead) break;
            std::vector<hipIpcMemHandle_t> handles(static_cast<size_t>(world));
            wait_all(handles.data(), sizeof(hipIpcMemHandle_t), "exported", bytes);
            if (dead) break;
            for (int r = 0; r < world; ++r) write_all(to_child[static_cast<size_t>(r)], handles.data(), sizeof(hipIpcMemHandle_t) * handles.size());
            if (std::getenv("IPC_PAIR"))
            {
                wait_all(handles.data(), sizeof(hipIpcMemHandle_t), "exported the second buffer", bytes);
                if (dead) break;
                for (int r = 0; r < world; ++r) write_all(to_child[static_cast<size_t>(r)], handles.data(), sizeof(hipIpcMemHandle_t) * handles.size());
            }
            std::vector<char> tok(static_cast<size_t>(world));
            wait_all(tok.data(), 1, "opened + copied", bytes);
            if (dead) break;
            for (int r = 0; r < world; ++r) write_all(to_child[static_cast<size_t>(r)], tok.data(), 1);
            std::vector<Report> reps(static_cast<size_t>(world));
            wait_all(reps.data(), sizeof(Report), "closed", bytes);
            if (dead) break;
            for (int r = 0; r < world; ++r) write_all(to_child[static_cast<size_t>(r)], tok.data(), 1);
            Report m;
            std::memset(&m, 0, sizeof m);
            for (const Report& x : reps)
            {
                m.malloc_s = x.malloc_s > m.malloc_s ? x.malloc_s : m.malloc_s;
                m.export_s = x.export_s > m.export_s ? x.export_s : m.export_s;
                m.open_sum_s = x.open_sum_s > m.open_sum_s ? x.open_sum_s : m.open_sum_s;
                m.open_max_s = x.open_max_s > m.open_max_s ? x.open_max_s : m.open_max_s;
                m.first_touch_max_s = x.first_touch_max_s > m.first_touch_max_s ? x.first_touch_max_s : m.first_touch_max_s;
                m.slab_copy_s = x.slab_copy_s > m.slab_copy_s ? x.slab_copy_s : m.slab_copy_s;
                m.close_s = x.close_s > m.close_s ? x.close_s : m.close_s;
                m.rc |= x.rc;
            }
            const double gb = static_cast<double>(bytes) * (world - 1) / 1e9;
            std::printf("  %5d %9zu | %9.4f %9.4f | %12.4f %12.4f %12.4f | %12.4f %10.4f | %9.4f%s\n", world, bytes >> 20, m.malloc_s, m.export_s, m.open_sum_s, m.open_max_s, m.open_sum_s / gb,
                        m.first_touch_max_s, m.slab_copy_s, m.close_s, m.rc ? "  (a rank reported an error)" : "");
            std::fflush(stdout);
        }
        for (int r = 0; r < world; ++r)
        {
            close(to_child[static_cast<size_t>(r)]), close(from_child[static_cast<size_t>(r)]);
            if (dead) kill(pids[static_cast<size_t>(r)], SIGKILL);
            int st;
            waitpid(pids[static_cast<size_t>(r)], &st, 0);
        }
        if (dead) break;
    }
    return 0;
}
