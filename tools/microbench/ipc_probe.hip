// ipc_probe.hip — which cross-process primitives work on this box (two processes, ONE GPU)?  Decides what the
// peer-to-peer transport of ddgi_exchange.cpp may rely on.
//   hipcc --offload-arch=gfx950 -O2 ipc_probe.hip -o ipc_probe.bin && ./ipc_probe.bin
// parent: allocates a data buffer + a flag buffer, exports both (hipIpcGetMemHandle), waits on the flag with
//         hipStreamWaitValue32 (no CU is held), then checks the data the child pushed.
// child : opens both, hipMemcpyAsync's a pattern into the parent's buffer, then writes the flag
//         (A) with hipStreamWriteValue32, (B) with a 4-byte hipMemcpyAsync, (C) from a one-lane kernel.
// Also: hipIpcGetEventHandle / hipIpcOpenEventHandle (interprocess events).
#include <hip/hip_runtime.h>
#include <sys/wait.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x)                                                                                   \
    do                                                                                          \
    {                                                                                           \
        hipError_t e_ = (x);                                                                    \
        if (e_ != hipSuccess)                                                                   \
        {                                                                                       \
            std::printf("[%s] %s -> %s\n", who, #x, hipGetErrorString(e_));                     \
            std::fflush(stdout);                                                                \
            return 1;                                                                           \
        }                                                                                       \
    } while (0)

__global__ void k_set(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }

struct Msg
{
    hipIpcMemHandle_t data, flags;
    hipIpcEventHandle_t ev;
    int have_ev;
};

static const char* who = "?";
constexpr size_t kBytes = 8u << 20;

static int child(int rfd, int wfd)
{
    who = "child";
    Msg m;
    if (read(rfd, &m, sizeof m) != static_cast<ssize_t>(sizeof m)) return 2;
    CK(hipSetDevice(0));
    void *data = nullptr, *flags = nullptr;
    CK(hipIpcOpenMemHandle(&data, m.data, hipIpcMemLazyEnablePeerAccess));
    CK(hipIpcOpenMemHandle(&flags, m.flags, hipIpcMemLazyEnablePeerAccess));
    std::printf("[child] opened both handles\n");
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    uint32_t* src = nullptr;
    CK(hipMalloc(reinterpret_cast<void**>(&src), kBytes));
    uint32_t* seq = nullptr;
    CK(hipMalloc(reinterpret_cast<void**>(&seq), 64));
    std::vector<uint32_t> host(kBytes / 4);
    for (int round = 1; round <= 3; ++round)
    {
        for (size_t i = 0; i < host.size(); ++i) host[i] = static_cast<uint32_t>(i * 2654435761u) ^ static_cast<uint32_t>(round);
        CK(hipMemcpy(src, host.data(), kBytes, hipMemcpyHostToDevice));
        usleep(200000);  // the parent is certainly waiting by now
        CK(hipMemcpyAsync(data, src, kBytes, hipMemcpyDeviceToDevice, s));
        uint32_t* flag = static_cast<uint32_t*>(flags) + round;
        hipError_t e = hipSuccess;
        if (round == 1) e = hipStreamWriteValue32(s, flag, static_cast<uint32_t>(round), 0);
        if (round == 2)
        {
            const uint32_t v = 2;
            CK(hipMemcpy(seq, &v, 4, hipMemcpyHostToDevice));
            e = hipMemcpyAsync(flag, seq, 4, hipMemcpyDeviceToDevice, s);
        }
        if (round == 3)
        {
            hipLaunchKernelGGL(k_set, dim3(1), dim3(1), 0, s, flag, 3u);
            e = hipGetLastError();
        }
        std::printf("[child] round %d flag write issued: %s\n", round, hipGetErrorString(e));
        CK(hipStreamSynchronize(s));
        char ack;
        if (read(rfd, &ack, 1) != 1) return 3;
    }
    if (m.have_ev)
    {
        hipEvent_t ev;
        hipError_t e = hipIpcOpenEventHandle(&ev, m.ev);
        std::printf("[child] hipIpcOpenEventHandle: %s\n", hipGetErrorString(e));
        if (e == hipSuccess)
        {
            e = hipStreamWaitEvent(s, ev, 0);
            std::printf("[child] hipStreamWaitEvent(ipc event): %s\n", hipGetErrorString(e));
            e = hipStreamSynchronize(s);
            std::printf("[child] sync after waiting on the ipc event: %s\n", hipGetErrorString(e));
        }
    }
    char done = 'd';
    (void)!write(wfd, &done, 1);
    CK(hipIpcCloseMemHandle(data));
    CK(hipIpcCloseMemHandle(flags));
    return 0;
}

int main()
{
    int p2c[2], c2p[2];
    if (pipe(p2c) || pipe(c2p)) return 1;
    const pid_t pid = fork();  // before any HIP call
    if (pid == 0) return child(p2c[0], c2p[1]);
    who = "parent";
    CK(hipSetDevice(0));
    int can_wait = 0;
    CK(hipDeviceGetAttribute(&can_wait, hipDeviceAttributeCanUseStreamWaitValue, 0));
    std::printf("[parent] hipDeviceAttributeCanUseStreamWaitValue = %d\n", can_wait);
    uint32_t *data = nullptr, *flags = nullptr;
    CK(hipMalloc(reinterpret_cast<void**>(&data), kBytes));
    CK(hipMalloc(reinterpret_cast<void**>(&flags), 4096));
    CK(hipMemset(data, 0, kBytes));
    CK(hipMemset(flags, 0, 4096));
    Msg m{};
    CK(hipIpcGetMemHandle(&m.data, data));
    CK(hipIpcGetMemHandle(&m.flags, flags));
    hipEvent_t ev = nullptr;
    hipError_t ee = hipEventCreateWithFlags(&ev, hipEventDisableTiming | hipEventInterprocess);
    if (ee == hipSuccess) ee = hipIpcGetEventHandle(&m.ev, ev);
    m.have_ev = ee == hipSuccess;
    std::printf("[parent] interprocess event: %s\n", hipGetErrorString(ee));
    if (write(p2c[1], &m, sizeof m) != static_cast<ssize_t>(sizeof m)) return 1;
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    std::vector<uint32_t> host(kBytes / 4);
    for (int round = 1; round <= 3; ++round)
    {
        hipError_t e = hipStreamWaitValue32(s, flags + round, static_cast<uint32_t>(round), hipStreamWaitValueGte, 0xffffffffu);
        std::printf("[parent] round %d hipStreamWaitValue32 issued: %s\n", round, hipGetErrorString(e));
        CK(hipMemcpyAsync(host.data(), data, kBytes, hipMemcpyDeviceToHost, s));
        // bounded wait: never hang the box
        int ok = 0;
        for (int i = 0; i < 100; ++i)
        {
            if (hipStreamQuery(s) == hipSuccess) { ok = 1; break; }
            usleep(100000);
        }
        size_t bad = 0;
        for (size_t i = 0; i < host.size(); ++i) bad += host[i] != (static_cast<uint32_t>(i * 2654435761u) ^ static_cast<uint32_t>(round));
        std::printf("[parent] round %d: stream %s, %zu of %zu words wrong\n", round, ok ? "completed" : "STILL WAITING after 10 s", bad, host.size());
        if (!ok)
        {
            // release the wait ourselves so that the process can end
            uint32_t v = 100;
            (void)hipMemcpy(flags + round, &v, 4, hipMemcpyHostToDevice);
            (void)hipStreamSynchronize(s);
        }
        char ack = 'a';
        (void)!write(p2c[1], &ack, 1);
    }
    if (m.have_ev)
    {
        usleep(100000);
        CK(hipEventRecord(ev, s));
        CK(hipStreamSynchronize(s));
    }
    char done;
    (void)!read(c2p[0], &done, 1);
    int status = 0;
    waitpid(pid, &status, 0);
    std::printf("[parent] child exit status %d\n", WEXITSTATUS(status));
    return 0;
}
