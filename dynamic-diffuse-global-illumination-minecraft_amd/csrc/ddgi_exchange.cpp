// ddgi_exchange.cpp — the multi-GPU exchange behind the C ABI (new design: the reference is
// single-GPU and has no collective).
//
// The probe grid is sharded by z-slab (ddgi_create_sharded): rank r traces and blends the probes with
// z in [r*cz/G, (r+1)*cz/G).  The device textures are slab-major, so a rank's contribution is ONE contiguous,
// equal-sized chunk of each texture and the exchange is one in-place ncclAllGather per texture (RCCL over
// xGMI) — no packing kernels, no host staging.
//
// Pipelined mode keeps the exchange off the critical path: two texture pairs are used alternately; update k
// writes pair k&1 on the handle's stream while the all-gather of pair (k-1)&1 still runs on a communication
// stream.  REF mode rewrites its whole slab every update; the DDGI blend reads the previous tiles of its OWN
// slab from the other pair (BlendArgs::*_old) — a rank never needs another rank's tiles to update its own.
// Consumers (ddgi_sample*, ddgi_render*, ddgi_read_*) make the handle's stream wait for the latest pair.
//
// RCCL is resolved at run time (dlopen of the librccl.so.1 already in the process — e.g. the one PyTorch
// loaded — else the system's): a host that never shards needs no RCCL, and the communicator handed to
// ddgi_exchange_init and the collectives issued here always come from the same library instance.
//
// Two transports behind the same calls (ddgi_exchange / _finish, the engine's hooks):
//   RCCL  ddgi_exchange_init(h, comm, pipelined): one ncclAllGather per texture.  The default on a node.
//   P2P   ddgi_exchange_p2p_export + ddgi_exchange_p2p_init: SURVEY.md §8e's one-shot alternative — every rank
//         pushes its contiguous slab straight into every other rank's buffers (hipMemcpyAsync into memory mapped
//         with hipIpcOpenMemHandle; one process per rank), one stream per peer,
//         so all 7 xGMI links of a GPU carry one slab each at the same time instead of a ring's 7 sequential
//         hops.  Rendezvous is two flag words per peer pair in device memory, written by the peer and waited for
//         by the command processor (hipStreamWaitValue32: no CU is held, nothing spins):
//             ready[q]    rank q has finished reading what the pair held and may be written to  (receiver -> sender)
//             arrived[r]  rank r's slab of exchange number `seq` has landed                      (sender -> receiver)
//         It needs no RCCL, and — unlike RCCL, which refuses two ranks on one device — it runs with several
//         ranks on ONE GPU, which is how the one-GPU test box executes a rank > 0 at all.
//         A ring of 2 GiB or more is not published itself (such buffers do not reach another process reliably on the
//         stack measured): the peers push into LANDING ZONES — per texture and parity of the exchange's number one
//         buffer of world - 1 slabs — and a third phase on a stream of the receiver's own copies them into the pair
//         (ddgi_exchange_p2p_export, p2p_exchange).  The flag words live in fine-grained device memory.
#include <dlfcn.h>
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>

#include <chrono>
#include <condition_variable>
#include <memory>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <random>
#include <thread>
#include <vector>

#include "ddgi_engine.h"

namespace {

// Debugging aid (DDGI_DEBUG_BACKTRACE=1 in the environment when the library is loaded): SIGUSR2 sent to a THREAD (tgkill) prints that thread's native
// backtrace on stderr — where a driver call that does not come back stands, in a container without ptrace (tools/hunt/p2p_hang_hunt5.sh).
struct BacktraceOnSignal
{
    static void handler(int)
    {
        void* frames[48];
        const int n = backtrace(frames, 48);
        const char head[] = "[ddgi] native backtrace of the signalled thread:\n";
        (void)!write(2, head, sizeof head - 1);
        backtrace_symbols_fd(frames, n, 2);
    }
    BacktraceOnSignal()
    {
        if (!std::getenv("DDGI_DEBUG_BACKTRACE")) return;
        struct sigaction sa;
        std::memset(&sa, 0, sizeof sa);
        sa.sa_handler = handler;
        sigaction(SIGUSR2, &sa, nullptr);
    }
} g_backtrace_on_signal;

// the few RCCL entry points used, with the signatures of rccl.h (ROCm 7.2: rccl/rccl.h:220-236, 678)
struct NcclId  // ncclUniqueId (rccl.h:40-43): 128 opaque bytes, passed by value
{
    char b[128];
};
struct Rccl
{
    void* lib = nullptr;
    int (*GetUniqueId)(void* id128) = nullptr;
    int (*CommInitRank)(void** comm, int nranks, NcclId id, int rank) = nullptr;
    int (*CommInitAll)(void** comms, int ndev, const int* devlist) = nullptr;
    int (*CommDestroy)(void* comm) = nullptr;
    int (*CommAbort)(void* comm) = nullptr;
    int (*CommCount)(void* comm, int* count) = nullptr;
    int (*CommUserRank)(void* comm, int* rank) = nullptr;
    int (*AllGather)(const void* send, void* recv, size_t count, int dtype, void* comm, hipStream_t stream) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string error;
};
constexpr int kNcclUint8 = 1;  // ncclDataType_t (rccl.h:459-460)

Rccl& rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {"librccl.so.1", "librccl.so"};
        for (const char* n : names)
            if (!r.lib) r.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);  // the instance the process already holds
        for (const char* n : names)
            if (!r.lib) r.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (!r.lib)
        {
            r.error = std::string("librccl.so.1 could not be loaded: ") + (dlerror() ? dlerror() : "?");
            return;
        }
        auto sym = [&](const char* name) {
            void* p = dlsym(r.lib, name);
            if (!p && r.error.empty()) r.error = std::string("RCCL symbol missing: ") + name;
            return p;
        };
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
        r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(sym("ncclCommInitAll"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.CommAbort = reinterpret_cast<decltype(r.CommAbort)>(dlsym(r.lib, "ncclCommAbort"));  // (optional: only used to get a stream back from a collective a peer never joined)
        r.CommCount = reinterpret_cast<decltype(r.CommCount)>(sym("ncclCommCount"));
        r.CommUserRank = reinterpret_cast<decltype(r.CommUserRank)>(sym("ncclCommUserRank"));
        r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
        r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
        r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    });
    return r;
}

int rccl_ready()
{
    Rccl& r = rccl();
    if (!r.error.empty()) return fail(DDGI_ERR_UNSUPPORTED, "%s", r.error.c_str());
    return DDGI_OK;
}

#define NCCL_TRY(expr)                                                                                         \
    do                                                                                                         \
    {                                                                                                          \
        const int r_ = (expr);                                                                                 \
        if (r_ != 0) return fail(DDGI_ERR_HIP, "%s failed: %s", #expr, rccl().GetErrorString ? rccl().GetErrorString(r_) : "?"); \
    } while (0)


// ---- peer-to-peer transport ----------------------------------------------------------------------------------

constexpr uint32_t kP2PMagic = 0x32503244u;  // "D2P2"
constexpr int kP2PMaxWorld = 64;
#ifndef DDGI_P2P_FLAGS_ALLOC
#define DDGI_P2P_FLAGS_ALLOC (4u << 20)
#endif
constexpr size_t kP2PFlagsAlloc = DDGI_P2P_FLAGS_ALLOC;

// What a rank publishes (ddgi_exchange_p2p_export): where its buffers are, as IPC handles for its peers' processes.
// Fits DDGI_P2P_ADDRESS_BYTES.
struct P2PAddress
{
    uint32_t magic, rank, world, pipelined;
    int32_t pid, device;
    uint64_t tex_bytes[2];
    uint32_t np, landing;        // texture pairs in the rank's ring; landing != 0: the handles below are LANDING ZONES (ring = parity 0, ring_b = parity 1)
    uint64_t process;            // a random number drawn once per process: two ranks with the same one share a process (a pid would
                                 // not do: one container per rank with a shared IPC namespace makes every rank pid 1)
    hipIpcMemHandle_t ring[2];   // the two textures' rings (pair k of texture i at k * tex_bytes[i])
    hipIpcMemHandle_t flags;
    hipIpcMemHandle_t ring_b[2]; // landing zones of parity 1 (landing != 0 only)
};
static_assert(sizeof(P2PAddress) <= DDGI_P2P_ADDRESS_BYTES, "the published address must fit the ABI's blob");

uint64_t process_nonce()
{
    static const uint64_t nonce = [] {
        std::random_device rd;
        uint64_t v = (static_cast<uint64_t>(rd()) << 32) ^ rd();
        v ^= static_cast<uint64_t>(std::chrono::steady_clock::now().time_since_epoch().count()) * 0x9e3779b97f4a7c15ull;
        return v ? v : 1ull;
    }();
    return nonce;
}

}  // namespace

struct ddgi_engine::P2P
{
    struct Peer
    {
        void* ring[2] = {nullptr, nullptr};  // the peer's texture rings, mapped (landing zones: those of parity 0)
        void* ring_b[2] = {nullptr, nullptr};  // landing zones of parity 1
        uint32_t* flags = nullptr;
        bool ipc = false;  // mapped with hipIpcOpenMemHandle (to be closed)
        hipStream_t stream = nullptr;
        hipEvent_t done = nullptr;  // this peer's copy of the latest exchange has been issued and finished
    };
    uint32_t* flags = nullptr;   // own: [q] = ready, written by rank q; [kP2PMaxWorld + r] = arrived, written by rank r
    // when a wait for a peer runs into its deadline (ddgi_sync_stream): a stream of its own PRIORITY — HIP keeps a pool of hardware queues per
    // priority, so nothing it carries can stand behind a wait packet of the handle's normal-priority streams — reads the flag words into
    // pinned host memory (who is behind?) and then writes them itself (the handle's own waits end; the exchange is broken from then on)
    hipStream_t diag = nullptr;
    uint32_t* diag_host = nullptr;
    uint32_t waited_arrived = 0;  // the highest exchange number any consumer of this handle has been made to wait for
    std::vector<Peer> peers;     // [world]; the own rank's entry is unused
    uint32_t seq = 0;            // exchanges issued (the flags carry it and are compared with >=: good for 2^32 exchanges per attachment — 50 days at 1000 per second)
    uint32_t pair_seq[ddgi_engine::kMaxPairs] = {};  // exchange number that last filled pair i
    bool exported_pipelined = false;
    bool connected = false;
    bool write_value_ok = true;  // hipStreamWriteValue32 accepts peer memory (else a one-word fill)
    bool flags_fine = false;     // the own flag words live in fine-grained device memory (ddgi_exchange_p2p_export)
    // LANDING ZONES (a ring of 2 GiB or more, or DDGI_P2P_LANDING=1): what the peers map and push into is not the handle's ring but, per texture and
    // parity of the exchange's number, one buffer of world - 1 slabs (rank r's slab in slot r, the own rank's slot left out); this rank copies them
    // into the pair the exchange belongs to on a stream of its own (`gather`) once they have all arrived.  See p2p_exchange.
    bool landing = false;
    int land_ntex = 0;                                   // textures with zones (1: REF mode at export, 2: DDGI mode)
    void* land[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};  // [texture][parity]
    hipStream_t gather = nullptr;
    hipEvent_t gathered[ddgi_engine::kMaxPairs] = {};    // gather stream: the other ranks' slabs of pair i's last exchange are in the pair
    bool gathered_valid[ddgi_engine::kMaxPairs] = {};
    hipEvent_t drained[2] = {nullptr, nullptr};          // gather stream: the zones of this parity have been copied out (the peers may push the exchange after next)
    bool drained_valid[2] = {false, false};
};

namespace {

// The exchange's own streams — one per peer, and the one that collects them — are created at the LOWEST stream priority.  Not for the scheduling: HIP maps a
// process's streams onto a few hardware queues PER PRIORITY (GPU_MAX_HW_QUEUES, 4), and a hipStreamWaitValue32 packet holds its queue — whatever
// another stream has put behind it included.  At the handle's own priority a peer stream that stands at a slow peer's `ready` can hold the queue the
// handle's next update is in (round 6's lost-rank test showed the effect between two peer streams: the push to a LIVE peer stood behind the wait for the
// dead one); in a pool of their own the exchange's waits can only hold each other.
hipError_t create_exchange_stream(hipStream_t* s)
{
    int least = 0, greatest = 0;  // (numerically higher = lower priority)
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) least = 0;
    hipError_t he = hipStreamCreateWithPriority(s, hipStreamNonBlocking, least);
    if (he != hipSuccess)
    {
        (void)hipGetLastError();
        he = hipStreamCreateWithFlags(s, hipStreamNonBlocking);
    }
    return he;
}

// one flag word of a peer := v, in stream order
int p2p_write_flag(ddgi_engine::P2P& p, hipStream_t s, uint32_t* flag, uint32_t v)
{
    if (p.write_value_ok)
    {
        if (hipStreamWriteValue32(s, flag, v, 0) == hipSuccess) return DDGI_OK;
        (void)hipGetLastError();
        p.write_value_ok = false;
    }
    HIP_TRY(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(flag), static_cast<int>(v), 1, s));
    return DDGI_OK;
}

int p2p_wait_flag(hipStream_t s, uint32_t* flag, uint32_t v)
{
    HIP_TRY(hipStreamWaitValue32(s, flag, v, hipStreamWaitValueGte, 0xffffffffu));
    return DDGI_OK;
}

// hipStreamQuery / hipEventQuery until done or `ms` have passed (hipErrorNotReady).  The first polls spin (a wait that is about to end costs
// what hipStreamSynchronize costs), later ones sleep 50 us.
template <class Query>
hipError_t poll_until(Query query, int ms)
{
    const auto t_end = std::chrono::steady_clock::now() + std::chrono::milliseconds(ms);
    for (unsigned polls = 0;; ++polls)
    {
        const hipError_t he = query();
        if (he != hipErrorNotReady) return he;
        (void)hipGetLastError();
        if (std::chrono::steady_clock::now() >= t_end) return hipErrorNotReady;
        if (polls < 4000u)
            std::this_thread::yield();
        else
            std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
}

// Ends every wait this rank's own streams stand at: all of them are waits on THIS rank's flag words (a rank waits on its own memory and
// writes into its peers').  True when the write is known to have landed.
bool p2p_force_own_flags(ddgi_engine::P2P& p)
{
    if (!p.flags || !p.diag) return false;
    if (hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(p.flags), static_cast<int>(0xffffffffu), 2 * kP2PMaxWorld, p.diag) != hipSuccess) return false;
    return poll_until([&] { return hipStreamQuery(p.diag); }, 2000) == hipSuccess;
}

void p2p_release(ddgi_engine* e)
{
    ddgi_engine::P2P* p = e->xch.p2p;
    if (!p) return;
    // a peer that is gone leaves this rank's copy streams at a wait for its `ready`: bounded, then released by this rank itself
    const int limit = e->tuning.wait_timeout_ms;
    bool forced = false;
    for (auto& peer : p->peers)
    {
        if (!peer.stream) continue;
        if (limit <= 0)
        {
            (void)hipStreamSynchronize(peer.stream);
            continue;
        }
        if (poll_until([&] { return hipStreamQuery(peer.stream); }, forced ? 2000 : limit) == hipErrorNotReady && !forced)
        {
            forced = true;
            (void)p2p_force_own_flags(*p);
            (void)poll_until([&] { return hipStreamQuery(peer.stream); }, 2000);
        }
    }
    if (p->gather)
    {
        if (limit <= 0) (void)hipStreamSynchronize(p->gather);
        else if (poll_until([&] { return hipStreamQuery(p->gather); }, forced ? 2000 : limit) == hipErrorNotReady && !forced)
        {
            forced = true;
            (void)p2p_force_own_flags(*p);
            (void)poll_until([&] { return hipStreamQuery(p->gather); }, 2000);
        }
        (void)hipStreamDestroy(p->gather);
    }
    for (auto& ev : p->gathered)
        if (ev) (void)hipEventDestroy(ev);
    for (auto& ev : p->drained)
        if (ev) (void)hipEventDestroy(ev);
    if (p->diag) (void)hipStreamDestroy(p->diag);
    if (p->diag_host) (void)hipHostFree(p->diag_host);
    for (auto& peer : p->peers)
    {
        if (peer.ipc)
        {
            for (void* q : peer.ring)
                if (q) (void)hipIpcCloseMemHandle(q);
            for (void* q : peer.ring_b)
                if (q) (void)hipIpcCloseMemHandle(q);
            if (peer.flags) (void)hipIpcCloseMemHandle(peer.flags);
        }
        if (peer.done) (void)hipEventDestroy(peer.done);
        if (peer.stream) (void)hipStreamDestroy(peer.stream);
    }
    if (p->flags) (void)hipFree(p->flags);
    for (auto& tex : p->land)
        for (void* zone : tex)
            if (zone) (void)hipFree(zone);
    delete p;
    e->xch.p2p = nullptr;
}

// The consumers of exchange number `seq` may run once every other rank's slab of it has landed.
int p2p_wait_arrived(ddgi_engine* e, uint32_t seq)
{
    ddgi_engine::P2P& p = *e->xch.p2p;
    if (seq == 0) return DDGI_OK;
    if (seq > p.waited_arrived) p.waited_arrived = seq;
    if (p.landing)
    {
        // the flag words are waited for by the gather stream (p2p_exchange); a consumer waits for the copies out of the landing zones.  Every pair's latest:
        // those of older exchanges have long fired, and an event is only recorded behind the waits for ITS exchange's slabs.
        for (int k = 0; k < ddgi_engine::kMaxPairs; ++k)
            if (p.gathered_valid[k] && p.pair_seq[k] <= seq) HIP_TRY(hipStreamWaitEvent(e->stream, p.gathered[k], 0));
        return DDGI_OK;
    }
    for (int r = 0; r < e->world; ++r)
        if (r != e->rank)
            if (int rc = p2p_wait_flag(e->stream, p.flags + kP2PMaxWorld + r, seq)) return rc;
    return DDGI_OK;
}

// The event the exchange's streams wait for: "this rank's slab is written".  Normally the last update's own end event — recorded
// for ddgi_last_update_ms anyway (an event costs the stream microseconds, and a slab's update is short) —, a fresh one if the
// update was not timed or if another call may have put work that touches the textures on the stream since.
static int update_written_event(ddgi_engine* e, hipEvent_t* out)
{
    ddgi_engine::Exchange& x = e->xch;
    if (!e->tex_ops_since_update && e->updates > 0)
    {
        const size_t slot = (e->updates - 1) % ddgi_engine::kRing;
        if (e->ev_valid[slot])
        {
            *out = e->ev[slot][e->ev_has_blend[slot] ? 2 : 1];
            return DDGI_OK;
        }
    }
    HIP_TRY(hipEventRecord(x.written, e->stream));
    *out = x.written;
    return DDGI_OK;
}

int p2p_exchange(ddgi_engine* e)
{
    ddgi_engine::Exchange& x = e->xch;
    ddgi_engine::P2P& p = *x.p2p;
    const uint32_t seq = ++p.seq;
    const int cur = e->pair_cur;
    p.pair_seq[cur] = seq;
    hipEvent_t written = nullptr;
    if (int rc = update_written_event(e, &written)) return rc;
    // REF mode: the reference never assigns its `distances` image — every rank's copy is all zeros already
    const int n_tex = e->mode == DDGI_MODE_DDGI ? 2 : 1;
    if (p.landing && n_tex > p.land_ntex)
        return fail(DDGI_ERR_NOT_READY, "the handle's mode changed after ddgi_exchange_p2p_export: its landing zones hold %d texture(s), the mode exchanges %d — attach the exchange again on every rank", p.land_ntex, n_tex);
    // Phase 1, receiver -> every sender: everything this rank enqueued that reads the pair is done (`written` follows it in
    // stream order), the peers may write their slabs of exchange `seq` into it.  ALL of these go out before the first wait
    // below: HIP may map several streams onto one hardware queue, where a waiting packet holds back whatever is queued
    // behind it — with every rank's "ready" already on its way, no chain of ranks waiting for each other can close.
    for (int step = 1; step < e->world; ++step)
    {
        const int q = (e->rank + step) % e->world;  // every rank starts with a different peer
        ddgi_engine::P2P::Peer& peer = p.peers[static_cast<size_t>(q)];
        HIP_TRY(hipStreamWaitEvent(peer.stream, written, 0));
        // (landing zones: what the peers write is the zone of this exchange's parity — free once the exchange before last has been copied out of it)
        if (p.landing && p.drained_valid[seq & 1u]) HIP_TRY(hipStreamWaitEvent(peer.stream, p.drained[seq & 1u], 0));
        if (int rc = p2p_write_flag(p, peer.stream, peer.flags + e->rank, seq)) return rc;
    }
    // Phase 2, sender: once q is ready, push the slab and tell q that it has landed
    for (int step = 1; step < e->world; ++step)
    {
        const int q = (e->rank + step) % e->world;
        ddgi_engine::P2P::Peer& peer = p.peers[static_cast<size_t>(q)];
        if (int rc = p2p_wait_flag(peer.stream, p.flags + q, seq)) return rc;
        for (int i = 0; i < n_tex; ++i)
        {
            const size_t slab = e->tex_bytes[i] / static_cast<size_t>(e->world);
            const size_t off = slab * static_cast<size_t>(e->rank);
            // the pair in the peer's ring — or, landing zones, this rank's slot in the peer's zone of the exchange's parity (q's own slot is left out)
            uint8_t* dst = !p.landing ? static_cast<uint8_t*>(peer.ring[i]) + static_cast<size_t>(cur) * e->tex_bytes[i] + off
                                      : static_cast<uint8_t*>((seq & 1u) ? peer.ring_b[i] : peer.ring[i]) + slab * static_cast<size_t>(e->rank < q ? e->rank : e->rank - 1);
            HIP_TRY(hipMemcpyAsync(dst, static_cast<const uint8_t*>(e->tex[i]) + off, slab, hipMemcpyDeviceToDevice, peer.stream));
        }
        if (int rc = p2p_write_flag(p, peer.stream, peer.flags + kP2PMaxWorld + e->rank, seq)) return rc;
        HIP_TRY(hipEventRecord(peer.done, peer.stream));
        HIP_TRY(hipStreamWaitEvent(x.comm_stream, peer.done, 0));
    }
    HIP_TRY(hipEventRecord(x.sent[cur], x.comm_stream));  // this rank's slab has left: the update after next may overwrite it
    x.sent_valid[cur] = true;
    if (p.landing)
    {
        // Phase 3, receiver (landing zones only): once every other rank's slab of exchange `seq` has landed in this parity's zones, copy them into the pair
        // the exchange belongs to — slots [0, rank) are slabs [0, rank), slots [rank, world - 1) are slabs (rank, world) — on a stream of its own (the
        // waits would hold `sent` back on the communication stream).  Nothing this rank enqueued before `written` still reads the pair's other slabs.
        HIP_TRY(hipStreamWaitEvent(p.gather, written, 0));
        for (int r = 0; r < e->world; ++r)
            if (r != e->rank)
                if (int rc = p2p_wait_flag(p.gather, p.flags + kP2PMaxWorld + r, seq)) return rc;
        for (int i = 0; i < n_tex; ++i)
        {
            const size_t slab = e->tex_bytes[i] / static_cast<size_t>(e->world);
            const uint8_t* zone = static_cast<const uint8_t*>(p.land[i][seq & 1u]);
            uint8_t* pair = static_cast<uint8_t*>(e->tex[i]);
            const size_t below = static_cast<size_t>(e->rank), above = static_cast<size_t>(e->world - 1 - e->rank);
            if (below) HIP_TRY(hipMemcpyAsync(pair, zone, slab * below, hipMemcpyDeviceToDevice, p.gather));
            if (above) HIP_TRY(hipMemcpyAsync(pair + slab * (below + 1), zone + slab * below, slab * above, hipMemcpyDeviceToDevice, p.gather));
        }
        HIP_TRY(hipEventRecord(p.gathered[cur], p.gather));
        p.gathered_valid[cur] = true;
        HIP_TRY(hipEventRecord(p.drained[seq & 1u], p.gather));
        p.drained_valid[seq & 1u] = true;
    }
    if (!x.pipelined)
    {
        // in order: what follows on the handle's stream sees the whole field, and does not touch the slab before it has left
        HIP_TRY(hipStreamWaitEvent(e->stream, x.sent[cur], 0));
        if (int rc = p2p_wait_arrived(e, seq)) return rc;
    }
    return DDGI_OK;
}

}  // namespace

// ---- host waits with a deadline ---------------------------------------------------------------------------------
//
// The reference's fences give up after DEFAULT_FENCE_TIMEOUT = 1 s (src/rvpt/vk_util.cpp:65, 94-97).  A handle with an exchange attached
// waits for other ranks: its streams stand at hipStreamWaitValue32 packets (peer-to-peer) or inside a collective (RCCL) that only a
// live peer ends.  Round 5's driver run showed what the missing deadline costs (GPUTEST_r05: one rank silent for 180 s, 79 tests lost).

namespace {

int exchange_timed_out(ddgi_engine* e, const char* waited_for)
{
    ddgi_engine::Exchange& x = e->xch;
    const int limit = e->tuning.wait_timeout_ms;
    x.broken = true;
    e->chain_break = true;
    if (x.transport == DDGI_EXCHANGE_P2P || x.p2p)
    {
        ddgi_engine::P2P& p = *x.p2p;
        char who[384] = "the flag words could not be read";
        bool have = false;
        if (p.diag && p.diag_host && p.flags &&
            hipMemcpyAsync(p.diag_host, p.flags, 2 * kP2PMaxWorld * sizeof(uint32_t), hipMemcpyDeviceToHost, p.diag) == hipSuccess)
            have = poll_until([&] { return hipStreamQuery(p.diag); }, 2000) == hipSuccess;
        if (have)
        {
            // Every peer that is behind, the one furthest behind first.  A peer that is gone is behind in BOTH flags; a live peer whose slab has
            // not arrived may itself be waiting for the one that is gone (its pushes share hardware queues with its waits) — so the order is by
            // `ready` (what it releases as soon as its own update is done), then by `arrived`.
            const uint32_t want_ready = p.seq, want_arrived = p.waited_arrived;
            int order[kP2PMaxWorld], n_lag = 0;
            for (int q = 0; q < e->world && q < kP2PMaxWorld; ++q)
                if (q != e->rank && (p.diag_host[q] < want_ready || p.diag_host[kP2PMaxWorld + q] < want_arrived)) order[n_lag++] = q;
            for (int a = 1; a < n_lag; ++a)
                for (int b = a; b > 0; --b)
                {
                    const int x = order[b], y = order[b - 1];
                    const bool less = p.diag_host[x] != p.diag_host[y] ? p.diag_host[x] < p.diag_host[y] : p.diag_host[kP2PMaxWorld + x] < p.diag_host[kP2PMaxWorld + y];
                    if (!less) break;
                    order[b] = y, order[b - 1] = x;
                }
            if (n_lag > 0)
            {
                int at = std::snprintf(who, sizeof who, "rank %d is behind: `ready` at exchange %u, `arrived` at exchange %u; this rank (%d of %d) waits for exchange %u / %u", order[0], p.diag_host[order[0]],
                                       p.diag_host[kP2PMaxWorld + order[0]], e->rank, e->world, want_ready, want_arrived);
                for (int a = 1; a < n_lag && at > 0 && at < static_cast<int>(sizeof who) - 48; ++a)
                    at += std::snprintf(who + at, sizeof who - static_cast<size_t>(at), "%s rank %d (%u / %u)", a == 1 ? "; also behind:" : ",", order[a], p.diag_host[order[a]], p.diag_host[kP2PMaxWorld + order[a]]);
            }
            else
                std::snprintf(who, sizeof who, "every peer's flags are up to date (exchange %u / %u): the wait is not for a peer", want_ready, want_arrived);
        }
        // end this rank's own waits: the streams drain, the handle can be detached / reconfigured / destroyed
        const bool forced = p2p_force_own_flags(p);
        bool drained = forced;
        if (forced)
        {
            for (auto& peer : p.peers)
                if (peer.stream) drained = drained && poll_until([&] { return hipStreamQuery(peer.stream); }, 2000) == hipSuccess;
            if (x.comm_stream) drained = drained && poll_until([&] { return hipStreamQuery(x.comm_stream); }, 2000) == hipSuccess;
            if (p.gather) drained = drained && poll_until([&] { return hipStreamQuery(p.gather); }, 2000) == hipSuccess;
            drained = drained && poll_until([&] { return hipStreamQuery(e->stream); }, 2000) == hipSuccess;
        }
        return fail(DDGI_ERR_TIMEOUT, "%s did not end within %d ms (tuning \"wait_timeout_ms\"): %s.  The exchange is broken — attach it again on every rank; %s", waited_for, limit, who,
                    drained ? "this rank's own waits were released and its streams have drained" : "this rank's streams could NOT be drained");
    }
    return fail(DDGI_ERR_TIMEOUT, "%s did not end within %d ms (tuning \"wait_timeout_ms\"): an RCCL all-gather of this handle (rank %d of %d) has not completed — a peer rank is not taking part.  The exchange is broken: "
                                  "abort the communicator (ncclCommAbort) and attach a new one on every rank",
                waited_for, limit, e->rank, e->world);
}

}  // namespace

int ddgi_sync_stream(ddgi_engine* e, hipStream_t s)
{
    const int limit = e->tuning.wait_timeout_ms;
    if (limit <= 0 || (!e->xch.transport && !e->xch.p2p))
    {
        HIP_TRY(hipStreamSynchronize(s));
        return DDGI_OK;
    }
    const hipError_t he = poll_until([&] { return hipStreamQuery(s); }, limit);
    if (he == hipSuccess) return DDGI_OK;
    if (he != hipErrorNotReady) return fail(DDGI_ERR_HIP, "hipStreamQuery failed: %s", hipGetErrorString(he));
    return exchange_timed_out(e, "a wait for the handle's stream");
}

int ddgi_sync_event(ddgi_engine* e, hipEvent_t ev)
{
    const int limit = e->tuning.wait_timeout_ms;
    if (limit <= 0 || (!e->xch.transport && !e->xch.p2p))
    {
        HIP_TRY(hipEventSynchronize(ev));
        return DDGI_OK;
    }
    const hipError_t he = poll_until([&] { return hipEventQuery(ev); }, limit);
    if (he == hipSuccess) return DDGI_OK;
    if (he != hipErrorNotReady) return fail(DDGI_ERR_HIP, "hipEventQuery failed: %s", hipGetErrorString(he));
    return exchange_timed_out(e, "a wait for an event of the handle's stream");
}

// ---- hooks for the engine -------------------------------------------------------------------------------

namespace {
// One process driving several handles brackets a frame's ddgi_exchange calls with ddgi_exchange_group_begin/end
// (≙ ncclGroupStart/End).  Inside the bracket RCCL only RECORDS the collectives; they reach their streams at the
// outermost ncclGroupEnd — so "this exchange is over" (Exchange::sent) can only be recorded there, not in
// ddgi_exchange.  The bracket is per thread, like RCCL's.
struct PendingSent
{
    ddgi_engine* e;
    int pair;
    std::thread::id owner;  // the thread whose bracket the exchange was recorded in: only ITS ncclGroupEnd puts the collective on its stream
};
thread_local int g_group_depth = 0;
// (one list for the process, under a lock: a handle destroyed from another thread than the one that exchanged must still be
// purged from it — ddgi_exchange_release)
std::mutex g_group_mu;
std::vector<PendingSent> g_group_pending;
}  // namespace

// Pipelined exchange: pairs [first, first + n) of the ring are about to be written (an update's launch, and the updates it may
// continue into): the stream waits until their previous exchanges have left the buffers.
int ddgi_exchange_before_update(ddgi_engine* e, int first_pair, int n_pairs)
{
    ddgi_engine::Exchange& x = e->xch;
    if (!x.transport || !x.pipelined) return DDGI_OK;
    for (int k = first_pair; k < first_pair + n_pairs && k < e->np; ++k)
        if (x.sent_valid[k]) HIP_TRY(hipStreamWaitEvent(e->stream, x.sent[k], 0));
    return DDGI_OK;
}

int ddgi_exchange_wait_latest(ddgi_engine* e)
{
    ddgi_engine::Exchange& x = e->xch;
    if (x.broken)
        return fail(DDGI_ERR_TIMEOUT, "the multi-GPU exchange of this handle is broken (a wait for another rank ran into \"wait_timeout_ms\" earlier): the other ranks' slabs are not valid — attach the "
                                      "exchange again on every rank, or detach it (ddgi_exchange_init(h, NULL, 0))");
    if (!x.transport || !x.pipelined) return DDGI_OK;  // in-order exchange: stream order already covers it
    if (x.transport == DDGI_EXCHANGE_P2P) return p2p_wait_arrived(e, x.p2p->pair_seq[e->pair_cur]);
    if (x.sent_valid[e->pair_cur]) HIP_TRY(hipStreamWaitEvent(e->stream, x.sent[e->pair_cur], 0));
    return DDGI_OK;
}

void ddgi_exchange_p2p_info(const ddgi_engine* e, int* landing_textures, int* exported_mb)
{
    *landing_textures = 0, *exported_mb = 0;
    const ddgi_engine::P2P* p = e->xch.p2p;
    if (!p) return;
    *landing_textures = p->landing ? p->land_ntex : 0;
    double bytes = static_cast<double>(kP2PFlagsAlloc);
    if (p->landing)
        for (int i = 0; i < p->land_ntex; ++i) bytes += 2.0 * static_cast<double>(e->tex_bytes[i]) / e->world * (e->world - 1);
    else
        for (int i = 0; i < 2; ++i) bytes += static_cast<double>(e->tex_bytes[i]) * e->np;
    *exported_mb = static_cast<int>(bytes / 1048576.0 + 0.5);
}

void ddgi_exchange_release(ddgi_engine* e)
{
    ddgi_engine::Exchange& x = e->xch;
    e->box_of = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_group_mu);
        for (auto it = g_group_pending.begin(); it != g_group_pending.end();)
            it = it->e == e ? g_group_pending.erase(it) : it + 1;
    }
    e->chain_break = true;
    p2p_release(e);  // (first: it ends this rank's waits for peers that are gone)
    if (x.comm_stream)
    {
        if (e->tuning.wait_timeout_ms <= 0) (void)hipStreamSynchronize(x.comm_stream);
        else if (poll_until([&] { return hipStreamQuery(x.comm_stream); }, e->tuning.wait_timeout_ms) == hipErrorNotReady && x.transport == DDGI_EXCHANGE_RCCL && x.comm && rccl().CommAbort)
        {
            // an all-gather that a peer never joined: the only way to get the stream back is to abort the communicator's kernels (the caller
            // still owns the ncclComm_t and may only ncclCommDestroy it from here on)
            (void)rccl().CommAbort(x.comm);
            (void)poll_until([&] { return hipStreamQuery(x.comm_stream); }, 2000);
        }
    }
    // (the ring keeps the pairs the pipelined exchange asked for: the handle goes on alternating them, which costs memory only;
    // the next configuration or "frames_in_flight" change sizes it anew)
    if (x.comm_stream) (void)hipStreamDestroy(x.comm_stream);
    if (x.written) (void)hipEventDestroy(x.written);
    for (auto& ev : x.sent)
        if (ev) (void)hipEventDestroy(ev);
    x = ddgi_engine::Exchange{};
}

namespace {
// Streams, events and (pipelined) the second texture pair, common to both transports.  On failure everything is released.
int exchange_common_setup(ddgi_engine* e, bool pipelined, bool always_streams)
{
    ddgi_engine::Exchange& x = e->xch;
    // Which pair of the ring an update writes — and a peer's push lands in — is a function of the update's NUMBER since the ring
    // was made (ddgi_engine::ring_k).  The ranks attach their exchange together but may have done different numbers of updates
    // before (asymmetric warm-up, one rank's failed update): every attachment starts the count over, on pair 0, with the
    // current contents.
    if (!e->caller_tex)
        if (int rc = ddgi_rebase_ring(e)) return rc;
    x.desync = false;
    if (pipelined)
    {
        if (e->caller_tex) return fail(DDGI_ERR_INVALID_ARGUMENT, "the pipelined exchange alternates the handle's own texture pairs: unbind caller textures first");
        // twice the pairs one launch may write (at least two): the exchanges of one group of updates run while the next group's
        // pairs are written.  The tiles so far move to pair 0 — what the next update's blend mixes with, what consumers read.
        e->pin_pair = false;
        if (int rc = ddgi_resize_ring(e, ddgi_pairs_wanted(e, true))) return rc;
        x.pipelined = true;
    }
    if (pipelined || always_streams)
    {
        // (peer-to-peer: the stream only collects the peer streams' events — in their pool; RCCL: it carries the all-gather's KERNEL, at the handle's priority)
        hipError_t he = always_streams ? create_exchange_stream(&x.comm_stream) : hipStreamCreateWithFlags(&x.comm_stream, hipStreamNonBlocking);
        if (he == hipSuccess) he = hipEventCreateWithFlags(&x.written, hipEventDisableTiming);
        for (int i = 0; i < ddgi_engine::kMaxPairs && he == hipSuccess; ++i) he = hipEventCreateWithFlags(&x.sent[i], hipEventDisableTiming);
        if (he != hipSuccess)
        {
            const int rc = fail(DDGI_ERR_HIP, "exchange stream/event creation failed: %s", hipGetErrorString(he));
            ddgi_exchange_release(e);
            return rc;
        }
    }
    return DDGI_OK;
}
}  // namespace

// ---- C ABI -------------------------------------------------------------------------------------------------

extern "C" {

int ddgi_comm_unique_id(uint8_t id128[128])
{
    if (!id128) return fail(DDGI_ERR_INVALID_ARGUMENT, "null id");
    if (int rc = rccl_ready()) return rc;
    NCCL_TRY(rccl().GetUniqueId(id128));
    return DDGI_OK;
}

int ddgi_comm_create(const uint8_t id128[128], int world, int rank, int device, void** comm)
{
    if (!id128 || !comm) return fail(DDGI_ERR_INVALID_ARGUMENT, "null id/comm");
    *comm = nullptr;
    if (world < 1 || rank < 0 || rank >= world) return fail(DDGI_ERR_INVALID_ARGUMENT, "rank %d not in [0,%d)", rank, world);
    if (int rc = rccl_ready()) return rc;
    HIP_TRY(hipSetDevice(device));
    NcclId id;
    std::memcpy(id.b, id128, 128);
    NCCL_TRY(rccl().CommInitRank(comm, world, id, rank));
    return DDGI_OK;
}

int ddgi_comm_create_all(int ndev, const int* devices, void** comms)
{
    if (ndev < 1 || !comms) return fail(DDGI_ERR_INVALID_ARGUMENT, "bad device list");
    if (int rc = rccl_ready()) return rc;
    NCCL_TRY(rccl().CommInitAll(comms, ndev, devices));
    return DDGI_OK;
}

int ddgi_comm_destroy(void* comm)
{
    if (!comm) return DDGI_OK;
    if (int rc = rccl_ready()) return rc;
    NCCL_TRY(rccl().CommDestroy(comm));
    return DDGI_OK;
}

int ddgi_exchange_group_begin(void)
{
    if (int rc = rccl_ready()) return rc;
    NCCL_TRY(rccl().GroupStart());
    g_group_depth += 1;
    return DDGI_OK;
}

int ddgi_exchange_group_end(void)
{
    if (int rc = rccl_ready()) return rc;
    if (g_group_depth > 0) g_group_depth -= 1;
    NCCL_TRY(rccl().GroupEnd());
    if (g_group_depth == 0)
    {
        // the collectives recorded inside the bracket are on their streams NOW: this is where "exchange over" is in stream order
        // (every pending handle gets its event, whatever happens to another one's: a pair left without it would be overwritten by
        // the update after next while its all-gather may still be reading it; the first error is what the call returns)
        // (only this thread's: another thread's bracket may still be open — its all-gathers are not on their streams yet)
        std::vector<PendingSent> pending;
        {
            std::lock_guard<std::mutex> lock(g_group_mu);
            const std::thread::id me = std::this_thread::get_id();
            for (auto it = g_group_pending.begin(); it != g_group_pending.end();)
                if (it->owner == me)
                {
                    pending.push_back(*it);
                    it = g_group_pending.erase(it);
                }
                else
                    ++it;
        }
        int first_rc = DDGI_OK;
        for (const PendingSent& ps : pending)
        {
            ddgi_engine::Exchange& x = ps.e->xch;
            if (x.transport != DDGI_EXCHANGE_RCCL || !x.pipelined) continue;
            hipError_t he = hipSetDevice(ps.e->device);
            if (he == hipSuccess) he = hipEventRecord(x.sent[ps.pair], x.comm_stream);
            if (he == hipSuccess) x.sent_valid[ps.pair] = true;
            else if (first_rc == DDGI_OK) first_rc = fail(DDGI_ERR_HIP, "recording the end of an exchange failed: %s", hipGetErrorString(he));
        }
        return first_rc;
    }
    return DDGI_OK;
}

int ddgi_exchange_init(ddgi_handle e, void* nccl_comm, int pipelined)
{
    if (!e) return fail(DDGI_ERR_INVALID_ARGUMENT, "null handle");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipStreamSynchronize(e->stream));
    ddgi_exchange_release(e);
    if (!nccl_comm) return DDGI_OK;  // exchange switched off
    if (int rc = rccl_ready()) return rc;
    int count = 0, rank = -1;
    NCCL_TRY(rccl().CommCount(nccl_comm, &count));
    NCCL_TRY(rccl().CommUserRank(nccl_comm, &rank));
    if (count != e->world || rank != e->rank)
        return fail(DDGI_ERR_INVALID_ARGUMENT, "communicator is rank %d of %d, the handle is slab %d of %d", rank, count, e->rank, e->world);
    if (int rc = exchange_common_setup(e, pipelined != 0, false)) return rc;
    e->xch.comm = nccl_comm;
    e->xch.transport = DDGI_EXCHANGE_RCCL;
    return DDGI_OK;
}

int ddgi_exchange_p2p_export(ddgi_handle e, int pipelined, uint8_t address[DDGI_P2P_ADDRESS_BYTES])
{
    if (!e || !address) return fail(DDGI_ERR_INVALID_ARGUMENT, "null handle/address");
    if (e->world > kP2PMaxWorld) return fail(DDGI_ERR_UNSUPPORTED, "the peer-to-peer exchange serves at most %d ranks", kP2PMaxWorld);
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipStreamSynchronize(e->stream));
    ddgi_exchange_release(e);
    if (e->caller_tex) return fail(DDGI_ERR_INVALID_ARGUMENT, "the peer-to-peer exchange publishes the handle's own textures: unbind caller textures first");
    if (int rc = exchange_common_setup(e, pipelined != 0, true)) return rc;
    ddgi_engine::Exchange& x = e->xch;
    e->pin_pair = false;  // (the peers write into whichever pair of the ring an update has just written)
    x.p2p = new (std::nothrow) ddgi_engine::P2P();
    if (!x.p2p)
    {
        ddgi_exchange_release(e);
        return fail(DDGI_ERR_OUT_OF_MEMORY, "host allocation failed");
    }
    ddgi_engine::P2P& p = *x.p2p;
    p.exported_pipelined = x.pipelined;
    P2PAddress a;
    std::memset(&a, 0, sizeof a);
    a.magic = kP2PMagic, a.rank = static_cast<uint32_t>(e->rank), a.world = static_cast<uint32_t>(e->world), a.pipelined = x.pipelined ? 1u : 0u;
    a.pid = static_cast<int32_t>(getpid()), a.device = e->device;
    a.process = process_nonce();
    a.np = static_cast<uint32_t>(e->np);
    // The flag words are 512 bytes — allocated as kP2PFlagsAlloc: ROCr serves small device allocations from 2 MB blocks it carves up itself
    // ("fragments"), and exporting a fragment exports its block, shared with whatever else the process keeps there.  A block of its own
    // keeps the peers' mappings of this handle's flags apart from every other allocation's life cycle (round 6, docs/LAB_NOTES.md).
    // FINE-GRAINED device memory where the runtime offers it: the words are written by ANOTHER GPU over xGMI and polled by this GPU's command
    // processor (hipStreamWaitValue32) — coarse-grained memory is only guaranteed coherent at kernel boundaries of its own device, which a
    // wait packet is not.  (On the one-GPU test box both kinds behave alike; RCCL keeps its own cross-GPU flags in fine-grained memory too.)
    // DDGI_P2P_COARSE_FLAGS=1 in the environment keeps plain hipMalloc (A/B).
    hipError_t he = hipErrorUnknown;
    const char* coarse = std::getenv("DDGI_P2P_COARSE_FLAGS");
    if (!coarse || coarse[0] != '1')
    {
        he = hipExtMallocWithFlags(reinterpret_cast<void**>(&p.flags), kP2PFlagsAlloc, hipDeviceMallocFinegrained);
        if (he == hipSuccess)
        {
            hipIpcMemHandle_t probe;
            if (hipIpcGetMemHandle(&probe, p.flags) != hipSuccess)  // (a runtime that cannot publish such memory: back to plain memory)
            {
                (void)hipGetLastError();
                (void)hipFree(p.flags);
                p.flags = nullptr;
                he = hipErrorUnknown;
            }
        }
        else
        {
            (void)hipGetLastError();
            p.flags = nullptr;
        }
    }
    p.flags_fine = he == hipSuccess;
    if (he != hipSuccess) he = hipMalloc(reinterpret_cast<void**>(&p.flags), kP2PFlagsAlloc);
    if (e->tuning.verbose) std::fprintf(stderr, "[ddgi p2p, rank %d of %d] flag words in %s device memory\n", e->rank, e->world, p.flags_fine ? "fine-grained" : "coarse-grained");
    // ON THE HANDLE'S STREAM, which is synchronised below before the address leaves this call.  (Rounds 3 - 5 zeroed the words with hipMemset: asynchronous
    // for device memory, on the null stream, which nothing here ever waited for.  On a GPU that other processes keep full — four ranks' persistent trace
    // kernels on the test box's one GPU — the fill could run AFTER the peers' first flag writes had landed and wipe them: `ready` 0 / `arrived` 1 from every
    // peer at once in round 6's logs, a state no peer can produce; with the in-order exchange nobody writes a higher number later, and the rank waits
    // forever.  That is GPUTEST_r05's hang: docs/LAB_NOTES.md "Round 6".)
    if (he == hipSuccess) he = hipMemsetAsync(p.flags, 0, 2 * kP2PMaxWorld * sizeof(uint32_t), e->stream);
    if (he == hipSuccess)
    {
        int lo = 0, hi = 0;  // (numerically lower = higher priority)
        he = hipDeviceGetStreamPriorityRange(&lo, &hi);
        if (he == hipSuccess) he = hipStreamCreateWithPriority(&p.diag, hipStreamNonBlocking, hi);
        if (he == hipSuccess) he = hipHostMalloc(reinterpret_cast<void**>(&p.diag_host), 2 * kP2PMaxWorld * sizeof(uint32_t), hipHostMallocDefault);
    }
    if (he == hipSuccess) he = hipStreamSynchronize(e->stream);  // (the second pair's first contents)
    if (he == hipSuccess) he = hipIpcGetMemHandle(&a.flags, p.flags);
    // What the peers map: the handle's own rings (their pushes land where consumers read) — unless a ring reaches 2 GiB: on the stack measured a buffer of
    // 2^31 bytes or more is not handed to another process reliably (round 6: hipIpcOpenMemHandle that never returns, hipIpcGetMemHandle that refuses;
    // profiles/r06_p2p_ring_size_bisection.txt, r06_ipc_stage_probe.txt).  Then — or with DDGI_P2P_LANDING=1 in the environment (tests) — the peers get
    // LANDING ZONES: per texture and parity of the exchange's number a buffer of world - 1 slabs, (world - 1) / world of ONE pair, whatever the ring's depth.
    const int n_tex_now = e->mode == DDGI_MODE_DDGI ? 2 : 1;
    bool landing = false;
    {
        const char* force = std::getenv("DDGI_P2P_LANDING");
        landing = force && force[0] == '1';
        for (int i = 0; i < n_tex_now; ++i)
            if (e->tex_bytes[i] * static_cast<size_t>(e->np) >= (static_cast<size_t>(2) << 30)) landing = true;
        if (e->world < 2) landing = false;
    }
    if (he == hipSuccess && landing)
    {
        p.landing = true, p.land_ntex = n_tex_now;
        for (int i = 0; i < n_tex_now && he == hipSuccess; ++i)
        {
            const size_t zone = e->tex_bytes[i] / static_cast<size_t>(e->world) * static_cast<size_t>(e->world - 1);
            if (zone >= (static_cast<size_t>(2) << 30))
            {
                const int rc = fail(DDGI_ERR_UNSUPPORTED, "the peer-to-peer exchange of this grid needs landing zones of %.2f GB; buffers of 2 GiB and more are not shared between processes reliably on this stack — use the RCCL transport", zone / 1e9);
                ddgi_exchange_release(e);
                return rc;
            }
            for (int b = 0; b < 2 && he == hipSuccess; ++b)
            {
                he = hipMalloc(&p.land[i][b], zone);
                if (he == hipSuccess) he = hipIpcGetMemHandle(b ? &a.ring_b[i] : &a.ring[i], p.land[i][b]);
            }
        }
        if (he == hipSuccess) he = create_exchange_stream(&p.gather);
        for (int k = 0; k < ddgi_engine::kMaxPairs && he == hipSuccess; ++k) he = hipEventCreateWithFlags(&p.gathered[k], hipEventDisableTiming);
        for (int b = 0; b < 2 && he == hipSuccess; ++b) he = hipEventCreateWithFlags(&p.drained[b], hipEventDisableTiming);
        a.landing = static_cast<uint32_t>(n_tex_now);
        if (e->tuning.verbose)
            std::fprintf(stderr, "[ddgi p2p, rank %d of %d] landing zones: %d texture(s) x 2 parities, %.1f + %.1f MB each (the rings: %d pairs of %.1f + %.1f MB)\n", e->rank, e->world, n_tex_now,
                         e->tex_bytes[0] / static_cast<double>(e->world) * (e->world - 1) / 1e6, n_tex_now > 1 ? e->tex_bytes[1] / static_cast<double>(e->world) * (e->world - 1) / 1e6 : 0.0, e->np,
                         e->tex_bytes[0] / 1e6, e->tex_bytes[1] / 1e6);
    }
    else
        for (int i = 0; i < 2 && he == hipSuccess; ++i) he = hipIpcGetMemHandle(&a.ring[i], e->own_tex[i]);
    for (int i = 0; i < 2; ++i) a.tex_bytes[i] = e->tex_bytes[i];
    if (he != hipSuccess)
    {
        const int rc = fail(DDGI_ERR_HIP, "publishing the probe textures for peer access failed: %s", hipGetErrorString(he));
        ddgi_exchange_release(e);
        return rc;
    }
    std::memset(address, 0, DDGI_P2P_ADDRESS_BYTES);
    std::memcpy(address, &a, sizeof a);
    return DDGI_OK;
}

int ddgi_exchange_p2p_init(ddgi_handle e, const uint8_t* addresses, int world)
{
    if (!e || !addresses) return fail(DDGI_ERR_INVALID_ARGUMENT, "null handle/addresses");
    ddgi_engine::Exchange& x = e->xch;
    if (!x.p2p || x.p2p->connected) return fail(DDGI_ERR_NOT_READY, "ddgi_exchange_p2p_init needs a fresh ddgi_exchange_p2p_export on this handle");
    if (world != e->world) return fail(DDGI_ERR_INVALID_ARGUMENT, "%d addresses for a handle that is slab %d of %d", world, e->rank, e->world);
    HIP_TRY(hipSetDevice(e->device));
    ddgi_engine::P2P& p = *x.p2p;
    p.peers.assign(static_cast<size_t>(world), ddgi_engine::P2P::Peer{});
    int rc = DDGI_OK;
    for (int q = 0; q < world && rc == DDGI_OK; ++q)
    {
        P2PAddress a;
        std::memcpy(&a, addresses + static_cast<size_t>(q) * DDGI_P2P_ADDRESS_BYTES, sizeof a);
        if (a.magic != kP2PMagic || static_cast<int>(a.rank) != q || static_cast<int>(a.world) != world || (a.pipelined != 0) != x.pipelined || a.tex_bytes[0] != e->tex_bytes[0] ||
            a.tex_bytes[1] != e->tex_bytes[1] || static_cast<int>(a.np) != e->np || static_cast<int>(a.landing) != (p.landing ? p.land_ntex : 0))
        {
            rc = fail(DDGI_ERR_INVALID_ARGUMENT, "address %d does not describe rank %d of %d with this handle's textures, mode, pipelining and frames in flight (and landing zones on every rank or on none)", q, q, world);
            break;
        }
        if (q == e->rank) continue;
        ddgi_engine::P2P::Peer& peer = p.peers[static_cast<size_t>(q)];
        hipError_t he = hipSuccess;
        if (a.process == process_nonce())
        {
            // The rendezvous is a flag another rank writes LATER, waited for by the command processor.  Between processes each
            // rank's queues make progress on their own; inside one process the host enqueues rank after rank, and two handles'
            // streams may share a hardware queue — rank 0's wait would sit in front of the very write it waits for.
            rc = fail(DDGI_ERR_UNSUPPORTED, "rank %d lives in this process: the peer-to-peer exchange is one process per rank (one process driving several handles uses the RCCL transport with ddgi_exchange_group_begin/end)", q);
            break;
        }
        peer.ipc = true;
        // hipIpcOpenMemHandle is a driver call that can stand forever (round 6: a texture ring of 2 GiB or more — the importer waits in recvmsg for the exporter's
        // process, profiles/r06_c5_bring_up_backtrace.txt, r06_p2p_ring_size_bisection.txt).  A blocking call cannot be given a deadline, so it runs on a helper thread that
        // this call waits for with tuning "wait_timeout_ms"; one that does not come back is LEFT BEHIND (with its result block, which it owns) and the call
        // returns DDGI_ERR_TIMEOUT naming the peer and the buffer.  (tuning "verbose": every mapping with its time on stderr.)
        struct MapJob
        {
            hipIpcMemHandle_t buf[5];  // flags, then the rings (or: the landing zones of texture 0 / 1, parity 0, then parity 1)
            bool have[5] = {true, false, false, false, false};
            int device = 0, rank = 0, world = 0, q = 0;
            bool say = false;
            std::mutex mu;
            std::condition_variable cv;
            int stage = 0;  // the buffer being mapped; 5 done
            bool finished = false;
            hipError_t he = hipSuccess;
            void* ptr[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
        };
        auto job = std::make_shared<MapJob>();
        job->buf[0] = a.flags, job->buf[1] = a.ring[0], job->buf[2] = a.ring[1], job->buf[3] = a.ring_b[0], job->buf[4] = a.ring_b[1];
        if (p.landing)
            for (int i = 0; i < p.land_ntex; ++i) job->have[1 + i] = job->have[3 + i] = true;
        else
            job->have[1] = job->have[2] = true;
        job->device = e->device, job->rank = e->rank, job->world = world, job->q = q, job->say = e->tuning.verbose != 0;
        if (job->say)
            std::fprintf(stderr, "[ddgi p2p, rank %d of %d] mapping rank %d (pid %d): flags, then %s (%u pairs of %.1f + %.1f MB in its rings)\n", e->rank, world, q, a.pid, p.landing ? "its landing zones" : "its 2 rings", a.np,
                         a.tex_bytes[0] / 1e6, a.tex_bytes[1] / 1e6);
        bool stall = false;
#ifdef DDGI_PROFILING
        stall = (e->tuning.ablate & 64) != 0;  // fault injection (profiling build only, tuning "ablate" 64): the mapping thread does not come back for a minute
#endif
        auto work = [job, stall]() {
            if (stall) std::this_thread::sleep_for(std::chrono::seconds(60));
            hipError_t r = hipSetDevice(job->device);
            static const char* const names[5] = {"flags", "ring / zone 0", "ring / zone 1", "zone 0 (odd exchanges)", "zone 1 (odd exchanges)"};
            for (int i = 0; i < 5 && r == hipSuccess; ++i)
            {
                if (!job->have[i]) continue;
                {
                    std::lock_guard<std::mutex> lock(job->mu);
                    job->stage = i;
                }
                const auto t0 = std::chrono::steady_clock::now();
                r = hipIpcOpenMemHandle(&job->ptr[i], job->buf[i], hipIpcMemLazyEnablePeerAccess);
                if (job->say)
                    std::fprintf(stderr, "[ddgi p2p, rank %d of %d] %s of rank %d: %s after %.3f s\n", job->rank, job->world, names[i], job->q, hipGetErrorString(r),
                                 std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
            }
            std::lock_guard<std::mutex> lock(job->mu);
            job->he = r, job->stage = 5, job->finished = true;
            job->cv.notify_all();
        };
        const int limit = e->tuning.wait_timeout_ms;
        if (limit <= 0)
            work();
        else
        {
            std::thread(work).detach();
            std::unique_lock<std::mutex> lock(job->mu);
            if (!job->cv.wait_for(lock, std::chrono::milliseconds(limit), [&] { return job->finished; }))
            {
                const int at = job->stage;
                const int ti = at == 0 ? 0 : (at - 1) & 1;
                const double gb = at == 0 ? 0.0 : (p.landing ? static_cast<double>(a.tex_bytes[ti]) / world * (world - 1) : static_cast<double>(a.tex_bytes[ti]) * a.np) / 1e9;
                rc = fail(DDGI_ERR_TIMEOUT, "hipIpcOpenMemHandle of rank %d's %s (%.2f GB) did not return within %d ms (tuning \"wait_timeout_ms\"): the call is left behind on a helper thread.  "
                                            "Known: inside an engine's process a buffer of 2 GiB or more never comes back from this call on the stack measured (profiles/r06_p2p_ring_size_bisection.txt); "
                                            "use the RCCL transport for such grids",
                          q, at == 0 ? "flag words" : (p.landing ? (ti ? "landing zone of the second texture" : "landing zone of the first texture") : (ti ? "second texture ring" : "first texture ring")), gb, limit);
                peer.ipc = false;  // (nothing of this peer is mapped as far as this handle knows: the helper thread owns whatever it gets)
                break;
            }
        }
        he = job->he;
        peer.flags = static_cast<uint32_t*>(job->ptr[0]);
        peer.ring[0] = job->ptr[1], peer.ring[1] = job->ptr[2], peer.ring_b[0] = job->ptr[3], peer.ring_b[1] = job->ptr[4];
        if (he == hipSuccess) he = create_exchange_stream(&peer.stream);
        if (he == hipSuccess) he = hipEventCreateWithFlags(&peer.done, hipEventDisableTiming);
        if (he != hipSuccess && rc == DDGI_OK) rc = fail(DDGI_ERR_HIP, "mapping rank %d's probe textures failed: %s", q, hipGetErrorString(he));
    }
    if (rc)
    {
        ddgi_exchange_release(e);
        return rc;
    }
    p.connected = true;
    x.transport = DDGI_EXCHANGE_P2P;
    return DDGI_OK;
}

int ddgi_exchange_transport(ddgi_handle e, int* transport, int* pipelined)
{
    if (!e) return fail(DDGI_ERR_INVALID_ARGUMENT, "null handle");
    if (transport) *transport = e->xch.transport;
    if (pipelined) *pipelined = e->xch.pipelined ? 1 : 0;
    return DDGI_OK;
}

int ddgi_exchange_ranks(ddgi_handle e, int* ranks)
{
    if (!e || !ranks) return fail(DDGI_ERR_INVALID_ARGUMENT, "null handle/ranks");
    *ranks = 0;
    const ddgi_engine::Exchange& x = e->xch;
    if (x.transport == DDGI_EXCHANGE_RCCL)
    {
        if (int rc = rccl_ready()) return rc;
        NCCL_TRY(rccl().CommCount(x.comm, ranks));
    }
    else if (x.transport == DDGI_EXCHANGE_P2P && x.p2p)
    {
        int n = 1;
        for (int q = 0; q < static_cast<int>(x.p2p->peers.size()); ++q)
            if (q != e->rank && x.p2p->peers[static_cast<size_t>(q)].ring[0] && x.p2p->peers[static_cast<size_t>(q)].flags) n += 1;
        *ranks = n;
    }
    return DDGI_OK;
}

int ddgi_exchange(ddgi_handle e)
{
    if (!e) return fail(DDGI_ERR_INVALID_ARGUMENT, "null handle");
    ddgi_engine::Exchange& x = e->xch;
    if (!x.transport) return fail(DDGI_ERR_NOT_READY, "ddgi_exchange before ddgi_exchange_init / ddgi_exchange_p2p_init");
    if (x.broken)
        return fail(DDGI_ERR_TIMEOUT, "the multi-GPU exchange of this handle is broken (a wait for another rank ran into \"wait_timeout_ms\" earlier): attach it again on every rank");
    if (x.desync)
        return fail(DDGI_ERR_NOT_READY, "an update of this rank failed while the exchange was attached: its count of updates — which texture pair an update writes and a "
                                        "peer's slab lands in — no longer matches the other ranks'.  Attach the exchange again on EVERY rank (ddgi_exchange_init / ddgi_exchange_p2p_export + _init)");
    HIP_TRY(hipSetDevice(e->device));
    e->box_of = nullptr;  // the other ranks' slabs are about to change: the sampler's per-texel table is stale
    if (x.transport == DDGI_EXCHANGE_P2P) return p2p_exchange(e);
    hipStream_t s = e->stream;
    if (x.pipelined)
    {
        hipEvent_t written = nullptr;
        if (int rc = update_written_event(e, &written)) return rc;
        HIP_TRY(hipStreamWaitEvent(x.comm_stream, written, 0));
        s = x.comm_stream;
    }
    // REF mode: the reference never assigns its `distances` image (probe_pass.comp:276,302) — every rank's copy is
    // all zeros by construction, exchanging it would move no information
    const int n_tex = e->mode == DDGI_MODE_DDGI ? 2 : 1;
    NCCL_TRY(rccl().GroupStart());
    for (int i = 0; i < n_tex; ++i)
    {
        const size_t slab = e->tex_bytes[i] / static_cast<size_t>(e->world);
        uint8_t* full = static_cast<uint8_t*>(e->tex[i]);
        NCCL_TRY(rccl().AllGather(full + slab * static_cast<size_t>(e->rank), full, slab, kNcclUint8, x.comm, s));
    }
    NCCL_TRY(rccl().GroupEnd());
    if (x.pipelined)
    {
        if (g_group_depth > 0)
        {
            // inside ddgi_exchange_group_begin/end the all-gather is not on comm_stream yet: an event recorded now would fire
            // before it.  Until the bracket closes the pair counts as "exchange not over" for nobody — consumers and the next
            // update only come after ddgi_exchange_group_end, which records it.
            x.sent_valid[e->pair_cur] = false;
            std::lock_guard<std::mutex> lock(g_group_mu);
            g_group_pending.push_back(PendingSent{e, e->pair_cur, std::this_thread::get_id()});
        }
        else
        {
            HIP_TRY(hipEventRecord(x.sent[e->pair_cur], x.comm_stream));
            x.sent_valid[e->pair_cur] = true;
        }
    }
    return DDGI_OK;
}

int ddgi_exchange_finish(ddgi_handle e)
{
    if (!e) return fail(DDGI_ERR_INVALID_ARGUMENT, "null handle");
    ddgi_engine::Exchange& x = e->xch;
    if (!x.transport) return DDGI_OK;
    HIP_TRY(hipSetDevice(e->device));
    if (x.transport == DDGI_EXCHANGE_P2P)
    {
        for (int i = 0; i < ddgi_engine::kMaxPairs; ++i)
            if (x.sent_valid[i]) HIP_TRY(hipStreamWaitEvent(e->stream, x.sent[i], 0));
        return p2p_wait_arrived(e, x.p2p->seq);
    }
    if (!x.pipelined) return DDGI_OK;
    for (int i = 0; i < ddgi_engine::kMaxPairs; ++i)
        if (x.sent_valid[i]) HIP_TRY(hipStreamWaitEvent(e->stream, x.sent[i], 0));
    return DDGI_OK;
}

}  // extern "C"
