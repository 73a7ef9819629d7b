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
#include <dlfcn.h>

#include <cstring>
#include <mutex>

#include "ddgi_engine.h"

namespace {

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

}  // namespace

// ---- hooks for the engine -------------------------------------------------------------------------------

int ddgi_exchange_before_update(ddgi_engine* e)
{
    ddgi_engine::Exchange& x = e->xch;
    if (!x.comm || !x.pipelined)
    {
        e->tex_prev[0] = e->tex_prev[1] = nullptr;
        return DDGI_OK;
    }
    const int cur = static_cast<int>(x.k & 1ull);
    x.k += 1;
    if (x.sent_valid[cur]) HIP_TRY(hipStreamWaitEvent(e->stream, x.sent[cur], 0));  // its previous exchange has left the buffers
    for (int i = 0; i < 2; ++i)
    {
        e->tex[i] = x.pair[cur][i];
        e->tex_prev[i] = x.pair[cur ^ 1][i];
    }
    x.cur = cur;
    return DDGI_OK;
}

int ddgi_exchange_wait_latest(ddgi_engine* e)
{
    ddgi_engine::Exchange& x = e->xch;
    if (!x.comm || !x.pipelined) return DDGI_OK;  // in-order exchange: stream order already covers it
    if (x.sent_valid[x.cur]) HIP_TRY(hipStreamWaitEvent(e->stream, x.sent[x.cur], 0));
    return DDGI_OK;
}

void ddgi_exchange_release(ddgi_engine* e)
{
    ddgi_engine::Exchange& x = e->xch;
    if (x.comm_stream) (void)hipStreamSynchronize(x.comm_stream);
    if (x.pipelined)
    {
        // pair[0] is the handle's own pair; pair[1] was allocated by ddgi_exchange_init
        for (int i = 0; i < 2; ++i)
        {
            if (x.pair[1][i] && x.pair[1][i] != e->own_tex[i]) (void)hipFree(x.pair[1][i]);
            e->tex[i] = e->own_tex[i];
            e->tex_prev[i] = nullptr;
        }
    }
    if (x.comm_stream) (void)hipStreamDestroy(x.comm_stream);
    if (x.written) (void)hipEventDestroy(x.written);
    for (auto& ev : x.sent)
        if (ev) (void)hipEventDestroy(ev);
    x = ddgi_engine::Exchange{};
}

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
    return DDGI_OK;
}

int ddgi_exchange_group_end(void)
{
    if (int rc = rccl_ready()) return rc;
    NCCL_TRY(rccl().GroupEnd());
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
    ddgi_engine::Exchange& x = e->xch;
    if (pipelined)
    {
        if (e->tex[0] != e->own_tex[0]) return fail(DDGI_ERR_INVALID_ARGUMENT, "the pipelined exchange alternates the handle's own texture pairs: unbind caller textures first");
        void* second[2];
        if (int rc = ddgi_alloc_texture_pair(e, e->tex_bytes, second)) return rc;
        int rc = DDGI_OK;
        hipError_t he = hipStreamCreateWithFlags(&x.comm_stream, hipStreamNonBlocking);
        if (he == hipSuccess) he = hipEventCreateWithFlags(&x.written, hipEventDisableTiming);
        for (int i = 0; i < 2 && he == hipSuccess; ++i) he = hipEventCreateWithFlags(&x.sent[i], hipEventDisableTiming);
        if (he != hipSuccess) rc = fail(DDGI_ERR_HIP, "exchange stream/event creation failed: %s", hipGetErrorString(he));
        for (int i = 0; i < 2; ++i)
        {
            x.pair[0][i] = e->own_tex[i];
            x.pair[1][i] = second[i];
        }
        x.pipelined = true;
        if (rc)
        {
            ddgi_exchange_release(e);
            return rc;
        }
        // the first update writes pair 0 (the tiles so far), mixing with pair 1: start pair 1 as a copy, so that a
        // DDGI field that has already converged carries on
        for (int i = 0; i < 2; ++i) HIP_TRY(hipMemcpyAsync(x.pair[1][i], x.pair[0][i], e->tex_bytes[i], hipMemcpyDeviceToDevice, e->stream));
    }
    x.comm = nccl_comm;
    return DDGI_OK;
}

int ddgi_exchange(ddgi_handle e)
{
    if (!e) return fail(DDGI_ERR_INVALID_ARGUMENT, "null handle");
    ddgi_engine::Exchange& x = e->xch;
    if (!x.comm) return fail(DDGI_ERR_NOT_READY, "ddgi_exchange before ddgi_exchange_init");
    HIP_TRY(hipSetDevice(e->device));
    hipStream_t s = e->stream;
    if (x.pipelined)
    {
        HIP_TRY(hipEventRecord(x.written, e->stream));
        HIP_TRY(hipStreamWaitEvent(x.comm_stream, x.written, 0));
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
        HIP_TRY(hipEventRecord(x.sent[x.cur], x.comm_stream));
        x.sent_valid[x.cur] = true;
    }
    return DDGI_OK;
}

int ddgi_exchange_finish(ddgi_handle e)
{
    if (!e) return fail(DDGI_ERR_INVALID_ARGUMENT, "null handle");
    ddgi_engine::Exchange& x = e->xch;
    if (!x.comm || !x.pipelined) return DDGI_OK;
    HIP_TRY(hipSetDevice(e->device));
    for (int i = 0; i < 2; ++i)
        if (x.sent_valid[i]) HIP_TRY(hipStreamWaitEvent(e->stream, x.sent[i], 0));
    return DDGI_OK;
}

}  // extern "C"
