// plade_amd/csrc/comm.hip -- the exchange step of the multi-GPU paths over RCCL / xGMI, bound at run time.
//
// What the path exchanges is small and rare: batch mode (code/PLADE/main.cpp:122-148 is a loop over independent pairs,
// sharded pair i -> rank i % world) gathers 68 bytes of result per pair once per batch; the candidate axis of one pair
// (the verification loop code/PLADE/plade.cpp:547-564, candidates k % world == rank per GPU) all-gathers 8 bytes per
// candidate once per registration.  Both are ONE ncclAllGather -- no all-reduce, so the per-link bound of ring collectives
// over point-to-point xGMI never gates the throughput (a gather of a few KB is latency, ~20-40 us, whatever the topology).
//
// librccl is opened with dlopen() on first use: libplade_hip.so keeps no link dependency on a communication library (a
// single-GPU host never loads it), and a host that lives in another communication world (MPI, torch.distributed) keeps
// using plade_set_candidate_shard's callback form.  The 128-byte unique id of ncclGetUniqueId travels between the ranks by
// whatever the host has (bench.py: the loopback rendezvous file; the CLI: a pipe from the parent process).
#include "ctx.h"
#include <rccl/rccl.h>      // types and enums only: every function is looked up in the dlopen()ed library
#include <dlfcn.h>
#include <mutex>

namespace plade {

struct RcclApi {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};

static RcclApi &rccl_api() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, []() {
        const char *names[] = {getenv("PLADE_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (const char *n : names) {
            if (!n || !*n) continue;
            api.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (api.lib) break;
        }
        if (!api.lib) { api.error = std::string("librccl not found: ") + (dlerror() ? dlerror() : "dlopen failed"); return; }
        auto sym = [&](const char *s) -> void * {
            void *p = dlsym(api.lib, s);
            if (!p && api.error.empty()) api.error = std::string("librccl lacks ") + s;
            return p;
        };
        api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
        api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
        api.AllGather = reinterpret_cast<decltype(api.AllGather)>(sym("ncclAllGather"));
        api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
        api.CommAbort = reinterpret_cast<decltype(api.CommAbort)>(sym("ncclCommAbort"));
        api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
    });
    return api;
}

static thread_local std::string g_comm_error;

}  // namespace plade

struct plade_comm {
    ncclComm_t comm = nullptr;
    int device = 0;
    uint32_t rank = 0, world = 1;
    hipStream_t stream = nullptr;
    plade::DBuf<char> d_send, d_recv;
    std::mutex m;     // a communicator serves one collective at a time
    std::string last_error;
};

using namespace plade;

#define RCCL_TRY(c, expr)                                                                                     \
    do {                                                                                                      \
        ncclResult_t _r = (expr);                                                                             \
        if (_r != ncclSuccess) {                                                                              \
            RcclApi &_a = rccl_api();                                                                         \
            throw plade::Err{PLADE_EDEVICE, std::string(#expr) + ": " + (_a.GetErrorString ? _a.GetErrorString(_r) : "rccl error")}; \
        }                                                                                                     \
    } while (0)

extern "C" const char *plade_comm_last_error(const plade_comm *c) { return c ? c->last_error.c_str() : g_comm_error.c_str(); }

extern "C" int plade_comm_unique_id(void *id128) {
    if (!id128) return PLADE_EINVAL;
    RcclApi &a = rccl_api();
    if (!a.error.empty() || !a.GetUniqueId) { g_comm_error = a.error; return PLADE_EDEVICE; }
    static_assert(sizeof(ncclUniqueId) == PLADE_COMM_ID_BYTES, "include/plade_hip.h states the size of ncclUniqueId");
    ncclUniqueId id;
    const ncclResult_t r = a.GetUniqueId(&id);
    if (r != ncclSuccess) { g_comm_error = std::string("ncclGetUniqueId: ") + a.GetErrorString(r); return PLADE_EDEVICE; }
    memcpy(id128, &id, sizeof(id));
    return PLADE_OK;
}

extern "C" int plade_comm_create(int device, uint32_t rank, uint32_t world, const void *id128, plade_comm **out) {
    if (!out) return PLADE_EINVAL;
    *out = nullptr;
    if (!id128 || world < 1 || rank >= world) { g_comm_error = "plade_comm_create: bad argument"; return PLADE_EINVAL; }
    RcclApi &a = rccl_api();
    if (!a.error.empty()) { g_comm_error = a.error; return PLADE_EDEVICE; }
    plade_comm *c = new plade_comm;
    c->device = device; c->rank = rank; c->world = world;
    try {
        HIP_TRY(hipSetDevice(device));
        HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        ncclUniqueId id;
        memcpy(&id, id128, sizeof(id));
        RCCL_TRY(c, a.CommInitRank(&c->comm, (int)world, id, (int)rank));
    } catch (const Err &e) {
        g_comm_error = e.msg;
        if (c->stream) (void)hipStreamDestroy(c->stream);
        delete c;
        return e.code;
    }
    *out = c;
    return PLADE_OK;
}

extern "C" void plade_comm_destroy(plade_comm *c) {
    if (!c) return;
    RcclApi &a = rccl_api();
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->comm && a.CommDestroy) (void)a.CommDestroy(c->comm);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

namespace plade {
// device buffers, on `stream` (the caller orders it after its producers and waits for it): recv = world x bytes
void comm_all_gather_dev(plade_comm *c, const void *d_send, void *d_recv, size_t bytes, hipStream_t stream) {
    RcclApi &a = rccl_api();
    std::lock_guard<std::mutex> lk(c->m);
    RCCL_TRY(c, a.AllGather(d_send, d_recv, bytes, ncclInt8, c->comm, stream));
}
}  // namespace plade

extern "C" int plade_comm_all_gather(plade_comm *c, const void *send, void *recv, uint64_t bytes) {
    if (!c || (bytes && (!send || !recv))) return PLADE_EINVAL;
    if (!bytes) return PLADE_OK;
    try {
        HIP_TRY(hipSetDevice(c->device));
        char *ds = c->d_send.ensure(bytes + 64), *dr = c->d_recv.ensure(bytes * c->world + 64);
        HIP_TRY(hipMemcpyAsync(ds, send, bytes, hipMemcpyHostToDevice, c->stream));
        comm_all_gather_dev(c, ds, dr, bytes, c->stream);
        HIP_TRY(hipMemcpyAsync(recv, dr, bytes * c->world, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    } catch (const Err &e) { c->last_error = e.msg; return e.code; }
    return PLADE_OK;
}

extern "C" int plade_set_candidate_shard_comm(plade_ctx *ctx, plade_comm *comm, uint32_t min_candidates) {
    if (!ctx) return PLADE_EINVAL;
    if (!comm) { ctx->shard = plade_ctx::CandidateShard{}; return PLADE_OK; }
    if (comm->device != ctx->device) { ctx->last_error = "plade_set_candidate_shard_comm: the communicator lives on another device"; return PLADE_EINVAL; }
    ctx->shard = plade_ctx::CandidateShard{};
    ctx->shard.rank = comm->rank; ctx->shard.world = comm->world; ctx->shard.min_candidates = min_candidates; ctx->shard.comm = comm;
    return PLADE_OK;
}
