// Multi-GPU exchange inside the library: one process per GPU, RCCL over xGMI on the context's stream.
//
// No counterpart in the reference (single device).  With a communicator attached
// (nbody_comm_init), nbody_step_allpairs and nbody_step_bh are complete steps for a sharded run:
//   all-pairs, target shards     forces + integration of the context's own targets, then every
//                                rank's positions are gathered in place (16 B per body)
//   all-pairs, pair shards       (PLUMMER, N >= 65,536) the pair kernel's work items are dealt over
//                                the ranks, the per-body contributions are all-reduced, own targets
//                                integrated, positions gathered
//   Barnes-Hut, replicated state every rank builds the same octree, traverses its run of the
//                                sorted order, the accelerations are gathered by rank and every
//                                rank integrates all bodies (states stay bit-identical)
// RCCL is loaded at nbody_comm_init (dlopen), so the library itself has no link-time dependency
// on it; a process that already holds an RCCL (PyTorch ships one) shares that copy.
#include "ctx.h"
#include <dlfcn.h>
#include <cstring>
#include <rccl/rccl.h>
#include <vector>

struct NbComm {
    void* dl = nullptr;
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    std::vector<uint64_t> begin, count;  // every rank's target range (all-pairs)
    bool equal_counts = true;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

namespace {

void* open_rccl() {
    // a copy already in the process first (same SONAME: the loader would hand it out anyway)
    for (const char* name : {"librccl.so.1", "librccl.so"}) {
        if (void* h = dlopen(name, RTLD_NOW | RTLD_NOLOAD)) return h;
    }
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        if (void* h = dlopen(name, RTLD_NOW | RTLD_LOCAL)) return h;
    }
    return nullptr;
}

#define NB_NCCL(ctx, C, call)                                                                  \
    do {                                                                                       \
        ncclResult_t _r = (call);                                                              \
        if (_r != ncclSuccess)                                                                 \
            return nb_fail(ctx, NBODY_ERR_HIP, std::string(#call) + ": " + (C)->GetErrorString(_r)); \
    } while (0)

template <class F>
bool bind(void* dl, const char* name, F& fn) {
    fn = reinterpret_cast<F>(dlsym(dl, name));
    return fn != nullptr;
}

}  // namespace

int nb_comm_unique_id(void* id128) {
    void* dl = open_rccl();
    if (!dl) return NBODY_ERR_NO_DEVICE;
    decltype(&ncclGetUniqueId) get = nullptr;
    if (!bind(dl, "ncclGetUniqueId", get)) return NBODY_ERR_HIP;
    ncclUniqueId id;
    if (get(&id) != ncclSuccess) return NBODY_ERR_HIP;
    static_assert(sizeof(ncclUniqueId) == 128, "unique id size");
    std::memcpy(id128, &id, sizeof id);
    return NBODY_OK;
}

void nb_comm_destroy(nbody_ctx* ctx) {
    NbComm* C = ctx->comm;
    if (!C) return;
    if (C->comm && C->CommDestroy) C->CommDestroy(C->comm);
    // the RCCL handle stays open: other contexts, or the host, may be using the same copy
    delete C;
    ctx->comm = nullptr;
}

int nb_comm_init(nbody_ctx* ctx, const void* id128, int rank, int world) {
    if (ctx->comm) return nb_fail(ctx, NBODY_ERR_STATE, "nbody_comm_init: the context already has a communicator");
    NbComm* C = new NbComm;
    C->dl = open_rccl();
    if (!C->dl) {
        delete C;
        return nb_fail(ctx, NBODY_ERR_NO_DEVICE, std::string("nbody_comm_init: cannot load librccl: ") + dlerror());
    }
    const bool ok = bind(C->dl, "ncclCommInitRank", C->CommInitRank) && bind(C->dl, "ncclCommDestroy", C->CommDestroy) &&
                    bind(C->dl, "ncclAllGather", C->AllGather) && bind(C->dl, "ncclAllReduce", C->AllReduce) &&
                    bind(C->dl, "ncclBroadcast", C->Broadcast) && bind(C->dl, "ncclGroupStart", C->GroupStart) &&
                    bind(C->dl, "ncclGroupEnd", C->GroupEnd) && bind(C->dl, "ncclGetErrorString", C->GetErrorString);
    if (!ok) {
        delete C;
        return nb_fail(ctx, NBODY_ERR_HIP, "nbody_comm_init: librccl lacks a required symbol");
    }
    C->rank = rank;
    C->world = world;
    ctx->comm = C;
    auto fail = [&](int code, const std::string& msg) {
        nb_comm_destroy(ctx);
        return nb_fail(ctx, code, msg);
    };
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof id);
    ncclResult_t r = C->CommInitRank(&C->comm, world, id, rank);
    if (r != ncclSuccess) {
        C->comm = nullptr;
        return fail(NBODY_ERR_HIP, std::string("ncclCommInitRank: ") + C->GetErrorString(r));
    }
    // every rank's target range, so that unequal shards can be exchanged and the ranges checked
    uint64_t* dev = nullptr;
    if (hipMalloc(&dev, sizeof(uint64_t) * 2 * (size_t)world) != hipSuccess) return fail(NBODY_ERR_NOMEM, "nbody_comm_init: out of device memory");
    const uint64_t mine[2] = {ctx->i_begin, ctx->i_count};
    std::vector<uint64_t> all(2 * (size_t)world);
    hipError_t e = hipMemcpyAsync(dev + 2 * rank, mine, sizeof mine, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) r = C->AllGather(dev + 2 * rank, dev, 2, ncclUint64, C->comm, ctx->stream);
    if (e == hipSuccess && r == ncclSuccess)
        e = hipMemcpyAsync(all.data(), dev, all.size() * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess && r == ncclSuccess) e = hipStreamSynchronize(ctx->stream);
    hipFree(dev);
    if (r != ncclSuccess) return fail(NBODY_ERR_HIP, std::string("ncclAllGather: ") + C->GetErrorString(r));
    if (e != hipSuccess) return fail(NBODY_ERR_HIP, std::string("nbody_comm_init: ") + hipGetErrorString(e));
    C->begin.resize(world);
    C->count.resize(world);
    bool tiles = true, whole = true;
    uint64_t next = 0;
    for (int k = 0; k < world; ++k) {
        C->begin[k] = all[2 * k];
        C->count[k] = all[2 * k + 1];
        tiles = tiles && C->begin[k] == next;
        next = C->begin[k] + C->count[k];
        whole = whole && C->begin[k] == 0 && C->count[k] == ctx->n;
        C->equal_counts = C->equal_counts && C->count[k] == C->count[0];
    }
    tiles = tiles && next == ctx->n;
    // either the ranks' target ranges tile [0, N) in rank order (all-pairs), or every rank holds
    // all bodies (Barnes-Hut with replicated state; also all-pairs on one rank)
    if (!tiles && !whole)
        return fail(NBODY_ERR_INVALID, "nbody_comm_init: the ranks' shard ranges must tile [0, n_bodies) in rank order, or every rank must hold all bodies");
    return NBODY_OK;
}

// buf[begin_k, begin_k + count_k) of rank k reaches every rank, in place (rows of 16 bytes).
static int exchange_rows(nbody_ctx* ctx, float4* buf, const std::vector<uint64_t>& begin, const std::vector<uint64_t>& count,
                         bool equal) {
    NbComm* C = ctx->comm;
    if (equal) {
        NB_NCCL(ctx, C, C->AllGather(buf + begin[C->rank], buf, count[0] * 4, ncclFloat32, C->comm, ctx->stream));
        return NBODY_OK;
    }
    NB_NCCL(ctx, C, C->GroupStart());
    for (int k = 0; k < C->world; ++k) {
        if (!count[k]) continue;
        ncclResult_t r = C->Broadcast(buf + begin[k], buf + begin[k], count[k] * 4, ncclFloat32, k, C->comm, ctx->stream);
        if (r != ncclSuccess) {
            C->GroupEnd();
            return nb_fail(ctx, NBODY_ERR_HIP, std::string("ncclBroadcast: ") + C->GetErrorString(r));
        }
    }
    NB_NCCL(ctx, C, C->GroupEnd());
    return NBODY_OK;
}

int nb_comm_gather_positions(nbody_ctx* ctx) {
    NbComm* C = ctx->comm;
    if (ctx->i_count == ctx->n) return NBODY_OK;  // replicated state: nothing to exchange
    return exchange_rows(ctx, ctx->posm[ctx->cur], C->begin, C->count, C->equal_counts);
}

int nb_comm_allreduce_partial(nbody_ctx* ctx) {
    NbComm* C = ctx->comm;
    NB_NCCL(ctx, C, C->AllReduce(ctx->acc_tmp, ctx->acc_tmp, ctx->n * 4, ncclFloat32, ncclSum, C->comm, ctx->stream));
    return NBODY_OK;
}

// Rank k's run of the sorted order: [b(k), b(k + 1)) with b(k) = n k / world rounded down to a
// multiple of 64 (whole wavefronts of targets), b(world) = n.
void nb_comm_run(const nbody_ctx* ctx, int k, uint64_t* begin, uint64_t* count) {
    const uint64_t w = (uint64_t)ctx->comm->world;
    auto bound = [&](uint64_t r) { return r >= w ? ctx->n : (ctx->n * r / w) / 64 * 64; };
    *begin = bound((uint64_t)k);
    *count = bound((uint64_t)k + 1) - *begin;
}

int nb_comm_gather_ranked(nbody_ctx* ctx) {
    NbComm* C = ctx->comm;
    std::vector<uint64_t> begin(C->world), count(C->world);
    bool equal = true;
    for (int k = 0; k < C->world; ++k) {
        nb_comm_run(ctx, k, &begin[k], &count[k]);
        equal = equal && count[k] == count[0];
    }
    return exchange_rows(ctx, ctx->acc_tmp, begin, count, equal);
}

int nb_comm_rank(const nbody_ctx* ctx) { return ctx->comm->rank; }
int nb_comm_world(const nbody_ctx* ctx) { return ctx->comm->world; }
