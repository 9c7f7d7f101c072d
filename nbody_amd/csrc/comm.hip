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
#include <cstdlib>
#include <cstring>
#include <rccl/rccl.h>
#include <vector>

struct NbComm {
    void* dl = nullptr;
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    std::vector<uint64_t> begin, count;  // every rank's target range (all-pairs)
    bool equal_counts = true;
    int pair_path = -1;  // all-pairs, PLUMMER: 1 every rank can run the pair kernel, 0 some rank cannot, -1 not asked yet
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
    // NBODY_RCCL_LIB names another library with RCCL's entry points (a test transport, a newer
    // RCCL); nothing else is tried then
    if (const char* over = std::getenv("NBODY_RCCL_LIB")) {
        if (*over) return dlopen(over, RTLD_NOW | RTLD_LOCAL);
    }
    // The RCCL that sits beside the HIP runtime THIS library is bound to.  A process can hold two ROCm
    // stacks -- PyTorch ships its own libamdhip64 / libhsa-runtime64 / librccl, loaded under other file
    // names, so the loader does not merge them with the system's -- and an RCCL of the other stack talks
    // to a HIP runtime that knows nothing of this library's device, stream and buffers ("no ROCm-capable
    // device is detected" from ncclCommInitRank).  Whichever stack was loaded first is the one this
    // library's own HIP calls resolve to; its RCCL is the one to use.
    Dl_info hip{};
    if (dladdr(reinterpret_cast<void*>(&hipGetDeviceCount), &hip) && hip.dli_fname && std::strstr(hip.dli_fname, "amdhip")) {
        std::string dir(hip.dli_fname);
        const size_t slash = dir.rfind('/');
        if (slash != std::string::npos) {
            dir.resize(slash + 1);
            for (const char* name : {"librccl.so.1", "librccl.so"}) {
                if (void* h = dlopen((dir + name).c_str(), RTLD_NOW | RTLD_LOCAL)) return h;
            }
        }
    }
    // otherwise a copy already in the process (same SONAME: the loader would hand it out anyway)
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

// ---- who holds what: plain arithmetic, no device --------------------------------------------
// Rank k's run of the sorted order: [b(k), b(k + 1)) with b(k) = n k / world rounded down to a
// multiple of 64 (whole wavefronts of targets), b(world) = n.  Runs may be empty (n < 64 world).
void nb_sorted_run(uint64_t n, uint64_t world, uint64_t k, uint64_t* begin, uint64_t* count) {
    auto bound = [&](uint64_t r) { return r >= world ? n : (n * r / world) / 64 * 64; };
    *begin = bound(k);
    *count = bound(k + 1) - *begin;
}

// How the ranks' ranges [begin[k], begin[k] + count[k]) are exchanged.  NB_PLAN_ALLGATHER: they tile
// [0, n) in rank order with equal counts, so begin[k] == k * count[0] and ONE in-place all-gather
// does it; NB_PLAN_BROADCASTS: they tile [0, n) with unequal (possibly empty) counts, one broadcast
// per non-empty rank; NB_PLAN_REPLICATED: every rank holds all bodies, nothing to exchange;
// NB_PLAN_INVALID: anything else (gaps, overlaps, out of rank order).
int nb_exchange_plan(uint64_t n, int world, const uint64_t* begin, const uint64_t* count) {
    bool tiles = true, whole = true, equal = true;
    uint64_t next = 0;
    for (int k = 0; k < world; ++k) {
        tiles = tiles && begin[k] == next && count[k] <= n - next;
        if (tiles) next = begin[k] + count[k];
        whole = whole && begin[k] == 0 && count[k] == n;
        equal = equal && count[k] == count[0];
    }
    tiles = tiles && next == n;
    if (tiles) return equal ? NB_PLAN_ALLGATHER : NB_PLAN_BROADCASTS;
    return whole ? NB_PLAN_REPLICATED : NB_PLAN_INVALID;
}

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
    for (int k = 0; k < world; ++k) {
        C->begin[k] = all[2 * k];
        C->count[k] = all[2 * k + 1];
    }
    // either the ranks' target ranges tile [0, N) in rank order (all-pairs), or every rank holds
    // all bodies (Barnes-Hut with replicated state; also all-pairs on one rank)
    const int plan = nb_exchange_plan(ctx->n, world, C->begin.data(), C->count.data());
    if (plan == NB_PLAN_INVALID)
        return fail(NBODY_ERR_INVALID, "nbody_comm_init: the ranks' shard ranges must tile [0, n_bodies) in rank order, or every rank must hold all bodies");
    C->equal_counts = plan != NB_PLAN_BROADCASTS;
    return NBODY_OK;
}

// Whether EVERY rank can run the pair kernel (its scratch depends on the free memory of each GPU):
// asked once, the answer kept.  A rank that took the other path alone would leave the rest waiting
// in the all-reduce.
int nb_comm_all_agree(nbody_ctx* ctx, int mine, int* all) {
    NbComm* C = ctx->comm;
    int* dev = nullptr;
    if (hipMalloc(&dev, sizeof(int)) != hipSuccess) {
        (void)hipGetLastError();
        return nb_fail(ctx, NBODY_ERR_NOMEM, "nbody_step_allpairs: out of device memory");
    }
    int got = 0;
    hipError_t e = hipMemcpyAsync(dev, &mine, sizeof mine, hipMemcpyHostToDevice, ctx->stream);
    ncclResult_t r = ncclSuccess;
    if (e == hipSuccess) r = C->AllReduce(dev, dev, 1, ncclInt32, ncclMin, C->comm, ctx->stream);
    if (e == hipSuccess && r == ncclSuccess) e = hipMemcpyAsync(&got, dev, sizeof got, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess && r == ncclSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(dev);
    if (r != ncclSuccess) return nb_fail(ctx, NBODY_ERR_HIP, std::string("ncclAllReduce: ") + C->GetErrorString(r));
    if (e != hipSuccess) return nb_fail(ctx, NBODY_ERR_HIP, std::string("nbody_step_allpairs: ") + hipGetErrorString(e));
    *all = got;
    return NBODY_OK;
}

int nb_comm_pair_path(nbody_ctx* ctx, bool fits_here, bool* use_pair) {
    NbComm* C = ctx->comm;
    if (C->pair_path < 0) {
        int all = 0;
        if (int rc = nb_comm_all_agree(ctx, fits_here ? 1 : 0, &all)) return rc;
        C->pair_path = all ? 1 : 0;
    }
    *use_pair = C->pair_path == 1;
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

void nb_comm_run(const nbody_ctx* ctx, int k, uint64_t* begin, uint64_t* count) {
    nb_sorted_run(ctx->n, (uint64_t)ctx->comm->world, (uint64_t)k, begin, count);
}

int nb_comm_gather_ranked(nbody_ctx* ctx) {
    NbComm* C = ctx->comm;
    std::vector<uint64_t> begin(C->world), count(C->world);
    for (int k = 0; k < C->world; ++k) nb_comm_run(ctx, k, &begin[k], &count[k]);
    const bool equal = nb_exchange_plan(ctx->n, C->world, begin.data(), count.data()) == NB_PLAN_ALLGATHER;
    return exchange_rows(ctx, ctx->acc_tmp, begin, count, equal);
}

// buf holds world parts of bytes_per_rank bytes; this rank's part (at rank * bytes_per_rank) reaches every rank.
int nb_comm_allgather_inplace(nbody_ctx* ctx, void* buf, size_t bytes_per_rank) {
    NbComm* C = ctx->comm;
    unsigned char* b = static_cast<unsigned char*>(buf);
    NB_NCCL(ctx, C, C->AllGather(b + (size_t)C->rank * bytes_per_rank, b, bytes_per_rank, ncclUint8, C->comm, ctx->stream));
    return NBODY_OK;
}

// Several broadcasts as one group (one per rank and array of an exchange of unequal parts).  On its root an
// item's send and recv may be the same pointer; items of zero bytes are skipped (on every rank alike).
int nb_comm_broadcasts(nbody_ctx* ctx, const NbBroadcast* items, int count) {
    NbComm* C = ctx->comm;
    NB_NCCL(ctx, C, C->GroupStart());
    for (int i = 0; i < count; ++i) {
        if (!items[i].bytes) continue;
        ncclResult_t r = C->Broadcast(items[i].send, items[i].recv, items[i].bytes, ncclUint8, items[i].root, C->comm, ctx->stream);
        if (r != ncclSuccess) {
            C->GroupEnd();
            return nb_fail(ctx, NBODY_ERR_HIP, std::string("ncclBroadcast: ") + C->GetErrorString(r));
        }
    }
    NB_NCCL(ctx, C, C->GroupEnd());
    return NBODY_OK;
}

int nb_comm_rank(const nbody_ctx* ctx) { return ctx->comm->rank; }
int nb_comm_world(const nbody_ctx* ctx) { return ctx->comm->world; }
