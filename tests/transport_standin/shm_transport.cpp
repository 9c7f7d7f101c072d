// TEST INFRASTRUCTURE, not part of the product: a stand-in for the RCCL entry points that
// libnbody_hip.so binds (nbody_amd/csrc/comm.hip), so that the library's multi-rank exchange can run
// with several ranks SHARING ONE GPU.  RCCL itself refuses two ranks on one device, and the boxes
// the GPU tests run on have one GPU.  Selected with the environment variable NBODY_RCCL_LIB.
//
// Transport: every collective is staged through one POSIX shared-memory segment (a header with a
// barrier + one slot per rank), in chunks of the slot size.  Ordering: the call first waits for the
// stream it is given, then copies device -> slot, meets the other ranks at a barrier, copies
// slot(s) -> device on that same stream and waits for it, and meets them again before any slot is
// reused.  So a collective sees everything enqueued before it on the stream, and everything enqueued
// after it sees its result -- the ordering RCCL gives, without the overlap.  Sums run over the ranks
// in rank order on every rank, so all ranks obtain bit-identical results (as ring/tree RCCL does).
// Calls between ncclGroupStart and ncclGroupEnd are simply executed one by one: every rank issues
// them in the same order, which is all these blocking collectives need.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <vector>

namespace {

struct Header {
    std::atomic<uint32_t> arrived;
    std::atomic<uint32_t> generation;
    std::atomic<uint32_t> failed;  // a rank gave up (time-out, HIP error): nobody waits for it any more
    uint32_t pad[13];
};
static_assert(sizeof(Header) == 64, "header is one cache line");

struct Comm {
    int rank = 0, world = 1;
    size_t slot = 0;  // bytes per rank
    Header* head = nullptr;
    unsigned char* slots = nullptr;
    size_t mapped = 0;
    std::vector<unsigned char> tmp;  // reduction result of one chunk
};

constexpr double BARRIER_TIMEOUT_S = 180.0;

size_t slot_bytes() {  // NBODY_SHM_SLOT_KB (tests: slots far smaller than the messages, so that every exchange is chunked) or NBODY_SHM_SLOT_MB
    if (const char* kb = std::getenv("NBODY_SHM_SLOT_KB")) {
        const long v = std::atol(kb);
        if (v > 0) return (size_t)v << 10;
    }
    const char* mb = std::getenv("NBODY_SHM_SLOT_MB");
    const long v = mb ? std::atol(mb) : 0;
    return (size_t)(v > 0 ? v : 32) << 20;
}

bool barrier(Comm* c) {
    Header* h = c->head;
    if (h->failed.load()) return false;
    const uint32_t gen = h->generation.load();
    if (h->arrived.fetch_add(1) + 1 == (uint32_t)c->world) {
        h->arrived.store(0);
        h->generation.store(gen + 1);
        return true;
    }
    const auto t0 = std::chrono::steady_clock::now();
    for (uint64_t spins = 0; h->generation.load() == gen; ++spins) {
        if (h->failed.load()) return false;
        if ((spins & 1023) == 1023) {
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > BARRIER_TIMEOUT_S) {
                h->failed.store(1);
                return false;
            }
            usleep(50);
        } else {
            sched_yield();
        }
    }
    return true;
}

size_t type_size(ncclDataType_t t) {
    switch (t) {
        case ncclInt8: case ncclUint8: return 1;
        case ncclFloat16: case ncclBfloat16: return 2;
        case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
        case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
        default: return 0;
    }
}

ncclResult_t fail(Comm* c) {
    if (c && c->head) c->head->failed.store(1);
    return ncclSystemError;
}

#define HIPCHK(c, call)                            \
    do {                                           \
        if ((call) != hipSuccess) return fail(c);  \
    } while (0)

template <class T, class Op>
void reduce_chunk(Comm* c, size_t len, Op op) {
    const size_t cnt = len / sizeof(T);
    T* out = reinterpret_cast<T*>(c->tmp.data());
    std::memcpy(out, c->slots, len);  // rank 0 first, then the others in rank order
    for (int r = 1; r < c->world; ++r) {
        const T* in = reinterpret_cast<const T*>(c->slots + (size_t)r * c->slot);
        for (size_t i = 0; i < cnt; ++i) out[i] = op(out[i], in[i]);
    }
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    static std::atomic<unsigned> counter{0};
    std::memset(id, 0, sizeof *id);
    const auto now = std::chrono::steady_clock::now().time_since_epoch().count();
    std::snprintf(id->internal, sizeof id->internal, "/nbody_shm_%ld_%u_%llx", (long)getpid(), counter.fetch_add(1),
                  (unsigned long long)now);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
    if (!comm || nranks < 1 || rank < 0 || rank >= nranks || id.internal[0] != '/') return ncclInvalidArgument;
    Comm* c = new Comm;
    c->rank = rank;
    c->world = nranks;
    c->slot = slot_bytes();
    c->mapped = sizeof(Header) + (size_t)nranks * c->slot;
    // every rank opens (and, if it is first, creates) the segment; a fresh segment is all zeros,
    // which is the barrier's initial state
    const int fd = shm_open(id.internal, O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)c->mapped) != 0) {
        if (fd >= 0) close(fd);
        delete c;
        return ncclSystemError;
    }
    void* p = mmap(nullptr, c->mapped, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) {
        delete c;
        return ncclSystemError;
    }
    c->head = static_cast<Header*>(p);
    c->slots = static_cast<unsigned char*>(p) + sizeof(Header);
    c->tmp.resize(c->slot);
    const bool ok = barrier(c);
    if (rank == 0) shm_unlink(id.internal);  // everyone has it mapped: the name can go (the memory goes with the last unmap)
    if (!ok) {
        munmap(p, c->mapped);
        delete c;
        return ncclSystemError;
    }
    *comm = reinterpret_cast<ncclComm_t>(c);
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    Comm* c = reinterpret_cast<Comm*>(comm);
    if (!c) return ncclSuccess;
    if (c->head) munmap(c->head, c->mapped);
    delete c;
    return ncclSuccess;
}

const char* ncclGetErrorString(ncclResult_t r) {
    switch (r) {
        case ncclSuccess: return "no error";
        case ncclInvalidArgument: return "invalid argument (shared-memory test transport)";
        case ncclSystemError: return "a rank failed or timed out (shared-memory test transport)";
        default: return "error (shared-memory test transport)";
    }
}

ncclResult_t ncclGroupStart() { return ncclSuccess; }
ncclResult_t ncclGroupEnd() { return ncclSuccess; }

ncclResult_t ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm,
                           hipStream_t stream) {
    Comm* c = reinterpret_cast<Comm*>(comm);
    const size_t ts = type_size(datatype);
    if (!c || !ts) return ncclInvalidArgument;
    const size_t bytes = sendcount * ts;
    const unsigned char* src = static_cast<const unsigned char*>(sendbuff);
    unsigned char* dst = static_cast<unsigned char*>(recvbuff);
    HIPCHK(c, hipStreamSynchronize(stream));
    for (size_t off = 0; off < bytes || off == 0; off += c->slot) {
        const size_t len = bytes - off < c->slot ? bytes - off : c->slot;
        if (len) HIPCHK(c, hipMemcpyAsync(c->slots + (size_t)c->rank * c->slot, src + off, len, hipMemcpyDeviceToHost, stream));
        HIPCHK(c, hipStreamSynchronize(stream));
        if (!barrier(c)) return ncclSystemError;
        for (int r = 0; r < c->world && len; ++r)  // (the own part too: recvbuff may or may not alias sendbuff)
            HIPCHK(c, hipMemcpyAsync(dst + (size_t)r * bytes + off, c->slots + (size_t)r * c->slot, len, hipMemcpyHostToDevice, stream));
        HIPCHK(c, hipStreamSynchronize(stream));
        if (!barrier(c)) return ncclSystemError;
        if (bytes == 0) break;
    }
    return ncclSuccess;
}

ncclResult_t ncclBroadcast(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype, int root, ncclComm_t comm,
                           hipStream_t stream) {
    Comm* c = reinterpret_cast<Comm*>(comm);
    const size_t ts = type_size(datatype);
    if (!c || !ts || root < 0 || root >= c->world) return ncclInvalidArgument;
    const size_t bytes = count * ts;
    const unsigned char* src = static_cast<const unsigned char*>(sendbuff);
    unsigned char* dst = static_cast<unsigned char*>(recvbuff);
    HIPCHK(c, hipStreamSynchronize(stream));
    for (size_t off = 0; off < bytes || off == 0; off += c->slot) {
        const size_t len = bytes - off < c->slot ? bytes - off : c->slot;
        if (c->rank == root && len) HIPCHK(c, hipMemcpyAsync(c->slots, src + off, len, hipMemcpyDeviceToHost, stream));
        HIPCHK(c, hipStreamSynchronize(stream));
        if (!barrier(c)) return ncclSystemError;
        if (len && (c->rank != root || dst != src))
            HIPCHK(c, hipMemcpyAsync(dst + off, c->slots, len, hipMemcpyHostToDevice, stream));
        HIPCHK(c, hipStreamSynchronize(stream));
        if (!barrier(c)) return ncclSystemError;
        if (bytes == 0) break;
    }
    return ncclSuccess;
}

ncclResult_t ncclAllReduce(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op,
                           ncclComm_t comm, hipStream_t stream) {
    Comm* c = reinterpret_cast<Comm*>(comm);
    const size_t ts = type_size(datatype);
    if (!c || !ts) return ncclInvalidArgument;
    const bool f32_sum = datatype == ncclFloat32 && op == ncclSum, i32_min = datatype == ncclInt32 && op == ncclMin;
    if (!f32_sum && !i32_min) return ncclInvalidArgument;  // the two reductions the library issues
    const size_t bytes = count * ts;
    const unsigned char* src = static_cast<const unsigned char*>(sendbuff);
    unsigned char* dst = static_cast<unsigned char*>(recvbuff);
    HIPCHK(c, hipStreamSynchronize(stream));
    for (size_t off = 0; off < bytes || off == 0; off += c->slot) {
        const size_t len = bytes - off < c->slot ? bytes - off : c->slot;
        if (len) HIPCHK(c, hipMemcpyAsync(c->slots + (size_t)c->rank * c->slot, src + off, len, hipMemcpyDeviceToHost, stream));
        HIPCHK(c, hipStreamSynchronize(stream));
        if (!barrier(c)) return ncclSystemError;
        if (len) {
            if (f32_sum) reduce_chunk<float>(c, len, [](float a, float b) { return a + b; });
            else reduce_chunk<int32_t>(c, len, [](int32_t a, int32_t b) { return a < b ? a : b; });
            HIPCHK(c, hipMemcpyAsync(dst + off, c->tmp.data(), len, hipMemcpyHostToDevice, stream));
        }
        HIPCHK(c, hipStreamSynchronize(stream));
        if (!barrier(c)) return ncclSystemError;
        if (bytes == 0) break;
    }
    return ncclSuccess;
}

}  // extern "C"
