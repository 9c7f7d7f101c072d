// Internal state of an nbody_ctx and helpers shared by the translation units of libnbody_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>
#include "../../include/nbody_hip.h"

struct nbody_ctx {
    nbody_config cfg;
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int num_cus = 256;
    std::string err;

    uint64_t n = 0;        // all bodies
    uint64_t i_begin = 0;  // this context's targets
    uint64_t i_count = 0;
    bool uploaded = false;
    bool uniform_mass = false;  // every body has mass mass0 (checked at upload)
    float mass0 = 0.0f;

    // Body state in HBM.  Two sets (the reference's bind groups A and B, src/nbody_gpu.rs:145-288);
    // `cur` is the set the next step reads.  posm[s][i] = {x, y, z, mass_i}.
    float4* posm[2] = {nullptr, nullptr};
    float4* vel[2] = {nullptr, nullptr};
    float4* acc[2] = {nullptr, nullptr};
    float* masses = nullptr;
    int cur = 0;

    // All-pairs scratch: per source-slice partial accelerations [slices][i_count].
    float4* acc_part = nullptr;
    uint64_t acc_part_slices = 0;
    float4* acc_tmp = nullptr;  // [n] forces-only output / finalize target; NBODY_BUF_ACC_PARTIAL
    float4* stage = nullptr;    // [n] staging of positions on their way to / from the host (lazily allocated)
    int pending_forces = 0;     // acc_tmp holds forces awaiting integration: 1 nbody_allpairs_partial, 2 nbody_bh_forces_ranked
    float4* acc_body = nullptr; // [n] accelerations brought back from sorted-rank to body order (lazily allocated)

    // Octree state (octree.hip)
    struct Octree* tree = nullptr;
    // symmetric all-pairs plan and scratch (allpairs_sym.hip)
    struct SymPlan* sym = nullptr;
    // RCCL communicator and the ranks' ranges (comm.hip); null on a single-GPU context
    struct NbComm* comm = nullptr;

    // energy scratch
    double* energy_dev = nullptr;

    nbody_counters_t counters{};
    bool counting = false;

    bool leapfrog_started = false;  // NBODY_INT_LEAPFROG: a step has been taken since the last upload
    // kernel timing (HIP events on `stream`)
    bool timing = false;
    std::vector<hipEvent_t> ev[3];  // 0 all-pairs force kernel, 1 traversal kernel, 2 octree build (all its launches):
    size_t ev_used[3] = {0, 0, 0};  // a pool of event pairs, created when timing is switched on and reused
    unsigned long long* clk_dev = nullptr;  // {shader cycles, 100 MHz ticks} of one workgroup of the last pair-kernel launch
    double time_acc_ms[3] = {0, 0, 0};
    uint64_t time_launches[3] = {0, 0, 0};
};

extern thread_local std::string g_nbody_create_error;

#define NB_HIP(ctx, call)                                                                          \
    do {                                                                                           \
        hipError_t _e = (call);                                                                    \
        if (_e != hipSuccess) {                                                                    \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(_e);                        \
            return NBODY_ERR_HIP;                                                                  \
        }                                                                                          \
    } while (0)

static inline int nb_fail(nbody_ctx* ctx, int code, const std::string& msg) {
    if (ctx) ctx->err = msg;
    return code;
}

// all-pairs (allpairs.hip)
int nb_allpairs_forces(nbody_ctx* ctx, const float4* posm_in, const float4* vel_in, float4* acc_out);
int nb_integrate(nbody_ctx* ctx, const float4* acc_new);
int nb_integrate_clamped(nbody_ctx* ctx, const float4* acc_new);
int nb_energy(nbody_ctx* ctx, double* kinetic, double* potential);
void nb_time_begin(nbody_ctx* ctx, int which);
void nb_time_end(nbody_ctx* ctx, int which);
int nb_time_collect(nbody_ctx* ctx, int which);
int nb_time_reserve(nbody_ctx* ctx, size_t pairs);  // event pairs ready for each kind of launch

const char* nb_noalloc_last_error();
void nb_noalloc_clear_error();

// symmetric all-pairs (allpairs_sym.hip)
int nb_allpairs_forces_sym(nbody_ctx* ctx, const float4* posm_in, float4* acc_out, uint32_t seg_tiles, int tpl,
                           uint32_t part, uint32_t nparts);
int nb_allpairs_partial(nbody_ctx* ctx, uint32_t part, uint32_t nparts);
uint64_t nb_sym_scratch_bytes(uint64_t n, uint32_t seg_tiles, uint32_t sb);
void nb_sym_destroy(nbody_ctx* ctx);
// diag_out: 256 * tpl / NB_DIAG_TILE slices of n entries each, to be summed
constexpr int NB_DIAG_TILE = 512;
int nb_allpairs_diag(nbody_ctx* ctx, const float4* posm_in, float4* diag_out, int tpl);

// multi-GPU exchange (comm.hip)
int nb_comm_unique_id(void* id128);
int nb_comm_init(nbody_ctx* ctx, const void* id128, int rank, int world);
void nb_comm_destroy(nbody_ctx* ctx);
int nb_comm_gather_positions(nbody_ctx* ctx);   // every rank's own targets of posm[cur] to every rank
int nb_comm_allreduce_partial(nbody_ctx* ctx);  // acc_tmp summed over the ranks
int nb_comm_gather_ranked(nbody_ctx* ctx);      // acc_tmp by sorted rank: every rank's run to every rank
void nb_comm_run(const nbody_ctx* ctx, int k, uint64_t* begin, uint64_t* count);
void nb_sorted_run(uint64_t n, uint64_t world, uint64_t k, uint64_t* begin, uint64_t* count);
enum { NB_PLAN_INVALID = 0, NB_PLAN_ALLGATHER = 1, NB_PLAN_BROADCASTS = 2, NB_PLAN_REPLICATED = 3 };
int nb_exchange_plan(uint64_t n, int world, const uint64_t* begin, const uint64_t* count);
int nb_comm_pair_path(nbody_ctx* ctx, bool fits_here, bool* use_pair);  // collective: asked once, then remembered
bool nb_allpairs_pair_fits(nbody_ctx* ctx);
int nb_comm_rank(const nbody_ctx* ctx);
int nb_comm_allgather_inplace(nbody_ctx* ctx, void* buf, size_t bytes_per_rank);
struct NbBroadcast {
    const void* send;
    void* recv;
    size_t bytes;
    int root;
};
int nb_comm_broadcasts(nbody_ctx* ctx, const NbBroadcast* items, int count);
int nb_comm_world(const nbody_ctx* ctx);

// octree (octree.hip)
int nb_tree_create(nbody_ctx* ctx);
void nb_tree_destroy(nbody_ctx* ctx);
int nb_tree_build(nbody_ctx* ctx, const float4* posm_in);
int nb_tree_walk(nbody_ctx* ctx, const float4* posm_in, const float4* vel_in, float4* acc_out);
int nb_tree_walk_ranked(nbody_ctx* ctx, const float4* posm_in, const float4* vel_in, float4* acc_ranked, uint64_t rank_begin,
                        uint64_t rank_count);
int nb_tree_check(nbody_ctx* ctx);
int nb_tree_unrank(nbody_ctx* ctx, const float4* acc_ranked, float4* acc_out);
int nb_tree_download(nbody_ctx* ctx, nbody_octree_node* host_nodes, uint64_t cap, uint64_t* n_nodes);
int nb_tree_upload(nbody_ctx* ctx, const nbody_octree_node* host_nodes, uint64_t n_nodes);
int nb_tree_keys(nbody_ctx* ctx, uint64_t* keys, uint8_t* depth, uint32_t* order);
int nb_tree_set_target_order(nbody_ctx* ctx, const uint32_t* host_order, uint64_t count);
int nb_tree_set_domains(nbody_ctx* ctx, uint32_t domains, uint32_t top_levels);
int nb_tree_domain_stats(nbody_ctx* ctx, nbody_domain_stats_t* out);
int nb_tree_set_domain_view(nbody_ctx* ctx, int domain);
bool nb_tree_distributed(const nbody_ctx* ctx);  // domains set, a communicator attached: nbody_step_bh exchanges essential trees
// One distributed Barnes-Hut force evaluation: this rank builds its own domain's tree, the ranks exchange
// the parts of their trees the others' targets can reach and the accelerations; acc_body_out: all bodies.
int nb_tree_forces_distributed(nbody_ctx* ctx, const float4* posm_in, const float4* vel_in, float4* acc_body_out);
void nb_tree_forget_order(nbody_ctx* ctx);  // new bodies: the kept grouping of the traversal's targets no longer fits
int nb_tree_device_nodes(nbody_ctx* ctx, void** ptr, size_t* bytes);
