/* nbody_hip.h -- C ABI of libnbody_hip.so: the gravity step of JMS55/nbody on MI355X (gfx950).
 *
 * The reference exposes no library, plugin or FFI interface (Cargo.toml:6-16 declares three
 * binaries).  Its only stable contract for this path is the GPU buffer layout its compute passes
 * bind and the per-frame step sequence.  Each entry point below names the reference code it
 * replaces.  Buffers keep the reference's layouts:
 *   masses                f32[N]                       src/nbody.wgsl:1
 *   positions/velocities/accelerations
 *                         f32[N][4], xyz + 4-byte pad  src/nbody.wgsl:3-8 (vec3 stride 16)
 *   octree                96-byte nodes                src/nbodybh.wgsl:1-9,26
 *
 * Conventions: every function returns 0 on success and a negative nbody_status on failure;
 * nbody_last_error(ctx) returns a message.  Nothing aborts.  A context is not thread-safe (the
 * reference steps from one thread).  Steps are enqueued asynchronously on the context's HIP
 * stream; nbody_download*, nbody_sync, nbody_energy and nbody_counters block.  All pointers named
 * host_* are host memory; pointers returned by nbody_device_buffer are device memory.
 */
#ifndef NBODY_HIP_H
#define NBODY_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct nbody_ctx nbody_ctx;

enum nbody_status {
    NBODY_OK = 0,
    NBODY_ERR_INVALID = -1,   /* bad argument */
    NBODY_ERR_HIP = -2,       /* a HIP runtime call failed */
    NBODY_ERR_NO_DEVICE = -3, /* no usable gfx950 device */
    NBODY_ERR_STATE = -4,     /* call made in the wrong state (e.g. step before upload) */
    NBODY_ERR_CAPACITY = -5,  /* a fixed capacity was exceeded (tree nodes, traversal stack) */
    NBODY_ERR_NOMEM = -6
};

/* Force law of one source term on a target at pos with velocity vel; d = p_j - pos. */
enum nbody_law {
    /* src/nbody_cpu.rs:337-340: G*m_j*d / (d.d)^3, self skipped, no softening. */
    NBODY_LAW_CPU_R6 = 0,
    /* src/nbody.wgsl:54-81: G*m_j*d / (|d|^2+eps2)^1.5 * acos(vel.d/(|d||vel|+1))^0.15, self included. */
    NBODY_LAW_WGSL_BIAS = 1,
    /* The WGSL law without the bias factor (Plummer-softened Newtonian gravity). */
    NBODY_LAW_PLUMMER = 2
};

enum nbody_integrator {
    /* src/nbody_cpu.rs:345-348,358-360: acc2=a; vel2+=a; pos2+=vel2; swap sets (two interleaved states). */
    NBODY_INT_CPU_EULER_XBUF = 0,
    /* src/nbody.wgsl:132-138: pos+=vel*dt+0.5*acc_in*dt^2; vel+=0.5*(acc_in+acc)*dt.  (The position
     * update uses the PREVIOUS step's acceleration: first order in the energy, about 6 % drift over
     * 1,000 steps of a Plummer sphere.) */
    NBODY_INT_WGSL_VERLET = 1,
    /* Opt-in, no counterpart in the reference: kick-drift leapfrog, vel+=acc*dt (dt/2 on the first
     * step after an upload); pos+=vel*dt.  Time-centred and symplectic; the stored velocity is the
     * half-step one. */
    NBODY_INT_LEAPFROG = 2
};

/* Node types of src/octree_maxdepth.rs:8-11. */
enum nbody_node_type { NBODY_NODE_DUMMY = 0, NBODY_NODE_LEAFBODY = 1, NBODY_NODE_LEAFLIST = 2, NBODY_NODE_INTERIOR = 3 };

/* src/octree_maxdepth.rs:13-23 in the storage layout of src/nbodybh.wgsl:1-9 (96 bytes).
 * child_indices[k]==0 means "no child" (the root, index 0, is never a child).  For a LEAFLIST
 * node child_indices[0] is the first list member and child_indices[1] the member count
 * (src/octree_maxdepth.rs:53-54). */
typedef struct nbody_octree_node {
    float center_of_mass[3];
    float _pad0;
    float pos_min[3];
    float _pad1;
    float pos_max[3];
    float range; /* longest side of the bodies' bounding box (src/octree_maxdepth.rs:101) */
    float total_mass;
    uint32_t child_indices[8];
    uint32_t node_type;
    uint32_t depth; /* padding in the reference (max_depth in src/octree.rs:14); here the node's depth */
    uint32_t _pad2;
} nbody_octree_node;

/* Compile-time constants of the reference gathered in one struct.  nbody_config_default fills the
 * reference's values: G, dt, eps2 (src/nbody.wgsl:13-15), theta (src/nbodybh.wgsl:61),
 * world (src/nbody_gpu.rs:15), max_depth (src/octree_maxdepth.rs:7). */
typedef struct nbody_config {
    uint64_t n_bodies;     /* N over all shards */
    int32_t law;           /* enum nbody_law */
    int32_t integrator;    /* enum nbody_integrator */
    float G;               /* 0.0066743 (CPU_R6 always uses this literal, as the reference does) */
    float dt;              /* 0.1 */
    float eps2;            /* 1.0 */
    float theta;           /* 0.5 */
    float world;           /* 250.0: octree root cube [0,world]^3 and the Barnes-Hut clamp box */
    uint32_t max_depth;    /* 1..16: depth cap of src/octree_maxdepth.rs.  0: the uncapped builder of src/octree.rs
                              (dormant in the reference), reproduced while no two bodies share a level-16
                              cell (else the build fails) and no body has mass 0 (nbody_upload refuses:
                              src/octree.rs:49 takes a massless node for an empty one): no leaf lists;
                              nbody_download_octree then writes
                              is_leaf at offset 84 and, on the root, max_depth at offset 88 (src/octree.rs:5-15,36) */
    int32_t leaflist_mode; /* 0: literal src/octree_maxdepth.rs:123-133 (a k-body list holds the k-1 prefix
                              aggregates); 1: every body of the cell is a list member */
    uint64_t shard_begin;  /* first target body integrated by this context */
    uint64_t shard_count;  /* number of target bodies integrated by this context (0 => all) */
    int32_t device;        /* HIP device ordinal, -1 = current device */
    int32_t quadrupole;    /* opt-in, no counterpart in the reference: 1 adds each accepted cell's quadrupole
                              moment to its monopole term in the Barnes-Hut traversal (PLUMMER law only; the
                              acceptance test is unchanged).  0: the reference's monopole-only traversal */
    void* stream;          /* hipStream_t to enqueue on, NULL = a stream owned by the context */
} nbody_config;

/* Device buffers a host may need to reach (for the per-step all-gather between shards). */
enum nbody_buffer {
    NBODY_BUF_POSM = 0,  /* current positions, f32[N][4]; the pad lane carries the body's mass */
    NBODY_BUF_VEL = 1,   /* current velocities f32[N][4] */
    NBODY_BUF_ACC = 2,   /* current accelerations f32[N][4] */
    NBODY_BUF_OCTREE = 3, /* nbody_octree_node[n_nodes], exported on request */
    NBODY_BUF_ACC_PARTIAL = 4 /* f32[N][4] exchanged between GPUs: written by nbody_allpairs_partial (read by
                                 nbody_integrate_partial) or by nbody_bh_forces_ranked (read by nbody_integrate_ranked).
                                 Every step and every nbody_accel_* call also writes it: an integrate_* call
                                 that does not follow its own force call directly returns NBODY_ERR_STATE.
                                 Uploads and downloads do not touch it. */
};

/* Work done since nbody_create or the last nbody_counters_reset; for roofline arithmetic. */
typedef struct nbody_counters_t {
    uint64_t steps_allpairs;
    uint64_t steps_bh;
    uint64_t interactions_allpairs; /* body-body terms evaluated */
    uint64_t bh_node_visits;        /* per-body node visits of the last counted walk */
    uint64_t bh_leaf_interactions;
    uint64_t bh_aggregate_interactions;
    uint64_t bh_wave_node_visits;   /* per-wavefront node visits of the last counted walk */
    uint64_t octree_nodes;          /* nodes of the last built tree */
    uint64_t bh_max_stack;          /* deepest traversal stack of the last counted walk */
    uint64_t allpairs_path;         /* kernel of the last all-pairs force evaluation: 1 = one evaluation per ordered
                                       pair, 2 = the pair kernel (one per unordered pair; PLUMMER law, N >= 65,536,
                                       scratch permitting), 0 = none yet */
    uint64_t reserved[6];
} nbody_counters_t;

const char* nbody_version(void);
/* Message of the last failing call on ctx (or on creation when ctx is NULL). */
const char* nbody_last_error(const nbody_ctx* ctx);

void nbody_config_default(nbody_config* cfg);

/* Replaces device/buffer/pipeline setup, src/nbody_gpu.rs:83-309. */
int nbody_create(nbody_ctx** out, const nbody_config* cfg);
void nbody_destroy(nbody_ctx* ctx);

/* Replaces the create_buffer_init calls of src/nbody_gpu.rs:85-143.  host_pos16 is f32[N][4].
 * host_vel16 / host_acc16 may be NULL => zeros (the reference's initial state, :122-137). */
int nbody_upload(nbody_ctx* ctx, const float* host_masses, const float* host_pos16, const float* host_vel16,
                 const float* host_acc16);
/* Second buffer set of the CPU binary (src/nbody_cpu.rs:58-60); NULL pointers => zeros, which is
 * also what nbody_upload leaves there. Only read by NBODY_INT_CPU_EULER_XBUF. */
int nbody_upload_alt(nbody_ctx* ctx, const float* host_pos16, const float* host_vel16, const float* host_acc16);

/* Replaces one MainEventsCleared tick of src/nbody_gpu.rs:418-451 (or of src/nbody_cpu.rs:319-360
 * with NBODY_INT_CPU_EULER_XBUF), nsteps times: forces over all N sources for this context's
 * targets, integration, swap of the in/out sets.  With shard_count < N only nsteps == 1 is
 * accepted unless the context has a communicator (nbody_comm_init): the host must all-gather
 * NBODY_BUF_POSM between steps. */
int nbody_step_allpairs(nbody_ctx* ctx, int nsteps);
/* Replaces one tick of src/nbody_gpu_bh.rs:445-527: octree build from the current positions,
 * traversal, integration with clamp to [0,world]^3, swap.  The laws are those of
 * src/nbodybh.wgsl (NBODY_LAW_CPU_R6 and NBODY_INT_CPU_EULER_XBUF are refused: NBODY_ERR_INVALID). */
int nbody_step_bh(nbody_ctx* ctx, int nsteps);

/* Multi-GPU all-pairs with ONE evaluation per unordered pair (PLUMMER law, Verlet).  The pair
 * tiles of all N bodies are dealt out over nparts parts (one per GPU): part `part` writes its
 * contribution to EVERY body's acceleration into NBODY_BUF_ACC_PARTIAL.  The host sums that
 * buffer over the parts (RCCL all-reduce), then nbody_integrate_partial advances this context's
 * own targets with the summed accelerations and swaps the sets; the host then all-gathers
 * NBODY_BUF_POSM as after nbody_step_allpairs.  No counterpart in the reference (single device). */
int nbody_allpairs_partial(nbody_ctx* ctx, uint32_t part, uint32_t nparts);
int nbody_integrate_partial(nbody_ctx* ctx);

/* Multi-GPU Barnes-Hut with REPLICATED state: every GPU holds all N bodies (a context without a
 * shard range) and builds the same octree; the traversal is split by SORTED RANK, so that each
 * GPU walks a spatially compact run of the Morton order (src/nbodybh.wgsl:63-161 per target).
 * nbody_bh_forces_ranked builds the octree from the current positions and writes the
 * accelerations of the bodies at ranks [rank_begin, rank_begin + rank_count) to
 * NBODY_BUF_ACC_PARTIAL[rank].  The host all-gathers that buffer (16 bytes per body); then
 * nbody_integrate_ranked brings it back to body order and advances ALL bodies (integration with
 * clamp, src/nbodybh.wgsl:163-169) and swaps the sets.  Every GPU computes bit-identical states,
 * so positions need no exchange.  No counterpart in the reference (single device). */
int nbody_bh_forces_ranked(nbody_ctx* ctx, uint64_t rank_begin, uint64_t rank_count);
int nbody_integrate_ranked(nbody_ctx* ctx);

/* Multi-GPU inside the library, one process per GPU.  No counterpart in the reference (single
 * device).  nbody_comm_unique_id fills 128 bytes (an RCCL unique id) on ONE rank; the host hands
 * them to the other ranks by its own means (a file, MPI, torch.distributed ...).  nbody_comm_init
 * (collective: every rank calls it with the same id) attaches an RCCL communicator to the context;
 * RCCL is loaded then: the librccl beside the HIP runtime this library is bound to (a process that
 * imported PyTorch first shares PyTorch's copies of both).  The ranks' contexts either have
 * shard ranges that tile [0, n_bodies) in rank order, or all hold every body (no shard range).
 * With a communicator nbody_step_allpairs and nbody_step_bh are complete steps, nsteps of them,
 * enqueued on the context's stream with their collectives:
 *   nbody_step_allpairs  own targets' forces and integration, then every rank's positions gathered
 *                        in place; for the PLUMMER law from N = 65,536 the pair kernel's work is
 *                        dealt over the ranks (nbody_allpairs_partial), the contributions
 *                        all-reduced, own targets integrated, positions gathered
 *   nbody_step_bh        (contexts without a shard range) octree build on every rank, traversal
 *                        of this rank's run [n k / world, n (k + 1) / world) of the sorted order,
 *                        accelerations gathered by rank, all bodies integrated on every rank
 * nbody_destroy releases the communicator. */
int nbody_comm_unique_id(void* id_out128);
int nbody_comm_init(nbody_ctx* ctx, const void* unique_id128, int rank, int world);
/* The environment variable NBODY_RCCL_LIB, when set, names the library nbody_comm_unique_id and
 * nbody_comm_init load in RCCL's place (it must export ncclGetUniqueId, ncclCommInitRank,
 * ncclCommDestroy, ncclAllGather, ncclAllReduce, ncclBroadcast, ncclGroupStart, ncclGroupEnd and
 * ncclGetErrorString).
 * With a communicator, a failure on ONE rank between two collectives of a step (out of memory,
 * a HIP error) leaves the other ranks waiting in the next collective: it ends the run.  The one
 * decision that depends on a rank's own free memory -- whether the pair kernel's scratch fits -- is
 * therefore taken by all ranks together (an all-reduce on the first nbody_step_allpairs).
 *
 * Plain arithmetic behind the exchange, callable without a device or a context (hosts that run the
 * exchange themselves, and tests).  nbody_comm_sorted_run: the run [*begin, *begin + *count) of the
 * sorted order whose targets `rank` of `world` traverses in nbody_step_bh: boundaries at
 * n_bodies k / world rounded down to a multiple of 64, the last run ends at n_bodies; runs are
 * empty where n_bodies < 64 world.  nbody_comm_exchange_plan: how ranges begin[k], count[k] (k <
 * world) are exchanged: returns 1 = they tile [0, n_bodies) in rank order with equal counts (one
 * in-place all-gather: begin[k] == k count[0]), 2 = they tile it with unequal or empty counts (one
 * broadcast per non-empty rank), 3 = every rank holds all bodies (nothing to exchange), or
 * NBODY_ERR_INVALID (gaps, overlaps, out of order), which nbody_comm_init refuses. */
int nbody_comm_sorted_run(uint64_t n_bodies, int world, int rank, uint64_t* begin, uint64_t* count);
int nbody_comm_exchange_plan(uint64_t n_bodies, int world, const uint64_t* begin, const uint64_t* count);

/* Forces only (no integration, no swap), for parity checks against src/nbody.wgsl:50-116 and
 * src/nbodybh.wgsl:63-161.  host_acc16_out is f32[N][4]; only this context's targets are written. */
int nbody_accel_allpairs(nbody_ctx* ctx, float* host_acc16_out);
/* No counterpart in the reference.  For callers that cannot allocate device memory (code under HIP
 * graph capture, tight serving loops, harnesses that forbid hipMalloc): the accelerations of
 * nbody_accel_allpairs, computed by the same kernels, without a context and without any
 * allocation.  dev_posm is f32[n][4] = {x, y, z, mass} and dev_vel16 f32[n][4] (read by
 * NBODY_LAW_WGSL_BIAS only, else may be NULL), both device memory on the CURRENT device;
 * dev_acc16_out, f32[n][4] device memory, receives G * sum (pad lane 0).  Enqueued on `stream`
 * (a hipStream_t, NULL = the null stream); does not block.  The per-slice partial sums live in a
 * fixed 16 MiB device array inside the library (one per GPU, held until the process ends), which
 * limits n_bodies to 131,072: larger inputs return NBODY_ERR_CAPACITY.  The array also caps the
 * number of source slices (16 MiB / (16 B n_bodies)), and the pair kernel is never used here, so
 * above N = 16,384 (and for the PLUMMER law from N = 65,536) the sums are split differently from
 * nbody_accel_allpairs': the two agree to rounding (2e-6 relative), bit for bit only where both
 * use the same slicing.  Calls that overlap on one GPU (two streams, two threads) share that array
 * and race; calls on one stream are ordered.  The FIRST call on a device looks up the array's
 * address and the device's size (hipGetSymbolAddress, hipDeviceGetAttribute): make it outside a
 * graph capture.  nbody_last_error(NULL) has the message of the last failure of a call made
 * without a context; a successful call clears it. */
int nbody_accel_allpairs_noalloc(const float* dev_posm, const float* dev_vel16, uint64_t n_bodies, int law, float G,
                                 float eps2, float* dev_acc16_out, void* stream);
/* Uses the tree currently held by the context (built or uploaded). */
int nbody_accel_bh(nbody_ctx* ctx, float* host_acc16_out);

/* Replaces OctreeNode::new_tree, src/octree_maxdepth.rs:26-61 (called at src/nbody_gpu_bh.rs:452). */
int nbody_build_octree(nbody_ctx* ctx);
/* Writes the current tree in the 96-byte layout.  cap_nodes: capacity of host_nodes; *n_nodes
 * receives the node count (also when host_nodes is NULL, to size the buffer). */
int nbody_download_octree(nbody_ctx* ctx, nbody_octree_node* host_nodes, uint64_t cap_nodes, uint64_t* n_nodes);
/* Replaces the per-frame octree upload of src/nbody_gpu_bh.rs:454-469: traversals use this tree
 * (any child numbering, e.g. the reference's allocation order) until the next build. */
int nbody_upload_octree(nbody_ctx* ctx, const nbody_octree_node* host_nodes, uint64_t n_nodes);
/* Per body: the 3*max_depth-bit Morton key of its depth-max_depth cell (octant bits x->4, y->2,
 * z->1 per level, src/octree_maxdepth.rs:210-222), the depth of its leaf in the built tree, and
 * the sorted order (body index of rank r).  Any pointer may be NULL.  Requires a built tree. */
int nbody_morton_keys(nbody_ctx* ctx, uint64_t* host_keys, uint8_t* host_leaf_depth, uint32_t* host_order);

/* Tuning hook, no counterpart in the reference (whose traversal is one thread per body).  The
 * traversal walks 64 targets per wavefront; by default those are 64 consecutive bodies of the
 * library's own target order.  host_order (count entries; body indices, every body exactly once,
 * 0xffffffff = a lane left empty; count in [n_bodies, 4 n_bodies + 64]) replaces that grouping for
 * nbody_accel_bh / nbody_step_bh on an unsharded context, until it is reset with NULL.  The
 * accelerations do not depend on the grouping, bit for bit: a lane interacts with the same nodes in
 * the same order whoever shares its wavefront; only the time does. */
int nbody_bh_set_target_order(nbody_ctx* ctx, const uint32_t* host_order, uint64_t count);

/* Domain-decomposed octree build; no counterpart in the reference, which builds ONE tree on one host
 * every frame (src/nbody_gpu_bh.rs:452, src/octree_maxdepth.rs:26-61).  With domains >= 2 every build
 * of this (unsharded) context deals the bodies to `domains` domains -- runs of whole depth-top_levels
 * cells in Morton order holding about n_bodies / domains bodies each -- builds each domain's tree from
 * its own bodies alone (work in proportion to its share: what one GPU of several would do), and
 * rebuilds the cells above depth top_levels from the domains' depth-top_levels cells (the "top tree",
 * the only part every domain must hold).  The traversal runs over the merged tree, whose nodes are
 * those of the one tree over all bodies, bit for bit: accelerations, visit and interaction counts do
 * not change.  domains <= 32, 1 <= top_levels <= 6 and below max_depth; domains <= 1 switches back.
 * Not with max_depth 0 or quadrupoles; nbody_download_octree and leaf depths are not available while on.
 * nbody_bh_domain_stats (after a build; blocks): sizes and times of the last build, and -- after a
 * COUNTING traversal of all bodies (nbody_set_counting) -- how many nodes of OTHER domains' trees each
 * domain's targets visited: the nodes it would have to import if every domain sat on its own GPU. */
typedef struct nbody_domain_stats_t {
    uint32_t domains, top_levels;
    uint64_t top_entries;        /* depth-top_levels cells and lone bodies handed to the top tree */
    uint64_t top_nodes;          /* nodes of the top tree (replicated on every domain) */
    uint64_t merged_nodes;       /* nodes of the merged tree = those of the one tree over all bodies */
    uint64_t top_nodes_visited;  /* top-tree nodes some target visited (last counting traversal) */
    double partition_ms;         /* dealing the bodies to the domains */
    double merge_ms;             /* nodes into the merged array, merged sorted order, top tree */
    uint64_t bodies[32];
    uint64_t nodes[32];          /* nodes of each domain's own tree */
    uint64_t imported_nodes[32]; /* per domain: nodes of other domains' trees its targets visited */
    double build_ms[32];         /* each domain's own build (gather of its bodies + the ordinary builder) */
    int32_t view_domain;         /* nbody_bh_set_domain_view: the domain the merged tree was assembled for, else -1 */
    int32_t distributed;         /* 1: the last build was this rank's part of a distributed step (nbody_step_bh) */
    uint64_t exported_nodes[32]; /* essential trees: nodes a domain hands to the others (view: every other domain's;
                                    distributed: every rank's, this rank's own included) */
    double export_ms[32];        /* choosing, numbering and packing them (the exports made by THIS process) */
} nbody_domain_stats_t;
int nbody_bh_set_domains(nbody_ctx* ctx, uint32_t domains, uint32_t top_levels);
int nbody_bh_domain_stats(nbody_ctx* ctx, nbody_domain_stats_t* out);
/* Essential trees (no counterpart in the reference).  A domain on a GPU of its own does not need the
 * other domains' whole trees: of each it needs the nodes its targets can reach, i.e. the children of
 * every node that some point of its bodies' bounding box would refuse under the acceptance test of
 * src/nbodybh.wgsl:104-106 -- decided by the exporting domain with the traversal's own thresholds and
 * a margin far above rounding, so what is sent is a superset of what is visited and the accelerations
 * stay those of the one tree, bit for bit.
 *   nbody_bh_set_domain_view(ctx, d), d in [0, domains): one-GPU rehearsal.  Every later build still
 *   makes all domains' trees, but assembles the merged tree as the GPU of domain d would hold it: its
 *   own tree whole, of every other domain only its export.  nbody_accel_bh / nbody_step_bh then serve
 *   the bodies of domain d ONLY (the other bodies' accelerations are meaningless: their own subtrees
 *   are pruned); nbody_bh_domain_stats reports the exports.  d < 0 switches back.
 *   With a communicator attached (nbody_comm_init, world == domains) nbody_step_bh does the same for
 *   real: rank r builds the tree of domain r alone, the ranks exchange bounding boxes, exports and
 *   top-tree entries (grouped broadcasts), every rank traverses for its own domain's bodies, and the
 *   accelerations are exchanged domain by domain; all ranks then integrate all bodies, as in the
 *   replicated-tree step. */
int nbody_bh_set_domain_view(nbody_ctx* ctx, int32_t domain);

/* Replaces the position readback of src/nbody_gpu_bh.rs:496-516 (and extends it to all state).
 * Any pointer may be NULL.  The pad lane of host_pos16 is written as 0. */
int nbody_download(nbody_ctx* ctx, float* host_pos16, float* host_vel16, float* host_acc16);
int nbody_download_alt(nbody_ctx* ctx, float* host_pos16, float* host_vel16, float* host_acc16);
int nbody_sync(nbody_ctx* ctx);
/* Kinetic and softened potential energy of the current state, accumulated in f64 on the device
 * (all-pairs; O(N^2)).  No counterpart in the reference; used for the 1000-step drift check. */
int nbody_energy(nbody_ctx* ctx, double* kinetic, double* potential);
int nbody_counters(nbody_ctx* ctx, nbody_counters_t* out);
int nbody_counters_reset(nbody_ctx* ctx);
/* When on, the next nbody_accel_bh / nbody_step_bh uses the counting traversal (slower). */
int nbody_set_counting(nbody_ctx* ctx, int on);

/* Device pointer and size of one of the context's current buffers (enum nbody_buffer), for hosts
 * that exchange shards themselves (RCCL all-gather of NBODY_BUF_POSM).  No counterpart in the
 * reference (it is single-device). */
int nbody_device_buffer(nbody_ctx* ctx, int which, void** dev_ptr, size_t* bytes);
/* Copies the first `bytes` bytes of one of those buffers to the host (blocks). */
int nbody_read_buffer(nbody_ctx* ctx, int which, void* host_dst, size_t bytes);
/* The hipStream_t the context enqueues on. */
void* nbody_stream(nbody_ctx* ctx);
/* Average duration in milliseconds, over the launches since the last call, of: which = 0 the
 * all-pairs force kernel, 1 the Barnes-Hut traversal kernel, 2 one whole octree build (all its
 * kernels).  Measured with HIP events on the context's stream; *launches receives the count.  Blocks. */
int nbody_kernel_time(nbody_ctx* ctx, int which, double* avg_ms, uint64_t* launches);
int nbody_set_kernel_timing(nbody_ctx* ctx, int on);
/* The shader clock (MHz) one workgroup of the LAST pair-kernel launch (all-pairs, PLUMMER law,
 * N >= 65,536) ran at: its shader-cycle count over its 100 MHz real-time count, stamped around a
 * work item in the middle of the dispatch while kernel timing was on.  0 when there was no such
 * launch.  For pricing the FP32 peak at the clock the chip actually held.  Blocks.  No
 * counterpart in the reference. */
int nbody_kernel_clock(nbody_ctx* ctx, double* mhz);

/* Synthetic initial conditions (host code, no GPU).  Counter-based splitmix64 stream: draw k of
 * body i is mix64(seed + (64*i + k + 1) * 0x9E3779B97F4A7C15), u = (x >> 40) * 2^-24.
 * refcube follows src/nbody_gpu.rs:61-81: pos uniform in [world/5, 4*world/5)^3,
 * mass = U[0.5,8]*U[0.5,8], zero velocity.  first/count select a body range. */
int nbody_gen_refcube(uint64_t seed, uint64_t first, uint64_t count, float world, float* host_masses,
                      float* host_pos16);
/* Plummer sphere centred in the world cube, scale radius a, radius cut rmax, equal masses
 * total_mass/n_total.  host_vel16 may be NULL; otherwise velocities are zero (equilibrium == 0) or
 * drawn from the Aarseth-Henon-Wielen distribution for gravitational constant G. */
int nbody_gen_plummer(uint64_t seed, uint64_t first, uint64_t count, uint64_t n_total, float world, float a,
                      float rmax, float total_mass, float G, int equilibrium, float* host_masses,
                      float* host_pos16, float* host_vel16);

#ifdef __cplusplus
}
#endif
#endif /* NBODY_HIP_H */
