// Barnes-Hut on gfx950: octree build from Morton keys and per-wavefront traversal.
//
// Replaces OctreeNode::new_tree (src/octree_maxdepth.rs:26-186; single-threaded, on the host,
// every frame) and the traversal of src/nbodybh.wgsl:63-161.
//
// The reference inserts bodies one by one; the tree it ends with is a function of the body SET:
// a cell is a node iff its parent holds >= 2 bodies, a cell with one body is a LEAFBODY, a
// depth-cap cell with >= 2 bodies is a LEAFLIST.  Build here:
//   1. k_morton      key = the octant path of the reference's own comparison descent (bit-exact)
//   2. radix sort    (key, body index), 8-bit digits, stable => equal keys stay in index order
//   3. k_prefix      per sorted body: common-prefix depth with both neighbours => the range of
//                    levels [lo, top] at which it is the first body of an existing cell
//   4. row scans     breadth-first numbering: node id = level_start[d] + rank of the cell in d
//   5. k_emit        writes the nodes; children are consecutive (octant order), so a node only
//                    records where they start
//   6. k_leaflist    depth-cap cells: sequential prefix aggregates in index order (literal
//                    reference arithmetic, src/octree_maxdepth.rs:91-136)
//   7. k_aggregate   bottom-up, one launch per level: mass, centre of mass, bounds, range and
//                    the acceptance thresholds derived from it
// Node numbering differs from the reference's allocation order; the cell set, every body's leaf,
// bounds and range are identical, masses/centres agree to rounding (different summation order).
//
// Traversal: one wavefront walks 64 Morton-adjacent bodies with ONE shared stack of
// (first child, count, lane mask) entries in LDS.  A node is opened if any lane needs it; lanes
// that accepted an ancestor are masked out of its subtree, so each lane interacts with exactly the
// node SET of the reference's private depth-first stack (the order within a body's sum differs:
// the children of a node are handled together).  Node records are 32 bytes and are fetched a
// group at a time with wave-uniform (scalar) loads.
#include "ctx.h"
#include "laws.h"
#include <algorithm>
#include <cstring>
#include <cmath>
#include <cstdlib>

namespace {

constexpr int NCNT = 18;       // level counters 0..16 + leaf-list members
constexpr int LIST_ROW = 17;
constexpr uint32_t NO_NODE = 0xffffffffu;
constexpr int NWALK_COUNTERS = 24;  // 8 totals + 16 bins of wavefront visits by active lanes

// Node record (32 B): written by the build, fetched by the traversal with scalar loads.
// Breadth-first numbering makes the children of a node (octant order) and the members of a leaf
// list consecutive: [first, first + count).
// The acceptance test range / distance < theta (src/nbodybh.wgsl:104-106) is prepared per node
// as two thresholds on the squared distance: r^2 > r2_accept settles "accept", r^2 < r2_open
// settles "open"; in the narrow band between them (relative width 8e-6) the traversal evaluates
// the reference's expression itself, with the node's range from a side array.  A LEAFBODY record
// has both thresholds below zero (QN_ALWAYS).  (Under the PLUMMER law both sides of these
// compares carry the softening eps2, see MacScale.)
struct QNode {
    float cx, cy, cz, mass;
    float r2_open, r2_accept;
    uint32_t first;
    uint32_t count_type;  // count in the low 30 bits, nbody_node_type in the top two
};
__host__ __device__ __forceinline__ uint32_t qn_type(const QNode& nd) { return nd.count_type >> 30; }
__host__ __device__ __forceinline__ uint32_t qn_count(const QNode& nd) { return nd.count_type & 0x3fffffffu; }
__host__ __device__ __forceinline__ uint32_t qn_pack(uint32_t count, uint32_t type) { return (type << 30) | count; }

// Thresholds of a LEAFBODY record: any squared distance exceeds them, so the traversal takes the
// record as a source for every lane without looking at its type.
constexpr float QN_ALWAYS = -1.0f;

// The squared-distance thresholds of a node are range^2 * {open, accept} + {soft_open, soft_accept}:
// open, accept = 1 / (theta^2 (1 +- 4e-6)).  Under the PLUMMER law the traversal compares the
// SOFTENED squared distance r^2 + eps2 (the only form of it the force term needs), so the
// thresholds carry eps2 (1 -+ 1e-6) as well; the margins cover the f32 rounding of both sides
// (1.8e-7 of r^2 + eps2 for the sum, 5e-7 of r^2 for the reference's own sqrt and division).
struct MacScale {
    float open, accept;
    float soft_open, soft_accept;
};
__device__ __forceinline__ void qn_set_range(QNode& nd, float range, MacScale mac) {
    const float q = range * range;
    nd.r2_open = q * mac.open + mac.soft_open;
    nd.r2_accept = q * mac.accept + mac.soft_accept;
}
static_assert(sizeof(QNode) == 32, "node record layout");


}  // namespace

struct Domains;
struct TopPatch;

struct Octree {
    uint64_t n = 0;      // bodies of the present build
    uint64_t n_cap = 0;  // bodies the per-body arrays hold
    uint32_t D = 16;
    // sort buffers
    uint64_t* keys[2] = {nullptr, nullptr};
    uint32_t* idx[2] = {nullptr, nullptr};
    int sorted = 0;  // which buffer holds the sorted result
    uint8_t* lo = nullptr;
    uint8_t* top = nullptr;
    uint8_t* flags = nullptr;
    uint32_t* body_node = nullptr;  // per sorted rank: node of the body's leaf (NO_NODE for list members)
    uint32_t* rs_counts = nullptr;  // [256][rs_nb]
    uint32_t rs_nb = 0;
    uint32_t* row_total = nullptr;  // [256]
    uint32_t* blk = nullptr;        // [NCNT][pf_nb]
    uint32_t pf_nb = 0;
    uint32_t* lvl_total = nullptr;  // [NCNT]
    uint32_t* lvl_start = nullptr;  // [NCNT+1]
    uint32_t* h_lvl_start = nullptr;  // pinned host copy of lvl_start
    hipEvent_t lvl_ready = nullptr;   // the copy into it has completed
    // nodes
    QNode* nodes = nullptr;
    float4* bmin = nullptr;          // bounding box of the bodies below each node
    float4* bmax = nullptr;
    float* range = nullptr;          // longest side of that box (src/octree_maxdepth.rs:101)
    uint32_t* node_head = nullptr;   // sorted rank of the first body of each node's cell
    float* quad = nullptr;           // opt-in: traceless quadrupole about the node's centre of mass,
                                     // 8 floats per node {xx, xy, xz, yy, yz, zz, 0, 0}
    uint64_t cap_nodes = 0;
    uint64_t n_nodes = 0;
    bool have_tree = false;
    bool built = false;  // tree came from nb_tree_build (keys/order valid)
    bool octree_rs = false;  // export in the layout of src/octree.rs (is_leaf, max_depth); no leaf lists
    // export buffer
    nbody_octree_node* export_buf = nullptr;
    uint64_t export_cap = 0;
    bool export_is_upload = false;  // export_buf holds the (renumbered) uploaded tree
    // walk
    uint32_t* shard_blk = nullptr;   // per-workgroup counts of the compaction below, then their scan
    uint32_t* shard_order = nullptr;  // sorted ranks' body indices restricted to the shard
    uint64_t* walk_counters = nullptr;  // [NWALK_COUNTERS]
    uint32_t* err_flag = nullptr;
    // launch order of the traversal's wavefronts: node visits of every wavefront in the last
    // traversal, and the order (longest first within each XCD's share) made from them
    uint32_t* wave_cost = nullptr;
    uint32_t* wave_perm = nullptr;
    uint32_t perm_nblk = 0;            // grid the order was made for (0: none)
    const uint32_t* perm_order = nullptr;  // ... and the target list it belongs to
    uint32_t perm_age = 0;
    // caller-given grouping of the traversal's targets (nbody_bh_set_target_order): body indices,
    // NO_NODE = an empty lane
    uint32_t* user_order = nullptr;
    uint64_t user_m = 0;
    // the library's own grouping for an unsharded traversal: body indices along the Hilbert curve,
    // made from a build's sorted keys and kept for HILBERT_KEEP builds (bodies move little per step,
    // and any permutation of the bodies is a valid grouping)
    uint32_t* hkeys[2] = {nullptr, nullptr};
    uint32_t* hidx[2] = {nullptr, nullptr};
    const uint32_t* horder = nullptr;  // hidx[0] or hidx[1] once made
    uint32_t horder_age = 0;           // builds since it was made
    // domain-decomposed build (nbody_bh_set_domains): local trees of the domains and the merged top
    Domains* dom = nullptr;
};

namespace {

// ---- 1. Morton keys -----------------------------------------------------------------------
// The reference's descent, literally (src/octree_maxdepth.rs:139,142,156,210-230): compare with
// the cell centre (>= goes high; x->4, y->2, z->1), halve the extent, move the centre.
__global__ void k_morton(const float4* __restrict__ posm, uint32_t n, float world, int D, uint64_t* __restrict__ keys,
                         uint32_t* __restrict__ idx) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 p = posm[i];
    float ext = world / 2.0f;
    float cx = ext, cy = ext, cz = ext;
    uint64_t key = 0;
    for (int l = 0; l < D; ++l) {
        ext = ext / 2.0f;
        uint32_t ci = 0;
        float wx = -1.0f, wy = -1.0f, wz = -1.0f;
        if (p.x >= cx) { ci |= 4u; wx = 1.0f; }
        if (p.y >= cy) { ci |= 2u; wy = 1.0f; }
        if (p.z >= cz) { ci |= 1u; wz = 1.0f; }
        key = (key << 3) | ci;
        cx = cx + (ext * wx);
        cy = cy + (ext * wy);
        cz = cz + (ext * wz);
    }
    keys[i] = key;
    idx[i] = i;
}

// ---- 1b. Hilbert keys (grouping of the traversal's targets) ----------------------------------
// The position of a body's depth-L cell on the Hilbert curve through the root cube (Skilling's
// transform of the cell coordinates, which the Morton key holds bit-interleaved).  Like the Morton
// order it keeps every octree cell contiguous; unlike it, consecutive cells are always neighbours,
// so 64 consecutive bodies make a more compact group.  No counterpart in the reference, whose
// traversal is one thread per body.
__device__ __forceinline__ uint32_t hilbert_from_morton(uint64_t key, int D, int L) {
    uint32_t X[3] = {0u, 0u, 0u};
    for (int l = 0; l < D; ++l) {
        const uint32_t d = (uint32_t)(key >> (3 * (D - 1 - l))) & 7u;
        X[0] = (X[0] << 1) | ((d >> 2) & 1u);
        X[1] = (X[1] << 1) | ((d >> 1) & 1u);
        X[2] = (X[2] << 1) | (d & 1u);
    }
    for (uint32_t Q = 1u << (D - 1); Q > 1u; Q >>= 1) {
        const uint32_t P = Q - 1u;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (X[i] & Q) {
                X[0] ^= P;
            } else {
                const uint32_t t = (X[0] ^ X[i]) & P;
                X[0] ^= t;
                X[i] ^= t;
            }
        }
    }
    X[1] ^= X[0];
    X[2] ^= X[1];
    uint32_t t = 0u;
    for (uint32_t Q = 1u << (D - 1); Q > 1u; Q >>= 1)
        if (X[2] & Q) t ^= Q - 1u;
    X[0] ^= t;
    X[1] ^= t;
    X[2] ^= t;
    uint32_t h = 0u;  // the top L levels: 3 L <= 30 bits
    for (int b = D - 1; b >= D - L; --b) {
        h = (h << 1) | ((X[0] >> b) & 1u);
        h = (h << 1) | ((X[1] >> b) & 1u);
        h = (h << 1) | ((X[2] >> b) & 1u);
    }
    return h;
}

// Per sorted rank: the Hilbert key of the body's cell and the body's index (the sort's payload).
__global__ void k_hilbert(const uint64_t* __restrict__ keys_sorted, const uint32_t* __restrict__ idx_sorted, uint32_t n, int D,
                          int L, uint32_t* __restrict__ hkeys, uint32_t* __restrict__ hidx) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    hkeys[r] = hilbert_from_morton(keys_sorted[r], D, L);
    hidx[r] = idx_sorted[r];
}

// ---- 2. LSD radix sort, 8-bit digits ------------------------------------------------------
constexpr int RS_THREADS = 256;
constexpr int RS_WAVES = RS_THREADS / 64;
constexpr int RS_ITEMS = 16;
constexpr int RS_TILE = RS_THREADS * RS_ITEMS;  // items of one workgroup: wavefront w owns [w, w+1) * RS_TILE / RS_WAVES
constexpr int RS_WTILE = RS_TILE / RS_WAVES;

template <class K>
__global__ __launch_bounds__(RS_THREADS) void k_rs_hist(const K* __restrict__ keys, uint32_t n, int shift,
                                                        uint32_t* __restrict__ counts, uint32_t nb) {
    __shared__ uint32_t h[256];
    const int tid = threadIdx.x;
    h[tid] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * RS_TILE;
#pragma unroll
    for (int it = 0; it < RS_ITEMS; ++it) {
        uint32_t i = base + it * RS_THREADS + tid;
        if (i < n) atomicAdd(&h[(uint32_t)(keys[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    counts[(size_t)tid * nb + blockIdx.x] = h[tid];
}

// Exclusive scan of each row of a [rows][nb] matrix in place (one workgroup per row), row sums
// to row_total.
__global__ __launch_bounds__(256) void k_rowscan(uint32_t* __restrict__ m, uint32_t nb, uint32_t* __restrict__ row_total) {
    __shared__ uint32_t wsum[4];
    __shared__ uint32_t carry_s;
    uint32_t* p = m + (size_t)blockIdx.x * nb;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (uint32_t base = 0; base < nb; base += 1024) {
        uint32_t i0 = base + tid * 4;
        uint32_t v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = (i0 + k < nb) ? p[i0 + k] : 0u;
        uint32_t s = v[0] + v[1] + v[2] + v[3];
        uint32_t x = s;  // inclusive wave scan
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            uint32_t y = __shfl_up(x, o, 64);
            if (lane >= o) x += y;
        }
        if (lane == 63) wsum[wave] = x;
        __syncthreads();
        uint32_t woff = 0;
        for (int w = 0; w < wave; ++w) woff += wsum[w];
        uint32_t carry = carry_s;
        uint32_t excl = carry + woff + x - s;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (i0 + k < nb) p[i0 + k] = excl;
            excl += v[k];
        }
        __syncthreads();
        if (tid == 255) carry_s = carry + woff + x;
        __syncthreads();
    }
    if (tid == 0) row_total[blockIdx.x] = carry_s;
}

// Exclusive scan of rows (<= 256) totals; out[rows] = grand total.
__global__ __launch_bounds__(256) void k_scan_small(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, int rows) {
    __shared__ uint32_t s[257];
    const int tid = threadIdx.x;
    s[tid] = tid < rows ? in[tid] : 0u;
    __syncthreads();
    if (tid == 0) {
        uint32_t acc = 0;
        for (int r = 0; r < rows; ++r) {
            uint32_t v = s[r];
            s[r] = acc;
            acc += v;
        }
        s[rows] = acc;
    }
    __syncthreads();
    if (tid <= rows) out[tid] = s[tid];
    if (tid == 0 && rows == 256) out[256] = s[256];
}

// One tile: rank every item among the tile's items of the same digit (stable: wavefront-major,
// then load order, then lane), stage the tile in LDS in digit order, then write each digit's run
// to its place in the output with consecutive lanes on consecutive addresses.
template <class K>
__global__ __launch_bounds__(RS_THREADS) void k_rs_scatter(const K* __restrict__ keys_in,
                                                           const uint32_t* __restrict__ idx_in,
                                                           K* __restrict__ keys_out, uint32_t* __restrict__ idx_out,
                                                           uint32_t n, int shift, const uint32_t* __restrict__ counts,
                                                           const uint32_t* __restrict__ row_total, uint32_t nb) {
    __shared__ uint32_t wcnt[RS_WAVES][256];  // per-wavefront digit counts, then tile-local offsets
    __shared__ uint32_t gbase[256];           // output position of staged item j of digit d = gbase[d] + j
    __shared__ uint32_t wsum[RS_WAVES], gsum[RS_WAVES];
    __shared__ K skey[RS_TILE];
    __shared__ uint32_t sval[RS_TILE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int w = 0; w < RS_WAVES; ++w) wcnt[w][tid] = 0;
    __syncthreads();
    const uint32_t tile0 = blockIdx.x * RS_TILE;
    const uint32_t wbase = tile0 + wave * RS_WTILE;
    const unsigned long long lt = (1ull << lane) - 1ull;
    K key[RS_ITEMS];
    uint32_t val[RS_ITEMS], rnk[RS_ITEMS];
#pragma unroll
    for (int it = 0; it < RS_ITEMS; ++it) {
        const uint32_t i = wbase + it * 64 + lane;
        key[it] = 0;
        val[it] = 0;
        if (i < n) {
            key[it] = keys_in[i];
            val[it] = idx_in[i];
        }
    }
    volatile uint32_t* wc = wcnt[wave];  // this wavefront's counters: its LDS accesses stay in program order
#pragma unroll
    for (int it = 0; it < RS_ITEMS; ++it) {
        const bool valid = wbase + it * 64 + lane < n;
        const uint32_t d = (uint32_t)(key[it] >> shift) & 255u;
        // lanes of this wavefront holding the same digit
        unsigned long long peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const bool bit = (d >> b) & 1u;
            const unsigned long long m = __ballot(bit);
            peers &= bit ? m : ~m;
        }
        const uint32_t r = __popcll(peers & lt);
        const uint32_t prior = wc[d];
        rnk[it] = prior + r;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (valid && r == 0) wc[d] = prior + (uint32_t)__popcll(peers);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    __syncthreads();
    {
        // thread d: the tile's count of digit d, its exclusive scan over the digits, and from
        // that each wavefront's first staged slot for the digit
        uint32_t c[RS_WAVES], tot = 0;
#pragma unroll
        for (int w = 0; w < RS_WAVES; ++w) {
            c[w] = wcnt[w][tid];
            tot += c[w];
        }
        uint32_t x = tot;  // inclusive scan over the wavefront
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t y = __shfl_up(x, o, 64);
            if (lane >= o) x += y;
        }
        // ... and the same scan over the digits' totals in the whole input: where digit d begins
        const uint32_t gt = row_total[tid];
        uint32_t gx = gt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t y = __shfl_up(gx, o, 64);
            if (lane >= o) gx += y;
        }
        if (lane == 63) {
            wsum[wave] = x;
            gsum[wave] = gx;
        }
        __syncthreads();
        uint32_t start = x - tot, gstart = gx - gt;
        for (int w = 0; w < wave; ++w) {
            start += wsum[w];
            gstart += gsum[w];
        }
        gbase[tid] = gstart + counts[(size_t)tid * nb + blockIdx.x] - start;
        uint32_t o = start;
#pragma unroll
        for (int w = 0; w < RS_WAVES; ++w) {
            wcnt[w][tid] = o;
            o += c[w];
        }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < RS_ITEMS; ++it) {
        if (wbase + it * 64 + lane < n) {
            const uint32_t d = (uint32_t)(key[it] >> shift) & 255u;
            const uint32_t slot = wcnt[wave][d] + rnk[it];
            skey[slot] = key[it];
            sval[slot] = val[it];
        }
    }
    __syncthreads();
    const uint32_t cnt = min((uint32_t)RS_TILE, n - tile0);
    for (uint32_t j = tid; j < cnt; j += RS_THREADS) {
        const K k = skey[j];
        const uint32_t pos = gbase[(uint32_t)(k >> shift) & 255u] + j;
        keys_out[pos] = k;
        idx_out[pos] = sval[j];
    }
}

// ---- 3. neighbour prefixes ----------------------------------------------------------------
// Levels (0..D) two keys share: D if equal.
__device__ __forceinline__ int common_levels(uint64_t a, uint64_t b, int D) {
    uint64_t x = a ^ b;
    if (x == 0) return D;
    int hb = 63 - __clzll((long long)x);
    return D - 1 - hb / 3;
}

// flags: bit0 = contributes a leaf-list member node, bit1 = first body of a depth-cap cell
// holding >= 2 bodies, bit2 = member of such a cell.
// multi (optional, per body index): the entry stands for a whole cell of >= 2 bodies at depth D, which
// exists down to that depth whatever its neighbours (the top tree over the domains' cells).
__global__ __launch_bounds__(256) void k_prefix(const uint64_t* __restrict__ keys, uint32_t n, int D, int list_mode,
                                                uint8_t* __restrict__ lo, uint8_t* __restrict__ top,
                                                uint8_t* __restrict__ flags, uint32_t* __restrict__ blk, uint32_t nb,
                                                const uint32_t* __restrict__ idx, const uint8_t* __restrict__ multi) {
    __shared__ uint32_t sh[NCNT];
    const int tid = threadIdx.x, lane = tid & 63;
    if (tid < NCNT) sh[tid] = 0;
    __syncthreads();
    const uint32_t i = blockIdx.x * 256 + tid;
    const bool valid = i < n;
    int l = 1, t = 0;
    uint32_t f = 0;
    if (valid) {
        const uint64_t k = keys[i];
        const int Lp = (i == 0) ? -1 : common_levels(keys[i - 1], k, D);
        const int Ln = (i + 1 == n) ? -1 : common_levels(k, keys[i + 1], D);
        l = Lp + 1;
        t = min(D, max(Lp, Ln) + 1);
        if (multi && multi[idx[i]]) t = D;
        const bool inlist = (Lp == D) || (Ln == D);
        const bool member = list_mode ? inlist : (inlist && Ln == D);
        f = (member ? 1u : 0u) | ((Lp < D && Ln == D) ? 2u : 0u) | (inlist ? 4u : 0u);
        lo[i] = (uint8_t)l;
        top[i] = (uint8_t)t;
        flags[i] = (uint8_t)f;
    }
#pragma unroll
    for (int d = 0; d < NCNT; ++d) {
        const bool has = valid && ((d < LIST_ROW) ? (l <= d && d <= t) : (f & 1u));
        const unsigned long long b = __ballot(has);
        if (lane == 0 && b) atomicAdd(&sh[d], (uint32_t)__popcll(b));
    }
    __syncthreads();
    if (tid < NCNT) blk[(size_t)tid * nb + blockIdx.x] = sh[tid];
}

// ---- 5. emit nodes ------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_emit(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ idx,
                                              const float4* __restrict__ posm, uint32_t n, int D,
                                              const uint8_t* __restrict__ lo, const uint8_t* __restrict__ top,
                                              const uint8_t* __restrict__ flags, const uint32_t* __restrict__ blk,
                                              uint32_t nb, const uint32_t* __restrict__ lvl_start, QNode* __restrict__ nodes,
                                              float4* __restrict__ bmin, float4* __restrict__ bmax, float* __restrict__ range,
                                              uint32_t* __restrict__ node_head, uint32_t* __restrict__ body_node,
                                              uint32_t cap_nodes, uint32_t* __restrict__ err_flag) {
    __shared__ uint32_t wtot[NCNT][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t i = blockIdx.x * 256 + tid;
    const bool valid = i < n;
    int l = 1, t = 0;
    uint32_t f = 0;
    if (valid) {
        l = lo[i];
        t = top[i];
        f = flags[i];
    }
    const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
    for (int d = 0; d < NCNT; ++d) {
        const bool has = valid && ((d < LIST_ROW) ? (l <= d && d <= t) : (f & 1u));
        const unsigned long long b = __ballot(has);
        if (lane == 0) wtot[d][wave] = (uint32_t)__popcll(b);
    }
    __syncthreads();
    // rank of this body among the leaf-list members
    uint32_t list_first = 0;
    {
        const unsigned long long b = __ballot(valid && (f & 1u));
        uint32_t base = lvl_start[LIST_ROW] + blk[(size_t)LIST_ROW * nb + blockIdx.x];
        for (int w = 0; w < wave; ++w) base += wtot[LIST_ROW][w];
        list_first = base + (uint32_t)__popcll(b & lt);
    }
    float4 p = make_float4(0, 0, 0, 0);
    if (valid && l <= t) p = posm[idx[i]];
    if (valid && l > t) body_node[i] = NO_NODE;
    // Node id at level d = level_start[d] + rank of the body among the level-d heads.  That same
    // rank, taken by a body that heads a cell at level d-1, is where that cell's children start
    // (and for a childless cell it makes the next cell's start minus its own come out as 0).
    uint32_t pend = NO_NODE;
    for (int d = 0; d <= D; ++d) {
        const bool has = valid && l <= d && d <= t;
        const unsigned long long b = __ballot(has);
        uint32_t base = lvl_start[d] + blk[(size_t)d * nb + blockIdx.x];
        for (int w = 0; w < wave; ++w) base += wtot[d][w];
        const uint32_t id = base + (uint32_t)__popcll(b & lt);
        if (pend != NO_NODE) nodes[pend].first = id;
        pend = NO_NODE;
        if (has) {
            if (id >= cap_nodes) {
                *err_flag = 1;
            } else {
                uint32_t type;
                if (d < t) type = NBODY_NODE_INTERIOR;
                else if (f & 2u) type = NBODY_NODE_LEAFLIST;  // d == t == D, first of >= 2 bodies
                else type = NBODY_NODE_LEAFBODY;
                QNode nd;
                nd.cx = p.x; nd.cy = p.y; nd.cz = p.z; nd.mass = p.w;  // src/octree_maxdepth.rs:75-86 for a LEAFBODY;
                nd.r2_open = nd.r2_accept = QN_ALWAYS;                 // aggregates are filled in later otherwise
                nd.first = 0;
                nd.count_type = qn_pack(0, type);
                nodes[id] = nd;
                // (a LEAFBODY's box is its position and its range 0: bmin/bmax/range hold entries
                // for the other node types only)
                node_head[id] = i;
                if (d == t) body_node[i] = id;
                pend = id;
            }
        }
    }
    if (pend != NO_NODE) nodes[pend].first = list_first;  // depth-cap cell: its list members (if any)
}

// ---- 6. depth-cap cells -------------------------------------------------------------------
// The reference's aggregate recurrence over a cell's bodies in index order, and the list members it
// creates (src/octree_maxdepth.rs:91-136).  The recurrence is a chain of f32 roundings (each new
// centre is ((c m) + (p_b m_b)) / (m + m_b) of the previous one), so it stays sequential; what a
// wavefront does together is everything around it: a cell's bodies are loaded 64 at a time, every
// lane follows the chain (so each holds the aggregate as it stood before its own body, which is the
// literal list member), the box is a reduction, and the member records go out side by side.  A
// wavefront takes the cells headed by its 64 sorted bodies one after the other.  (One thread per
// cell, each load waiting for the one before, took 50 ms for 100,000 bodies in one cell.)
__global__ __launch_bounds__(256) void k_leaflist(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ idx,
                                                  const float4* __restrict__ posm, uint32_t n, int list_mode,
                                                  const uint8_t* __restrict__ flags, const uint32_t* __restrict__ body_node,
                                                  QNode* __restrict__ nodes, float4* __restrict__ bmin, float4* __restrict__ bmax,
                                                  float* __restrict__ range, uint32_t* __restrict__ node_head, uint32_t cap_nodes,
                                                  MacScale mac) {
    const int lane = threadIdx.x & 63;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long heads = __ballot(i < n && (flags[min(i, n - 1)] & 2u));
    const uint32_t wave0 = i - (uint32_t)lane;
    while (heads) {
        const int h = (int)__builtin_ctzll(heads);
        heads &= heads - 1;
        const uint32_t ih = wave0 + (uint32_t)h;  // the cell's first body (wave-uniform)
        const uint32_t id = body_node[ih];
        if (id >= cap_nodes) continue;
        QNode nd = nodes[id];
        const uint64_t key = keys[ih];
        uint32_t member = nd.first, count = 0;
        float4 a = posm[idx[ih]];  // aggregate: xyz = centre of mass, w = mass
        float mnx = a.x, mny = a.y, mnz = a.z, mxx = a.x, mxy = a.y, mxz = a.z;
        auto put = [&](uint32_t slot, float4 q) {
            if (slot < cap_nodes) {
                QNode c;
                c.cx = q.x; c.cy = q.y; c.cz = q.z; c.mass = q.w;
                c.r2_open = c.r2_accept = QN_ALWAYS;
                c.first = 0;
                c.count_type = qn_pack(0, NBODY_NODE_LEAFBODY);
                nodes[slot] = c;
                node_head[slot] = ih;
            }
        };
        if (list_mode) {  // every body of the cell is a member, the first one included
            if (lane == 0) put(member, a);
            ++member;
            ++count;
        }
        for (uint32_t base = ih + 1; base < n; base += 64) {
            const uint32_t j = base + (uint32_t)lane;
            const bool ok = j < n && keys[min(j, n - 1)] == key;  // equal keys are consecutive: the lanes that hold are a prefix
            const int cnt = __popcll(__ballot(ok));
            if (cnt == 0) break;
            float4 b = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (ok) b = posm[idx[j]];
            const float bwx = b.x * b.w, bwy = b.y * b.w, bwz = b.z * b.w;
            float4 before = a;  // the aggregate as it stood before this lane's body
            for (int k = 0; k < cnt; ++k) {
                const float m_k = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, b.w), k));
                const float x_k = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, bwx), k));
                const float y_k = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, bwy), k));
                const float z_k = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, bwz), k));
                if (lane == k) before = a;
                const float tm = a.w + m_k;
                a.x = ((a.x * a.w) + x_k) / tm;
                a.y = ((a.y * a.w) + y_k) / tm;
                a.z = ((a.z * a.w) + z_k) / tm;
                a.w = tm;
            }
            // the box of the chunk's bodies (min and max are exact in any order)
            float lx = ok ? b.x : INFINITY, ly = ok ? b.y : INFINITY, lz = ok ? b.z : INFINITY;
            float hx = ok ? b.x : -INFINITY, hy = ok ? b.y : -INFINITY, hz = ok ? b.z : -INFINITY;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                lx = fminf(lx, __shfl_xor(lx, o, 64)); ly = fminf(ly, __shfl_xor(ly, o, 64)); lz = fminf(lz, __shfl_xor(lz, o, 64));
                hx = fmaxf(hx, __shfl_xor(hx, o, 64)); hy = fmaxf(hy, __shfl_xor(hy, o, 64)); hz = fmaxf(hz, __shfl_xor(hz, o, 64));
            }
            mnx = fminf(mnx, lx); mny = fminf(mny, ly); mnz = fminf(mnz, lz);
            mxx = fmaxf(mxx, hx); mxy = fmaxf(mxy, hy); mxz = fmaxf(mxz, hz);
            // literal: the member is the PREVIOUS aggregate (node_a_*); else the body itself
            if (ok) put(member + (uint32_t)lane, list_mode ? b : before);
            member += (uint32_t)cnt;
            count += (uint32_t)cnt;
            if (cnt < 64) break;
        }
        if (lane == 0) {
            nd.cx = a.x; nd.cy = a.y; nd.cz = a.z; nd.mass = a.w;
            const float rg = fmaxf(mxx - mnx, fmaxf(mxy - mny, mxz - mnz));
            qn_set_range(nd, rg, mac);
            nd.count_type = qn_pack(count, NBODY_NODE_LEAFLIST);
            nodes[id] = nd;
            range[id] = rg;
            bmin[id] = make_float4(mnx, mny, mnz, 0.0f);
            bmax[id] = make_float4(mxx, mxy, mxz, 0.0f);
        }
    }
}

// ---- 7. aggregates ------------------------------------------------------------------------
// One level: nodes [first_id, first_id + count).  The children of a node end where the next
// node's begin (`child_end` for the last node of the level).
__global__ void k_aggregate(QNode* __restrict__ nodes, float4* __restrict__ bmin, float4* __restrict__ bmax,
                            float* __restrict__ range, uint32_t first_id, uint32_t count, uint32_t child_end, MacScale mac) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= count) return;
    const uint32_t id = first_id + r;
    QNode nd = nodes[id];
    if (qn_type(nd) != NBODY_NODE_INTERIOR) return;
    const uint32_t c0 = nd.first, c1 = (r + 1 < count) ? nodes[id + 1].first : child_end;
    // An interior node has at most eight children: all their records (and the boxes of those that
    // are cells themselves) are requested before the first is used; the sums then run in child
    // order, as a loop over the children would.
    QNode ch[8];
    float4 lo[8], hi[8];
#pragma unroll
    for (int k = 0; k < 8; ++k)
        if (c0 + k < c1) ch[k] = nodes[c0 + k];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if (c0 + k < c1) {
            lo[k] = hi[k] = make_float4(ch[k].cx, ch[k].cy, ch[k].cz, 0.0f);
            if (qn_type(ch[k]) != NBODY_NODE_LEAFBODY) {
                lo[k] = bmin[c0 + k];
                hi[k] = bmax[c0 + k];
            }
        }
    }
    float M = 0.0f, wx = 0.0f, wy = 0.0f, wz = 0.0f;
    float mnx = INFINITY, mny = INFINITY, mnz = INFINITY, mxx = -INFINITY, mxy = -INFINITY, mxz = -INFINITY;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if (c0 + k < c1) {
            M += ch[k].mass;
            wx += ch[k].cx * ch[k].mass;
            wy += ch[k].cy * ch[k].mass;
            wz += ch[k].cz * ch[k].mass;
            mnx = fminf(mnx, lo[k].x); mny = fminf(mny, lo[k].y); mnz = fminf(mnz, lo[k].z);
            mxx = fmaxf(mxx, hi[k].x); mxy = fmaxf(mxy, hi[k].y); mxz = fmaxf(mxz, hi[k].z);
        }
    }
    nd.mass = M;
    nd.cx = wx / M;
    nd.cy = wy / M;
    nd.cz = wz / M;
    const float rg = fmaxf(mxx - mnx, fmaxf(mxy - mny, mxz - mnz));  // src/octree_maxdepth.rs:101
    qn_set_range(nd, rg, mac);
    nd.count_type = qn_pack(c1 - c0, NBODY_NODE_INTERIOR);
    nodes[id] = nd;
    range[id] = rg;
    bmin[id] = make_float4(mnx, mny, mnz, 0.0f);
    bmax[id] = make_float4(mxx, mxy, mxz, 0.0f);
}

// ---- 7b. quadrupole moments (opt-in, no counterpart in the reference) -----------------------
// One level, after its centres of mass are known: Q = sum over the children of
// Q_child + m (3 d d^T - |d|^2 I), d = child centre - own centre (parallel-axis shift; a body has
// no moment of its own).  Depth-cap cells are summed over their members when those are the bodies
// themselves (leaflist_mode 1); the literal member lists are not bodies and get Q = 0.
__global__ void k_quadrupole(const QNode* __restrict__ nodes, float* __restrict__ quad, uint32_t first_id, uint32_t count,
                             int list_mode) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= count) return;
    const uint32_t id = first_id + r;
    const QNode nd = nodes[id];
    const uint32_t type = qn_type(nd);
    if (type == NBODY_NODE_LEAFBODY) return;
    float q[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    if (type == NBODY_NODE_INTERIOR || (type == NBODY_NODE_LEAFLIST && list_mode)) {
        const uint32_t c0 = nd.first, c1 = nd.first + qn_count(nd);
        for (uint32_t c = c0; c < c1; ++c) {
            const QNode ch = nodes[c];
            const float dx = ch.cx - nd.cx, dy = ch.cy - nd.cy, dz = ch.cz - nd.cz;
            const float d2 = fmaf(dx, dx, fmaf(dy, dy, dz * dz)), m = ch.mass;
            q[0] += m * (3.0f * dx * dx - d2);
            q[1] += m * (3.0f * dx * dy);
            q[2] += m * (3.0f * dx * dz);
            q[3] += m * (3.0f * dy * dy - d2);
            q[4] += m * (3.0f * dy * dz);
            q[5] += m * (3.0f * dz * dz - d2);
            if (qn_type(ch) != NBODY_NODE_LEAFBODY) {
                const float* qc = quad + (size_t)c * 8;
#pragma unroll
                for (int k = 0; k < 6; ++k) q[k] += qc[k];
            }
        }
    }
    float4* dst = reinterpret_cast<float4*>(quad + (size_t)id * 8);
    dst[0] = make_float4(q[0], q[1], q[2], q[3]);
    dst[1] = make_float4(q[4], q[5], 0.0f, 0.0f);
}

// ---- export / import (96-byte reference layout) ---------------------------------------------
__global__ void k_export(const QNode* __restrict__ nodes, const float4* __restrict__ bmin, const float4* __restrict__ bmax,
                         const float* __restrict__ range, const uint32_t* __restrict__ node_head, const uint64_t* __restrict__ keys,
                         const uint32_t* __restrict__ lvl_start, int D, uint32_t count, int octree_rs,
                         nbody_octree_node* __restrict__ out) {
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= count) return;
    const QNode nd = nodes[id];
    const bool leaf = qn_type(nd) == NBODY_NODE_LEAFBODY;  // its box is its position, its range 0
    const float4 lo = leaf ? make_float4(nd.cx, nd.cy, nd.cz, 0.0f) : bmin[id];
    const float4 hi = leaf ? make_float4(nd.cx, nd.cy, nd.cz, 0.0f) : bmax[id];
    int depth = 0;
    while (depth < D && id >= lvl_start[depth + 1]) ++depth;  // list members land on D, the depth of their cell
    nbody_octree_node o;
    o.center_of_mass[0] = nd.cx; o.center_of_mass[1] = nd.cy; o.center_of_mass[2] = nd.cz; o._pad0 = 0.0f;
    o.pos_min[0] = lo.x; o.pos_min[1] = lo.y; o.pos_min[2] = lo.z; o._pad1 = 0.0f;
    o.pos_max[0] = hi.x; o.pos_max[1] = hi.y; o.pos_max[2] = hi.z;
    o.range = leaf ? 0.0f : range[id];
    o.total_mass = nd.mass;
    for (int k = 0; k < 8; ++k) o.child_indices[k] = 0;
    const uint32_t type = qn_type(nd), cnt = qn_count(nd);
    if (type == NBODY_NODE_INTERIOR) {
        // the octant of a child is the digit of its bodies' keys at the child's level
        for (uint32_t c = nd.first; c < nd.first + cnt; ++c)
            o.child_indices[(uint32_t)(keys[node_head[c]] >> (3 * (D - depth - 1))) & 7u] = c;
    } else if (type == NBODY_NODE_LEAFLIST) {
        o.child_indices[0] = nd.first;
        o.child_indices[1] = cnt;
    }
    o.node_type = type;
    o.depth = (uint32_t)depth;
    if (octree_rs) {
        // src/octree.rs:5-15: is_leaf where the capped builder has node_type, and max_depth, which
        // only the root carries (:36): the value returned for the deepest leaf (:58) = its depth + 1
        o.node_type = type == NBODY_NODE_LEAFBODY ? 1u : 0u;
        o.depth = 0;
        if (id == 0) {
            int deepest = 0;
            for (int d = 0; d <= D; ++d)
                if (lvl_start[d + 1] > lvl_start[d]) deepest = d;
            o.depth = (uint32_t)deepest + 1u;
        }
    }
    o._pad2 = 0;
    out[id] = o;
}

// Uploaded trees arrive renumbered breadth-first by the host: children consecutive in octant order.
__global__ void k_import(const nbody_octree_node* __restrict__ in, uint32_t count, QNode* __restrict__ nodes,
                         float4* __restrict__ bmin, float4* __restrict__ bmax, float* __restrict__ range, MacScale mac,
                         uint32_t* __restrict__ err_flag) {
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= count) return;
    const nbody_octree_node o = in[id];
    QNode nd;
    nd.cx = o.center_of_mass[0]; nd.cy = o.center_of_mass[1]; nd.cz = o.center_of_mass[2];
    nd.mass = o.total_mass;
    qn_set_range(nd, o.range, mac);
    if (o.node_type == NBODY_NODE_LEAFBODY) nd.r2_open = nd.r2_accept = QN_ALWAYS;
    nd.first = 0;
    uint32_t cnt = 0;
    if (o.node_type == NBODY_NODE_INTERIOR) {
        uint32_t last = 0;
        for (int k = 0; k < 8; ++k) {
            const uint32_t c = o.child_indices[k];
            if (c == 0) continue;
            if (cnt == 0) nd.first = c;
            else if (c != last + 1) *err_flag = 3;
            last = c;
            ++cnt;
        }
    } else if (o.node_type == NBODY_NODE_LEAFLIST) {
        nd.first = o.child_indices[0];
        cnt = max(o.child_indices[1], 1u);  // src/nbodybh.wgsl:139-146 is a do-while: at least one member
    }
    nd.count_type = qn_pack(cnt, o.node_type);
    nodes[id] = nd;
    range[id] = o.range;
    bmin[id] = make_float4(o.pos_min[0], o.pos_min[1], o.pos_min[2], 0.0f);
    bmax[id] = make_float4(o.pos_max[0], o.pos_max[1], o.pos_max[2], 0.0f);
}

// Body indices of the sorted order restricted to [i_begin, i_end): stable compaction, one
// workgroup (the shard lists are rebuilt once per step; n / 1024 trips).
__global__ void k_unrank(const uint32_t* __restrict__ order, const float4* __restrict__ by_rank, uint32_t n,
                         float4* __restrict__ by_body) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n) by_body[order[r]] = by_rank[r];
}

constexpr int CO_ITEMS = 16, CO_TILE = 256 * CO_ITEMS;

__global__ __launch_bounds__(256) void k_compact_count(const uint32_t* __restrict__ idx, uint32_t n, uint32_t i_begin,
                                                       uint32_t i_end, uint32_t* __restrict__ blk) {
    __shared__ uint32_t wsum[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t c = 0;
    for (int it = 0; it < CO_ITEMS; ++it) {
        const uint32_t r = blockIdx.x * CO_TILE + it * 256 + tid;
        if (r < n) {
            const uint32_t b = idx[r];
            c += (b >= i_begin && b < i_end) ? 1u : 0u;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
    if (lane == 0) wsum[wave] = c;
    __syncthreads();
    if (tid == 0) blk[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// blk holds the exclusive scan of the per-workgroup counts; the sorted order is kept.
__global__ __launch_bounds__(256) void k_compact_scatter(const uint32_t* __restrict__ idx, uint32_t n, uint32_t i_begin,
                                                         uint32_t i_end, const uint32_t* __restrict__ blk,
                                                         uint32_t* __restrict__ out, uint32_t cap) {
    __shared__ uint32_t wsum[2][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t run = blk[blockIdx.x];
    for (int it = 0; it < CO_ITEMS; ++it) {
        const uint32_t r = blockIdx.x * CO_TILE + it * 256 + tid;
        uint32_t b = 0;
        bool keep = false;
        if (r < n) {
            b = idx[r];
            keep = b >= i_begin && b < i_end;
        }
        const unsigned long long m = __ballot(keep);
        if (lane == 0) wsum[it & 1][wave] = (uint32_t)__popcll(m);
        __syncthreads();  // the other half of wsum is what the previous round may still be reading
        uint32_t off = run;
        for (int w = 0; w < wave; ++w) off += wsum[it & 1][w];
        off += (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        if (keep && off < cap) out[off] = b;
        run += wsum[it & 1][0] + wsum[it & 1][1] + wsum[it & 1][2] + wsum[it & 1][3];
    }
}

// ---- traversal ----------------------------------------------------------------------------
// The children of a node are CONTIGUOUS (breadth-first numbering; uploaded trees are renumbered on
// the host), so a node that must be opened is pushed as (first child, count, lane mask) and its
// children are fetched together when it is popped.
constexpr bool WALK_PRECISE = false;  // one Newton step on v_rsq_f32 in the traversal's interactions
constexpr int WALK_STACK = 160;  // >= 8 pushes per level x 17 levels + root
constexpr int WALK_THREADS = 64;  // one wavefront per workgroup: its slot is refilled as soon as it ends (3 % over 256)
constexpr int WALK_BATCH = 8;    // node records fetched together; the node array is padded by as many
constexpr int HILBERT_LEVELS = 10;      // levels of the octree the Hilbert keys resolve (30 bits, four sort passes)
constexpr uint32_t HILBERT_KEEP = 16;   // builds after which the order is made afresh
constexpr uint64_t HILBERT_MIN_N = 32768;  // below this the sort costs more than the grouping saves

// The lanes for which p holds, as a wave-uniform value the compiler keeps in scalar registers
// (without the empty asm it folds "ballot != 0" back into a per-lane test).
__device__ __forceinline__ unsigned long long lanes(bool p) {
    unsigned long long m = __ballot(p);
    asm volatile("" : "+s"(m));
    return m;
}

// Holds the traversal until a fetched record has arrived, so that the fetches of a batch are
// all issued before the first record is looked at.
__device__ __forceinline__ void pin_record(const QNode& nd) {
    asm volatile("" ::"s"(nd.cx), "s"(nd.cy), "s"(nd.cz), "s"(nd.mass), "s"(nd.r2_open), "s"(nd.r2_accept), "s"(nd.first),
                 "s"(nd.count_type));
}

// A node's traceless quadrupole moment about its centre of mass (opt-in), 32 B like the record.
struct QMoment {
    float xx, xy, xz, yy, yz, zz, pad0, pad1;
};
__device__ __forceinline__ void pin_moment(const QMoment& m) {
    asm volatile("" ::"s"(m.xx), "s"(m.xy), "s"(m.xz), "s"(m.yy), "s"(m.yz), "s"(m.zz));
}

// A popped group's lane set is held in EXEC while its records are visited: inside
// `if (lane in mask)` the compare of a record is made by the group's lanes only, so "some lane does
// not clearly accept" is the compare's own result (no mask arithmetic), and where every lane
// accepts, the term is accumulated without a select.  Groups longer than a batch leave their
// remainder on the stack.
// Seven wavefronts per SIMD (at most 96 scalar registers): measured 4 % faster at N = 16M than the
// six the unconstrained allocation is admitted with; eight (80) spills and is twice as slow.
template <int LAW, bool COUNT, bool QUAD>
__global__ __launch_bounds__(WALK_THREADS) __attribute__((amdgpu_waves_per_eu(7, 7))) void k_walk(const QNode* __restrict__ q, const uint32_t* __restrict__ order, uint32_t m,
                                              uint32_t i_begin, const float4* __restrict__ posm,
                                              const float4* __restrict__ vel, float4* __restrict__ acc_out, float G,
                                              float eps2, float theta, const float* __restrict__ range,
                                              unsigned long long* __restrict__ counters,
                                              uint32_t* __restrict__ err_flag, int out_by_rank,
                                              const float* __restrict__ quad, const uint32_t* __restrict__ perm,
                                              uint32_t* __restrict__ wave_cost, uint32_t* __restrict__ mark,
                                              const uint32_t* __restrict__ dom_off, uint32_t n_dom) {
    __shared__ uint4 stack[WALK_STACK];  // {first, count, mask.lo, mask.hi}
    const int lane = threadIdx.x;
    // Workgroups are dealt round-robin over the 8 XCDs (each with its own L2): give each XCD one
    // contiguous eighth of the Morton order, so that it keeps meeting the same part of the tree.
    // Within its eighth an XCD takes the wavefronts in the given order (longest first) if there is one.
    const uint32_t nblk = gridDim.x, q8 = nblk / 8, r8 = nblk % 8;
    const uint32_t xcd = blockIdx.x % 8, slot = blockIdx.x / 8;
    const uint32_t blk = perm ? perm[blockIdx.x] : xcd * q8 + min(xcd, r8) + slot;
    const uint32_t r = blk * WALK_THREADS + lane;
    uint32_t b = r < m ? (order ? order[r] : i_begin + r) : NO_NODE;
    const bool valid = b != NO_NODE;  // (a caller-given order may leave lanes empty)
    if (!valid) b = 0u;
    Tgt t;
    {
        const float4 p = posm[b];
        t.x = p.x; t.y = p.y; t.z = p.z;
        t.idx = (int)b;
        t.vx = t.vy = t.vz = t.vn = 0.0f;
        if (LAW == NBODY_LAW_WGSL_BIAS) {
            const float4 v = vel[b];
            t.vx = v.x; t.vy = v.y; t.vz = v.z;
            t.vn = sqrtf(fmaf(v.x, v.x, fmaf(v.y, v.y, v.z * v.z)));
        }
    }
    float ax = 0.0f, ay = 0.0f, az = 0.0f;
    uint32_t n_visit = 0, n_leaf = 0, n_agg = 0, n_wave = 0, max_top = 0;
    // (counting build, domain-decomposed tree) the domain this wavefront's targets belong to, as a bit:
    // every node it visits is marked with it
    uint32_t my_domain_bit = 0;
    if (COUNT && mark) {
        uint32_t d = 0;
        const uint32_t r0 = i_begin + blk * WALK_THREADS;
        while (d + 1 < n_dom && dom_off[d + 1] <= r0) ++d;
        my_domain_bit = 1u << d;
    }
    int top = 0;
    const unsigned long long all = __ballot(valid);
    if (all) {
        stack[0] = make_uint4(0u, 1u, (uint32_t)all, (uint32_t)(all >> 32));  // the root as a one-node group
        top = 1;
    }
    constexpr int BATCH = QUAD ? WALK_BATCH / 2 : WALK_BATCH;  // with quadrupoles the moments ride along
    // Where the term is cheap (PLUMMER) it is written twice, without a select where every lane of the
    // group takes the record; the longer terms are written once.
    constexpr bool TWO_FORMS = LAW == NBODY_LAW_PLUMMER && !QUAD;
    while (top > 0) {
        --top;
        const uint4 e = stack[top];
        uint32_t first = __builtin_amdgcn_readfirstlane(e.x);
        uint32_t cnt = __builtin_amdgcn_readfirstlane(e.y);
        const uint32_t mlo = (uint32_t)__builtin_amdgcn_readfirstlane(e.z);
        const uint32_t mhi = (uint32_t)__builtin_amdgcn_readfirstlane(e.w);
        const unsigned long long mask = ((unsigned long long)mhi << 32) | (unsigned long long)mlo;
        if (cnt > (uint32_t)BATCH) {  // the rest of a long group (a leaf list) goes back where it came from
            stack[top] = make_uint4(first + BATCH, cnt - BATCH, mlo, mhi);
            ++top;
            cnt = BATCH;
        }
        // All records of the batch are fetched before the first use, from consecutive addresses
        // whether the group has that many or not (the array is padded).
        QNode c[BATCH];
        QMoment qm[QUAD ? BATCH : 1];
        const QNode* group = q + first;
#pragma unroll
        for (int k = 0; k < BATCH; ++k) c[k] = group[k];
        if (QUAD) {
            const QMoment* gq = reinterpret_cast<const QMoment*>(quad) + first;
#pragma unroll
            for (int k = 0; k < BATCH; ++k) qm[k] = gq[k];
        }
#pragma unroll
        for (int k = 0; k < BATCH; ++k) pin_record(c[k]);
        if (QUAD) {
#pragma unroll
            for (int k = 0; k < BATCH; ++k) pin_moment(qm[k]);
        }
        __builtin_assume(cnt >= 1);
        n_wave += cnt;
        // (counting build) wavefront visits by the number of lanes they are made for, in 16 steps of 4
        if (COUNT && lane == 0) atomicAdd(&counters[8 + ((int)__popcll(mask) - 1) / 4], (unsigned long long)cnt);
        // Inside this branch EXEC is the group's lane set.  The stack pointer is advanced by those
        // lanes only (to the compiler it is a per-lane value there) and read back from one of them.
        int top_v = top;
        if (__builtin_amdgcn_inverse_ballot_w64(mask)) {
            auto rec = [&](const int k) __attribute__((always_inline)) {
                const QNode nd = c[k];
                if (COUNT) ++n_visit;
                if (COUNT && mark && lane == (int)__builtin_ctzll(mask)) atomicOr(&mark[first + k], my_domain_bit);
                const float dx = nd.cx - t.x, dy = nd.cy - t.y, dz = nd.cz - t.z;
                // PLUMMER: the softened squared distance, which the thresholds are prepared for
                const float r2 = LAW == NBODY_LAW_PLUMMER ? fmaf(dx, dx, fmaf(dy, dy, fmaf(dz, dz, eps2)))
                                                          : fmaf(dx, dx, fmaf(dy, dy, dz * dz));
                // src/nbodybh.wgsl:104-106 by the node's squared-distance thresholds; a LEAFBODY record
                // (a direct interaction, src/nbodybh.wgsl:76-102) carries thresholds below zero.
                // `off`: lanes of the group that do not take the record (select) -- none where !select
                auto term = [&](bool select, unsigned long long off) {
                    float ri = 0.0f;  // QUAD (PLUMMER law only): 1/R shared by the monopole and quadrupole terms
                    float sc;
                    if (LAW == NBODY_LAW_PLUMMER) {  // src/nbody.wgsl:56-65; r2 is softened already
                        ri = WALK_PRECISE ? rsq_refined(r2) : __builtin_amdgcn_rsqf(r2);
                        sc = (nd.mass * ri) * (ri * ri);
                    } else {
                        sc = pair_scale<LAW, WALK_PRECISE>(t, dx, dy, dz, r2, nd.mass, eps2);
                    }
                    const bool skip = select && __builtin_amdgcn_inverse_ballot_w64(off);
                    const float w = skip ? 0.0f : sc;
                    ax = fmaf(w, dx, ax);
                    ay = fmaf(w, dy, ay);
                    az = fmaf(w, dz, az);
                    if (QUAD && qn_type(nd) != NBODY_NODE_LEAFBODY) {
                        // quadrupole term of an accepted cell (d = centre - target, R^2 = r^2 + eps2):
                        // -Q d / R^5 + 5/2 (d.Q d) d / R^7
                        const QMoment& mo = qm[QUAD ? k : 0];
                        const float qxx = mo.xx, qxy = mo.xy, qxz = mo.xz, qyy = mo.yy, qyz = mo.yz, qzz = mo.zz;
                        const float ri2 = ri * ri;
                        const float ri5 = (ri2 * ri2) * ri, ri7 = ri5 * ri2;
                        const float ux = fmaf(qxx, dx, fmaf(qxy, dy, qxz * dz));
                        const float uy = fmaf(qxy, dx, fmaf(qyy, dy, qyz * dz));
                        const float uz = fmaf(qxz, dx, fmaf(qyz, dy, qzz * dz));
                        const float dqd = fmaf(dx, ux, fmaf(dy, uy, dz * uz));
                        const float c1 = skip ? 0.0f : -ri5, c2 = skip ? 0.0f : 2.5f * dqd * ri7;
                        ax = fmaf(c1, ux, fmaf(c2, dx, ax));
                        ay = fmaf(c1, uy, fmaf(c2, dy, ay));
                        az = fmaf(c1, uz, fmaf(c2, dz, az));
                    }
                };
                // The group's lanes that do not clearly accept the record: src/nbodybh.wgsl:104-106 by
                // the node's squared-distance thresholds; a LEAFBODY record (a direct interaction,
                // src/nbodybh.wgsl:76-102) carries thresholds below zero, which every lane passes.
                const unsigned long long rest = __ballot(!(r2 > nd.r2_accept));
                // The lanes among them that open it: clearly to be opened, or inside the band between
                // the thresholds, where the reference's own expression decides.
                auto opening = [&]() {
                    // A LEAFBODY record is never opened (src/nbodybh.wgsl:76-102 interacts with it whatever
                    // the distance); lanes reach this point for one only when r2 is NaN, and then take the
                    // term, NaN and all, as the reference does.
                    if (qn_type(nd) == NBODY_NODE_LEAFBODY) return 0ull;
                    unsigned long long open = rest;
                    const unsigned long long unsure = rest & lanes(!(r2 < nd.r2_open));
                    if (unsure) {
                        const float rg = range[first + k];
                        const float rr = sqrtf(((dx * dx) + (dy * dy)) + (dz * dz));  // distance(pos, com)
                        open &= ~(unsure & lanes((rg / rr) < theta));
                    }
                    if (open && qn_count(nd)) {  // (an empty group is never pushed: a popped group has >= 1 record)
                        // INTERIOR: its children; LEAFLIST: its members (LEAFBODY records, src/nbodybh.wgsl:137-146)
                        if (top_v < WALK_STACK) {
                            stack[top_v] = make_uint4(nd.first, qn_count(nd), (uint32_t)open, (uint32_t)(open >> 32));
                            ++top_v;
                        } else {
                            *err_flag = 2;  // (a tree deeper than any the build makes: uploaded)
                        }
                    }
                    return open;
                };
                auto count_taken = [&](unsigned long long open) {
                    if (COUNT && !__builtin_amdgcn_inverse_ballot_w64(open)) {
                        if (qn_type(nd) == NBODY_NODE_LEAFBODY) ++n_leaf;
                        else ++n_agg;
                    }
                };
                if (TWO_FORMS) {
                    // (the term is evaluated whether or not a lane takes it: the test costs more)
                    if (rest == 0) {
                        count_taken(0ull);
                        term(false, 0ull);
                    } else {
                        const unsigned long long open = opening();
                        count_taken(open);
                        term(true, open);
                    }
                } else {
                    unsigned long long open = 0;
                    if (rest != 0) open = opening();
                    count_taken(open);
                    if (open != mask) term(true, open);
                }
            };
#pragma unroll
            for (int k = 0; k < BATCH; ++k) {
                if ((uint32_t)k >= cnt) break;
                rec(k);
            }
        }
        top = __builtin_amdgcn_readlane(top_v, (int)__builtin_ctzll(mask));
        if (COUNT) max_top = max(max_top, (uint32_t)top);  // pushes only grow the stack within a batch
    }
    if (valid) acc_out[out_by_rank ? i_begin + r : b] = make_float4(G * ax, G * ay, G * az, 0.0f);
    if (wave_cost && lane == 0) wave_cost[blk] = n_wave;
    if (COUNT) {
        atomicAdd(&counters[0], (unsigned long long)n_visit);
        atomicAdd(&counters[1], (unsigned long long)n_leaf);
        atomicAdd(&counters[2], (unsigned long long)n_agg);
        if (lane == 0) atomicAdd(&counters[3], (unsigned long long)n_wave);
        atomicMax(&counters[4], (unsigned long long)max_top);
    }
}

// Launch order for the next traversal of the same targets: workgroup b runs on XCD b % 8 and is the
// (b / 8)-th that XCD starts; each XCD's contiguous share of the wavefronts is sorted by the node
// visits they made last time, longest first, so that the wavefronts still running when the others
// have finished are the short ones.  One workgroup per XCD; bitonic sort in LDS.
// Up to N = 2M targets: beyond that the last wavefronts are a small part of the time, and taking the
// wavefronts out of Morton order costs more in L2 misses than it saves (N = 16M: 27.3 against 23.4 ms).
constexpr int ORDER_MAX = 4096;  // wavefronts per XCD share; keys hold 15 index bits + 17 cost bits
__global__ __launch_bounds__(1024) void k_walk_order(const uint32_t* __restrict__ cost, uint32_t nblk, uint32_t* __restrict__ perm) {
    extern __shared__ uint32_t key[];  // power of two >= the share
    const uint32_t x = blockIdx.x, q8 = nblk / 8, r8 = nblk % 8;
    const uint32_t first = x * q8 + min(x, r8), cnt = q8 + (x < r8 ? 1u : 0u);
    uint32_t p2 = 1;
    while (p2 < cnt) p2 <<= 1;
    for (uint32_t i = threadIdx.x; i < p2; i += 1024)  // ascending keys = descending cost, ties in index order
        key[i] = i < cnt ? ((0x1ffffu - min(cost[first + i], 0x1ffffu)) << 15) | i : 0xffffffffu;
    __syncthreads();
    for (uint32_t k = 2; k <= p2; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t i = threadIdx.x; i < p2; i += 1024) {
                const uint32_t l = i ^ j;
                if (l > i) {
                    const uint32_t a = key[i], b = key[l];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) {
                        key[i] = b;
                        key[l] = a;
                    }
                }
            }
            __syncthreads();
        }
    }
    for (uint32_t i = threadIdx.x; i < cnt; i += 1024) perm[x + 8 * i] = first + (key[i] & 0x7fffu);
}

// theta == 0 gives infinite (or, for a zero range, NaN) thresholds: nothing is accepted by
// them and the reference's expression, evaluated in the band, opens every node as it should.
void domain_marks(Octree* T, uint32_t** mark, const uint32_t** dom_off, uint32_t* n_dom);

int env_int_or(const char* name, int dflt) {
    const char* v = std::getenv(name);
    return (v && *v) ? std::atoi(v) : dflt;
}

MacScale mac_scale(const nbody_ctx* ctx) {
    const float th2 = ctx->cfg.theta * ctx->cfg.theta;
    const float soft = ctx->cfg.law == NBODY_LAW_PLUMMER ? ctx->cfg.eps2 : 0.0f;
    return MacScale{1.0f / (th2 * (1.0f + 4e-6f)), 1.0f / (th2 * (1.0f - 4e-6f)), soft * (1.0f - 1e-6f), soft * (1.0f + 1e-6f)};
}

// Grows the node arrays.  They are released first; if one of the new allocations then fails the
// context is left WITHOUT a tree (have_tree false, capacity 0), so nothing traverses freed arrays.
int ensure_nodes(nbody_ctx* ctx, Octree* T, uint64_t need) {
    if (need <= T->cap_nodes) return NBODY_OK;
    uint64_t cap = std::max<uint64_t>(need + need / 4 + 1024, 2 * T->n + 1024);
    if (cap > 0xfffffff0ull) return nb_fail(ctx, NBODY_ERR_CAPACITY, "octree needs more than 2^32 nodes");
    void** arrays[] = {(void**)&T->nodes, (void**)&T->node_head, (void**)&T->bmin, (void**)&T->bmax, (void**)&T->range, (void**)&T->quad};
    for (void** a : arrays) {
        if (*a) (void)hipFree(*a);
        *a = nullptr;
    }
    T->cap_nodes = 0;
    T->have_tree = false;
    // the traversal fetches whole batches: pad so that the last group's batch stays inside
    hipError_t e = hipMalloc(&T->nodes, (cap + WALK_BATCH) * sizeof(QNode));
    if (e == hipSuccess) e = hipMalloc(&T->range, cap * sizeof(float));
    if (e == hipSuccess && ctx->cfg.quadrupole) e = hipMalloc(&T->quad, (cap + WALK_BATCH) * 8 * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(&T->node_head, cap * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc(&T->bmin, cap * sizeof(float4));
    if (e == hipSuccess) e = hipMalloc(&T->bmax, cap * sizeof(float4));
    if (e != hipSuccess) {
        (void)hipGetLastError();
        for (void** a : arrays) {
            if (*a) (void)hipFree(*a);
            *a = nullptr;
        }
        return nb_fail(ctx, NBODY_ERR_NOMEM, std::string("octree nodes: ") + hipGetErrorString(e));
    }
    T->cap_nodes = cap;
    return NBODY_OK;
}

int check_err_flag(nbody_ctx* ctx, Octree* T, const char* what) {
    uint32_t h = 0;
    NB_HIP(ctx, hipMemcpyAsync(&h, T->err_flag, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    NB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (h) {
        NB_HIP(ctx, hipMemsetAsync(T->err_flag, 0, sizeof(uint32_t), ctx->stream));
        return nb_fail(ctx, NBODY_ERR_CAPACITY,
                       std::string(what) + (h == 2 ? ": traversal stack overflow"
                                            : h == 3 ? ": children of a node are not consecutive"

                                                     : ": node capacity exceeded"));
    }
    return NBODY_OK;
}

template <int LAW>
void launch_walk(nbody_ctx* ctx, Octree* T, const uint32_t* order, uint32_t m, const float4* posm, const float4* vel,
                 float4* acc_out, uint32_t i_begin, int out_by_rank) {
    dim3 grid((m + WALK_THREADS - 1) / WALK_THREADS);
    const bool quad = T->quad != nullptr && T->built;  // uploaded trees carry no moments
    // The order made from the previous traversal of the same target list (the bodies of a
    // wavefront change little from step to step); none for a first traversal or a large grid.
    const bool ordered = grid.x <= 8 * ORDER_MAX && grid.x >= 64;
    const uint32_t* perm = ordered && T->perm_nblk == grid.x && T->perm_order == order ? T->wave_perm : nullptr;
    uint32_t* cost = ordered ? T->wave_cost : nullptr;
    // a counting traversal of a domain-decomposed tree marks every visited node with its visitors' domains
    uint32_t* mark = nullptr;
    const uint32_t* dom_off = nullptr;
    uint32_t n_dom = 0;
    if (ctx->counting && T->dom && T->built) domain_marks(T, &mark, &dom_off, &n_dom);
#define NB_WALK(COUNT, QUAD)                                                                                              \
    hipLaunchKernelGGL((k_walk<LAW, COUNT, QUAD>), grid, dim3(WALK_THREADS), 0, ctx->stream, T->nodes, order, m, i_begin, posm, \
                       vel, acc_out, ctx->cfg.G, ctx->cfg.eps2, ctx->cfg.theta, T->range,                                 \
                       (unsigned long long*)T->walk_counters, T->err_flag, out_by_rank, T->quad, perm, cost, mark, dom_off, n_dom)
    if (ctx->counting) {
        if (quad) { NB_WALK(true, true); }
        else { NB_WALK(true, false); }
    } else {
        if (quad) { NB_WALK(false, true); }
        else { NB_WALK(false, false); }
    }
#undef NB_WALK
    if (ordered) {
        // refreshed now and then: the visit counts drift slowly
        if (!perm || ++T->perm_age >= 8) {
            uint32_t p2 = 1;
            while (p2 < (grid.x + 7) / 8) p2 <<= 1;
            hipLaunchKernelGGL(k_walk_order, dim3(8), dim3(1024), p2 * sizeof(uint32_t), ctx->stream, T->wave_cost, grid.x,
                               T->wave_perm);
            T->perm_nblk = grid.x;
            T->perm_order = order;
            T->perm_age = 0;
        }
    } else {
        T->perm_nblk = 0;
    }
}

}  // namespace

// The per-body arrays and scan workspaces of one tree, for up to n_cap bodies.
static int tree_alloc(nbody_ctx* ctx, Octree* T, uint64_t n_cap, uint32_t D) {
    T->n = T->n_cap = n_cap;
    T->D = D;
    const uint64_t n = n_cap;
    T->rs_nb = (uint32_t)((n + RS_TILE - 1) / RS_TILE);
    T->pf_nb = (uint32_t)((n + 255) / 256);
    for (int s = 0; s < 2; ++s) {
        NB_HIP(ctx, hipMalloc(&T->keys[s], n * sizeof(uint64_t)));
        NB_HIP(ctx, hipMalloc(&T->idx[s], n * sizeof(uint32_t)));
    }
    NB_HIP(ctx, hipMalloc(&T->lo, n));
    NB_HIP(ctx, hipMalloc(&T->top, n));
    NB_HIP(ctx, hipMalloc(&T->flags, n));
    NB_HIP(ctx, hipMalloc(&T->body_node, n * sizeof(uint32_t)));
    NB_HIP(ctx, hipMalloc(&T->rs_counts, (size_t)256 * T->rs_nb * sizeof(uint32_t)));
    NB_HIP(ctx, hipMalloc(&T->row_total, 256 * sizeof(uint32_t)));
    NB_HIP(ctx, hipMalloc(&T->blk, (size_t)NCNT * T->pf_nb * sizeof(uint32_t)));
    NB_HIP(ctx, hipMalloc(&T->lvl_total, NCNT * sizeof(uint32_t)));
    NB_HIP(ctx, hipMalloc(&T->lvl_start, (NCNT + 1) * sizeof(uint32_t)));
    NB_HIP(ctx, hipHostMalloc(&T->h_lvl_start, (NCNT + 1) * sizeof(uint32_t), hipHostMallocDefault));
    NB_HIP(ctx, hipEventCreateWithFlags(&T->lvl_ready, hipEventDisableTiming));
    NB_HIP(ctx, hipMalloc(&T->walk_counters, NWALK_COUNTERS * sizeof(uint64_t)));
    NB_HIP(ctx, hipMalloc(&T->err_flag, sizeof(uint32_t)));
    NB_HIP(ctx, hipMemsetAsync(T->err_flag, 0, sizeof(uint32_t), ctx->stream));
    return ensure_nodes(ctx, T, 2 * n + 1024);
}

static void domains_free(Octree* T);

static void tree_free(Octree* T) {
    if (!T) return;
    domains_free(T);
    for (int s = 0; s < 2; ++s) {
        if (T->keys[s]) (void)hipFree(T->keys[s]);
        if (T->idx[s]) (void)hipFree(T->idx[s]);
    }
    for (int k = 0; k < 2; ++k) {
        if (T->hkeys[k]) (void)hipFree(T->hkeys[k]);
        if (T->hidx[k]) (void)hipFree(T->hidx[k]);
    }
    void* ptrs[] = {T->lo, T->top, T->flags, T->body_node, T->rs_counts, T->row_total, T->blk,
                    T->lvl_total, T->lvl_start, T->nodes, T->node_head, T->bmin, T->bmax, T->range, T->quad, T->export_buf, T->shard_order, T->shard_blk,
                    T->walk_counters, T->err_flag, T->wave_cost, T->wave_perm, T->user_order};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    if (T->h_lvl_start) (void)hipHostFree(T->h_lvl_start);
    if (T->lvl_ready) (void)hipEventDestroy(T->lvl_ready);
    delete T;
}

int nb_tree_create(nbody_ctx* ctx) {
    Octree* T = new Octree;
    ctx->tree = T;
    // max_depth 0: the uncapped builder of src/octree.rs, reproduced as far as 16 levels reach
    T->octree_rs = ctx->cfg.max_depth == 0;
    int r = tree_alloc(ctx, T, ctx->n, ctx->cfg.max_depth ? ctx->cfg.max_depth : 16);
    if (r) return r;
    NB_HIP(ctx, hipMalloc(&T->wave_cost, 8 * ORDER_MAX * sizeof(uint32_t)));
    NB_HIP(ctx, hipMalloc(&T->wave_perm, 8 * ORDER_MAX * sizeof(uint32_t)));
    if (ctx->i_count < ctx->n) {
        NB_HIP(ctx, hipMalloc(&T->shard_order, ctx->i_count * sizeof(uint32_t)));
        NB_HIP(ctx, hipMalloc(&T->shard_blk, ((ctx->n + CO_TILE - 1) / CO_TILE + 1) * sizeof(uint32_t)));
    }
    return NBODY_OK;
}

void nb_tree_destroy(nbody_ctx* ctx) {
    tree_free(ctx->tree);
    ctx->tree = nullptr;
}

// What the top tree over the domains' cells takes from the domains' own trees: which of its entries
// stand for whole cells (k_prefix), and the records, boxes and ranges its deepest nodes are given before
// its aggregates are made (one per entry, by entry index).
struct TopPatch {
    const uint8_t* multi;
    const QNode* rec;
    const float4* bmin;
    const float4* bmax;
    const float* range;
};

__global__ void k_top_patch(const uint32_t* __restrict__ idx, const uint32_t* __restrict__ body_node, uint32_t n, TopPatch p,
                            QNode* __restrict__ nodes, float4* __restrict__ bmin, float4* __restrict__ bmax,
                            float* __restrict__ range, uint32_t cap_nodes) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t id = body_node[i], e = idx[i];
    if (id >= cap_nodes) return;
    QNode r = p.rec[e];
    // A lone body may sit above the deepest level, where its record's `first` tells the node before it
    // where its own children end (k_aggregate): that stays.  The deepest level has no such reader.
    if (!p.multi[e]) r.first = nodes[id].first;
    nodes[id] = r;
    bmin[id] = p.bmin[e];
    bmax[id] = p.bmax[e];
    range[id] = p.range[e];
}

// One tree over the n bodies of posm_in (n <= T->n_cap), into T's arrays.
static int build_into(nbody_ctx* ctx, Octree* T, const float4* posm_in, uint32_t n, const TopPatch* patch) {
    const MacScale mac = mac_scale(ctx);
    if (n == 0 || n > T->n_cap) return nb_fail(ctx, NBODY_ERR_CAPACITY, "octree build: more bodies than the tree was made for");
    T->n = n;
    T->rs_nb = (n + RS_TILE - 1) / RS_TILE;
    T->pf_nb = (n + 255) / 256;
    const int D = (int)T->D;
    hipStream_t st = ctx->stream;
    hipLaunchKernelGGL(k_morton, dim3((n + 255) / 256), dim3(256), 0, st, posm_in, n, ctx->cfg.world, D, T->keys[0],
                       T->idx[0]);
    NB_HIP(ctx, hipGetLastError());
    int cur = 0;
    const int bits = 3 * D;
    for (int shift = 0; shift < bits; shift += 8) {
        hipLaunchKernelGGL(k_rs_hist<uint64_t>, dim3(T->rs_nb), dim3(RS_THREADS), 0, st, T->keys[cur], n, shift, T->rs_counts, T->rs_nb);
        hipLaunchKernelGGL(k_rowscan, dim3(256), dim3(256), 0, st, T->rs_counts, T->rs_nb, T->row_total);
        hipLaunchKernelGGL(k_rs_scatter<uint64_t>, dim3(T->rs_nb), dim3(RS_THREADS), 0, st, T->keys[cur], T->idx[cur], T->keys[cur ^ 1],
                           T->idx[cur ^ 1], n, shift, T->rs_counts, T->row_total, T->rs_nb);
        NB_HIP(ctx, hipGetLastError());
        cur ^= 1;
    }
    T->sorted = cur;
    hipLaunchKernelGGL(k_prefix, dim3(T->pf_nb), dim3(256), 0, st, T->keys[cur], n, D, ctx->cfg.leaflist_mode, T->lo, T->top,
                       T->flags, T->blk, T->pf_nb, T->idx[cur], patch ? patch->multi : (const uint8_t*)nullptr);
    hipLaunchKernelGGL(k_rowscan, dim3(NCNT), dim3(256), 0, st, T->blk, T->pf_nb, T->lvl_total);
    hipLaunchKernelGGL(k_scan_small, dim3(1), dim3(256), 0, st, T->lvl_total, T->lvl_start, NCNT);
    NB_HIP(ctx, hipGetLastError());
    // The host needs the level sizes for the launches below.  The node emission does not: it is
    // enqueued behind the copy and runs while the host waits for the copy alone (emission into
    // the present capacity drops what does not fit; only then is it repeated after growing).
    NB_HIP(ctx, hipMemcpyAsync(T->h_lvl_start, T->lvl_start, (NCNT + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    NB_HIP(ctx, hipEventRecord(T->lvl_ready, st));
    auto emit = [&]() {
        hipLaunchKernelGGL(k_emit, dim3(T->pf_nb), dim3(256), 0, st, T->keys[cur], T->idx[cur], posm_in, n, D, T->lo, T->top,
                           T->flags, T->blk, T->pf_nb, T->lvl_start, T->nodes, T->bmin, T->bmax, T->range, T->node_head,
                           T->body_node, (uint32_t)T->cap_nodes, T->err_flag);
    };
    emit();
    NB_HIP(ctx, hipGetLastError());
    NB_HIP(ctx, hipEventSynchronize(T->lvl_ready));
    const uint64_t total = T->h_lvl_start[NCNT];
    if (total > T->cap_nodes) {
        NB_HIP(ctx, hipStreamSynchronize(st));
        NB_HIP(ctx, hipMemsetAsync(T->err_flag, 0, sizeof(uint32_t), st));  // raised by the emission that did not fit
        int r = ensure_nodes(ctx, T, total);
        if (r) return r;
        emit();
        NB_HIP(ctx, hipGetLastError());
    }
    T->n_nodes = total;
    if (T->octree_rs && T->h_lvl_start[LIST_ROW + 1] > T->h_lvl_start[LIST_ROW]) {
        // list members exist: two bodies share a level-16 cell, where the uncapped builder would
        // descend further (or, for coincident bodies, never end)
        NB_HIP(ctx, hipStreamSynchronize(st));
        T->have_tree = false;
        return nb_fail(ctx, NBODY_ERR_CAPACITY,
                       "octree.rs variant (max_depth 0): two bodies share a level-16 cell, beyond the 16 levels reproduced");
    }
    if (T->h_lvl_start[LIST_ROW + 1] > T->h_lvl_start[LIST_ROW] || T->h_lvl_start[D + 1] > T->h_lvl_start[D]) {
        hipLaunchKernelGGL(k_leaflist, dim3((n + 255) / 256), dim3(256), 0, st, T->keys[cur], T->idx[cur], posm_in, n,
                           ctx->cfg.leaflist_mode, T->flags, T->body_node, T->nodes, T->bmin, T->bmax, T->range, T->node_head,
                           (uint32_t)T->cap_nodes, mac);
        NB_HIP(ctx, hipGetLastError());
    }
    if (patch) {  // the top tree's deepest nodes are the domains' own nodes
        hipLaunchKernelGGL(k_top_patch, dim3((n + 255) / 256), dim3(256), 0, st, T->idx[cur], T->body_node, n, *patch, T->nodes,
                           T->bmin, T->bmax, T->range, (uint32_t)T->cap_nodes);
        NB_HIP(ctx, hipGetLastError());
    }
    for (int d = D - 1; d >= 0; --d) {
        const uint32_t first = T->h_lvl_start[d], count = T->h_lvl_start[d + 1] - first;
        if (!count) continue;
        // the children of level d are the nodes of level d + 1
        hipLaunchKernelGGL(k_aggregate, dim3((count + 255) / 256), dim3(256), 0, st, T->nodes, T->bmin, T->bmax, T->range,
                           first, count, T->h_lvl_start[d + 2], mac);
    }
    if (T->quad) {
        for (int d = D; d >= 0; --d) {
            const uint32_t first = T->h_lvl_start[d], count = T->h_lvl_start[d + 1] - first;
            if (!count) continue;
            hipLaunchKernelGGL(k_quadrupole, dim3((count + 255) / 256), dim3(256), 0, st, T->nodes, T->quad, first, count,
                               ctx->cfg.leaflist_mode);
        }
    }
    NB_HIP(ctx, hipGetLastError());
    T->have_tree = true;
    T->built = true;
    T->export_is_upload = false;
    ++T->horder_age;
    return NBODY_OK;
}


// ---- domain-decomposed build (no counterpart in the reference, which builds one tree on one host:
// src/octree_maxdepth.rs:26-61) ---------------------------------------------------------------
// The bodies are dealt to G domains, each a run of whole depth-L cells in Morton order holding about
// n / G bodies.  Every domain builds the tree of ITS bodies with the ordinary builder (work in
// proportion to n / G: on several GPUs, one domain each).  Below depth L a cell lies in one domain, so
// its subtree there is the subtree of the one tree over all bodies, bit for bit; the cells above depth
// L are rebuilt from the domains' depth-L cells (and the bodies that are alone in a larger cell) by the
// same builder with those as its "bodies" -- the top tree, a few thousand nodes, the only part every
// domain must hold.  The traversal then runs over the top tree with the domains' trees hung below it.
constexpr int MAX_DOMAINS = 32;
constexpr int DOMAIN_TOP_LEVELS_MAX = 6;

namespace {
struct LetBox {
    float4 lo, hi;
};
struct LetEntry {  // an entry of the top tree as it travels: what k_top_entries writes, in one record
    float4 posm;
    QNode rec;
    float4 bmin, bmax;
    float range;
    uint32_t multi;
    uint32_t pad[2];
};
static_assert(sizeof(LetEntry) == 96, "entry record layout");
}  // namespace

struct Domains {
    uint32_t G = 0, L = 0, cells = 0;
    uint32_t* ckey = nullptr;        // [n] depth-L cell of each body
    uint32_t* cell_count = nullptr;  // [cells]
    uint8_t* cell_dom = nullptr;     // [cells]
    uint32_t* dkey[2] = {nullptr, nullptr};  // [n] one radix pass: key = domain ...
    uint32_t* dval[2] = {nullptr, nullptr};  // ... value = body; dval[1] ends up grouped by domain, index order kept
    uint32_t* d_off = nullptr;       // [G + 1] first body of each domain in dval[1]
    uint32_t* h_off = nullptr;       // pinned copy
    float4* posm_k = nullptr;        // [n] the domains' bodies, domain after domain
    Octree* tree[MAX_DOMAINS] = {};
    Octree* top = nullptr;
    uint32_t top_cap = 0, top_node_cap = 0;
    uint64_t share = 0;              // bodies a domain's tree is first given room for
    uint32_t* n_entries = nullptr;   // device counter, and its pinned copy
    uint32_t* h_entries = nullptr;
    float4* e_posm = nullptr;
    uint8_t* e_multi = nullptr;
    QNode* e_rec = nullptr;
    float4* e_bmin = nullptr;
    float4* e_bmax = nullptr;
    float* e_range = nullptr;
    uint32_t* mark = nullptr;        // [node capacity] which domains' targets visited a node (counting traversal)
    uint64_t mark_cap = 0;
    unsigned long long* imported = nullptr;  // [MAX_DOMAINS + 1] device
    uint64_t node_base[MAX_DOMAINS + 1] = {};
    hipEvent_t ev[MAX_DOMAINS + 3][2] = {};
    hipEvent_t ev_export[MAX_DOMAINS][2] = {};  // choosing and packing a domain's export
    nbody_domain_stats_t stats{};
    // essential-tree exchange: what a domain hands to the others instead of its whole tree
    bool export_timed[MAX_DOMAINS] = {};
    int view = -1;                    // >= 0: the merged tree is assembled as that domain's GPU would hold it
    uint8_t* let_flags = nullptr;     // [let_cap] per node of the exporting domain's tree: exported
    uint32_t* let_newid = nullptr;    // [let_cap] ... and its index among the exported nodes
    uint32_t* let_blk = nullptr;      // per-tile counts of the compaction, then their scan
    uint64_t let_cap = 0;
    QNode* send_nodes = nullptr;      // [send_cap] the exported nodes, children's indices among themselves
    float* send_range = nullptr;
    uint64_t send_cap = 0;
    LetEntry* send_entries = nullptr;    // [top_cap] this domain's entries of the top tree, as sent
    LetEntry* recv_entries = nullptr;    // [top_cap] one other domain's, as received
    LetBox* boxes = nullptr;          // [MAX_DOMAINS][LET_BOXES] where the domains' bodies are
    uint32_t* let_counts = nullptr;   // [2 * MAX_DOMAINS] per domain {exported nodes, entries}, and its pinned copy
    uint32_t* h_let_counts = nullptr;
    float4* vel_k = nullptr;          // [n] velocities of this rank's domain, gathered like posm_k (the reference's law reads them)
    float4* acc_grouped = nullptr;    // [n] accelerations, domain after domain in the order of dval[1]
};

namespace {

// The depth-L cell of every body (the reference's descent, as k_morton), and the cells' populations
// among every `stride`-th body: a sample is all the balancing needs, and counting every body would
// put millions of atomic additions on the few cells at the centre of a cluster (15 ms at N = 16M).
__global__ void k_cellkey(const float4* __restrict__ posm, uint32_t n, float world, int L, uint32_t stride,
                          uint32_t* __restrict__ ckey, uint32_t* __restrict__ cell_count) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 p = posm[i];
    float ext = world / 2.0f;
    float cx = ext, cy = ext, cz = ext;
    uint32_t key = 0;
    for (int l = 0; l < L; ++l) {
        ext = ext / 2.0f;
        uint32_t ci = 0;
        float wx = -1.0f, wy = -1.0f, wz = -1.0f;
        if (p.x >= cx) { ci |= 4u; wx = 1.0f; }
        if (p.y >= cy) { ci |= 2u; wy = 1.0f; }
        if (p.z >= cz) { ci |= 1u; wz = 1.0f; }
        key = (key << 3) | ci;
        cx = cx + (ext * wx);
        cy = cy + (ext * wy);
        cz = cz + (ext * wz);
    }
    ckey[i] = key;
    if (i % stride == 0) atomicAdd(&cell_count[key], 1u);
}

// One workgroup: the cells in Morton order; a cell goes to the domain its middle body (of the n counted)
// would fall in if the bodies were dealt evenly.  Domains are runs of whole cells: their sizes differ by
// about half a cell's bodies at most.
__global__ __launch_bounds__(1024) void k_cell_domains(const uint32_t* __restrict__ cell_count, uint32_t cells, uint32_t n,
                                                       uint32_t G, uint8_t* __restrict__ cell_dom) {
    __shared__ uint32_t part[1024];
    const uint32_t tid = threadIdx.x;
    const uint32_t per = (cells + 1023u) / 1024u, c0 = tid * per, c1 = min(cells, c0 + per);
    uint32_t sum = 0;
    for (uint32_t c = c0; c < c1; ++c) sum += cell_count[c];
    part[tid] = sum;
    __syncthreads();
    if (tid == 0) {
        uint32_t acc = 0;
        for (int t = 0; t < 1024; ++t) {
            const uint32_t v = part[t];
            part[t] = acc;
            acc += v;
        }
    }
    __syncthreads();
    uint32_t before = part[tid];
    for (uint32_t c = c0; c < c1; ++c) {
        const uint32_t cnt = cell_count[c];
        const uint32_t d = min(G - 1u, (uint32_t)(((unsigned long long)(before + cnt / 2u) * G) / n));
        cell_dom[c] = (uint8_t)d;
        before += cnt;
    }
}

// The domains' first bodies from the radix pass's digit totals (digit = domain).
__global__ void k_domain_offsets(const uint32_t* __restrict__ row_total, uint32_t G, uint32_t* __restrict__ d_off) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        uint32_t acc = 0;
        for (uint32_t d = 0; d < G; ++d) {
            d_off[d] = acc;
            acc += row_total[d];
        }
        d_off[G] = acc;
    }
}

__global__ void k_body_domain(const uint32_t* __restrict__ ckey, const uint8_t* __restrict__ cell_dom, uint32_t n,
                              uint32_t* __restrict__ dkey, uint32_t* __restrict__ dval) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    dkey[i] = cell_dom[ckey[i]];
    dval[i] = i;
}

__global__ void k_gather_bodies(const float4* __restrict__ posm, const uint32_t* __restrict__ body, uint32_t n,
                                float4* __restrict__ out) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) out[j] = posm[body[j]];
}

// A domain's nodes into the merged array: child indices become indices of the merged array.
__global__ void k_rebase(const QNode* __restrict__ src, const float* __restrict__ src_range, uint32_t count, uint32_t base,
                         QNode* __restrict__ dst, float* __restrict__ dst_range) {
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= count) return;
    QNode nd = src[id];
    if (qn_type(nd) != NBODY_NODE_LEAFBODY) nd.first += base;
    dst[base + id] = nd;
    dst_range[base + id] = src_range[id];
}

// The merged sorted order: a domain's sorted bodies under their indices among all bodies.
__global__ void k_global_order(const uint64_t* __restrict__ keys_k, const uint32_t* __restrict__ idx_k,
                               const uint32_t* __restrict__ body, uint32_t n, uint64_t* __restrict__ keys_out,
                               uint32_t* __restrict__ idx_out) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    keys_out[r] = keys_k[r];
    idx_out[r] = body[idx_k[r]];
}

// What a domain contributes to the top tree: its nodes at depth L, and the bodies that are alone in a
// larger cell of its own tree (LEAFBODY nodes above depth L).  The domain's nodes above depth L hold
// only its own share of their cells and are not used.
__global__ void k_top_entries(const QNode* __restrict__ nodes, const float4* __restrict__ bmin, const float4* __restrict__ bmax,
                              const float* __restrict__ range, const uint32_t* __restrict__ node_head,
                              const uint32_t* __restrict__ idx_sorted, const float4* __restrict__ posm_k,
                              const uint32_t* __restrict__ lvl_start, int L, uint32_t limit, uint32_t base,
                              uint32_t* __restrict__ n_entries, uint32_t cap, float4* __restrict__ e_posm,
                              uint8_t* __restrict__ e_multi, QNode* __restrict__ e_rec, float4* __restrict__ e_bmin,
                              float4* __restrict__ e_bmax, float* __restrict__ e_range, uint32_t* __restrict__ err_flag) {
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= limit) return;
    int d = 0;
    while (d < L && id >= lvl_start[d + 1]) ++d;
    QNode nd = nodes[id];
    const bool leaf = qn_type(nd) == NBODY_NODE_LEAFBODY;
    if (d < L && !leaf) return;
    const uint32_t e = atomicAdd(n_entries, 1u);
    if (e >= cap) {
        *err_flag = 1;
        return;
    }
    float4 pos = make_float4(nd.cx, nd.cy, nd.cz, nd.mass);
    if (!leaf) {
        pos = posm_k[idx_sorted[node_head[id]]];  // any body of the cell names it
        nd.first += base;
    }
    e_posm[e] = pos;
    e_multi[e] = leaf ? 0 : 1;
    e_rec[e] = nd;
    e_bmin[e] = leaf ? make_float4(nd.cx, nd.cy, nd.cz, 0.0f) : bmin[id];
    e_bmax[e] = leaf ? make_float4(nd.cx, nd.cy, nd.cz, 0.0f) : bmax[id];
    e_range[e] = leaf ? 0.0f : range[id];
}

// After a counting traversal: per domain, the nodes of OTHER domains' trees its targets visited (what
// it would have to be sent if every domain sat on a GPU of its own); slot G counts the top tree's nodes
// that any target visited.
__global__ void k_imported(const uint32_t* __restrict__ mark, uint32_t first, uint32_t count, uint32_t owner, uint32_t G,
                           unsigned long long* __restrict__ imported) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= count) return;
    const uint32_t bits = mark[first + r];
    if (!bits) return;
    if (owner >= G) {
        atomicAdd(&imported[G], 1ull);
        return;
    }
    for (uint32_t k = 0; k < G; ++k)
        if (k != owner && (bits >> k) & 1u) atomicAdd(&imported[k], 1ull);
}

// ---- essential trees --------------------------------------------------------------------------
// A domain on a GPU of its own needs, of another domain's tree, only the nodes its targets can reach:
// the children of every node that some target inside its bodies' bounding box could refuse.  The
// exporting domain decides that with the traversal's own thresholds against the nearest point of the
// box (a superset of what will be visited, by a margin far above the rounding of either side), numbers
// the exported nodes in their breadth-first order -- children stay consecutive, since a node's children
// are exported together or not at all -- and sends them with the child indices renumbered.  A node
// whose children stay behind is sent with a child count of zero: nothing can follow it.
constexpr int LET_ITEMS = 16, LET_TILE = 256 * LET_ITEMS;

// Where a domain's bodies are, for the other domains' export decisions: box 0 bounds them all, the
// others bound its bodies cell by cell at depth LET_BOX_DEPTH of its own tree (a run of cells in Morton
// order is no box: one stray cell on the far side of a neighbouring octant would make a single box
// swallow that octant's core).  Unused boxes are empty (nothing is near them).
constexpr int LET_BOX_DEPTH = 2, LET_BOXES = 80;  // 1 + up to 64 cells + lone bodies above them (<= 8 + 1)
__global__ __launch_bounds__(128) void k_let_box(const QNode* __restrict__ nodes, const float4* __restrict__ bmin,
                                                 const float4* __restrict__ bmax, const uint32_t* __restrict__ lvl_start,
                                                 uint32_t bodies, LetBox* __restrict__ out) {
    __shared__ uint32_t used;
    const uint32_t tid = threadIdx.x;
    if (tid == 0) used = 1;
    LetBox empty;
    empty.lo = make_float4(INFINITY, INFINITY, INFINITY, 0.0f);
    empty.hi = make_float4(-INFINITY, -INFINITY, -INFINITY, 0.0f);
    if (tid < LET_BOXES) out[tid] = empty;
    __syncthreads();
    if (!bodies) return;
    const uint32_t limit = min(lvl_start[LET_BOX_DEPTH + 1], (uint32_t)blockDim.x);  // (<= 73 nodes down to that depth)
    if (tid < limit) {
        const QNode nd = nodes[tid];
        const bool leaf = qn_type(nd) == NBODY_NODE_LEAFBODY;
        LetBox b;
        if (leaf) {
            b.lo = b.hi = make_float4(nd.cx, nd.cy, nd.cz, 0.0f);
        } else {
            b.lo = bmin[tid];
            b.hi = bmax[tid];
        }
        if (tid == 0) out[0] = b;
        if (leaf || tid >= lvl_start[LET_BOX_DEPTH]) {
            const uint32_t slot = atomicAdd(&used, 1u);
            if (slot < (uint32_t)LET_BOXES) out[slot] = b;
        }
    }
}

// Whether a target somewhere in the box might not accept the node.  soft: what the traversal adds to
// the squared distance before it compares (the softening under the PLUMMER law, else 0); the traversal
// accepts where r^2 [+ soft] > r2_accept.
__device__ __forceinline__ bool let_reach(const LetBox& b, const QNode& nd, float soft) {
    const float dx = fmaxf(fmaxf(b.lo.x - nd.cx, nd.cx - b.hi.x), 0.0f);
    const float dy = fmaxf(fmaxf(b.lo.y - nd.cy, nd.cy - b.hi.y), 0.0f);
    const float dz = fmaxf(fmaxf(b.lo.z - nd.cz, nd.cz - b.hi.z), 0.0f);
    const float d2 = dx * dx + dy * dy + dz * dz;
    return !(d2 * 0.9999f > (nd.r2_accept - soft) * 1.0001f + soft * 1e-5f);
}

// Nodes [first_id, first_id + count) of domain `self`: the children of those that a target in some
// other domain's boxes might not accept are flagged.
__global__ void k_let_mark(const QNode* __restrict__ nodes, uint32_t first_id, uint32_t count, const LetBox* __restrict__ boxes,
                           uint32_t G, uint32_t self, float soft, uint8_t* __restrict__ flags) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= count) return;
    const QNode nd = nodes[first_id + r];
    if (qn_type(nd) == NBODY_NODE_LEAFBODY) return;
    const uint32_t cnt = qn_count(nd);
    if (!cnt) return;
    bool open = false;
    for (uint32_t s = 0; s < G && !open; ++s) {
        if (s == self) continue;
        const LetBox* bs = boxes + (size_t)s * LET_BOXES;
        if (!let_reach(bs[0], nd, soft)) continue;
        for (int j = 1; j < LET_BOXES && !open; ++j) open = let_reach(bs[j], nd, soft);
    }
    if (open)
        for (uint32_t j = 0; j < cnt; ++j) flags[nd.first + j] = 1;
}

__global__ __launch_bounds__(256) void k_let_count(const uint8_t* __restrict__ flags, uint32_t lo_id, uint32_t n_nodes,
                                                   uint32_t* __restrict__ blk) {
    __shared__ uint32_t wsum[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t c = 0;
    for (int it = 0; it < LET_ITEMS; ++it) {
        const uint32_t id = lo_id + blockIdx.x * LET_TILE + it * 256 + tid;
        if (id < n_nodes) c += flags[id] ? 1u : 0u;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
    if (lane == 0) wsum[wave] = c;
    __syncthreads();
    if (tid == 0) blk[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// blk holds the exclusive scan of the tiles' counts: every exported node's index among the exported, in id order.
__global__ __launch_bounds__(256) void k_let_number(const uint8_t* __restrict__ flags, uint32_t lo_id, uint32_t n_nodes,
                                                    const uint32_t* __restrict__ blk, uint32_t* __restrict__ newid) {
    __shared__ uint32_t wsum[2][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t run = blk[blockIdx.x];
    for (int it = 0; it < LET_ITEMS; ++it) {
        const uint32_t id = lo_id + blockIdx.x * LET_TILE + it * 256 + tid;
        const bool keep = id < n_nodes && flags[id];
        const unsigned long long m = __ballot(keep);
        if (lane == 0) wsum[it & 1][wave] = (uint32_t)__popcll(m);
        __syncthreads();  // the other half of wsum is what the previous round may still be reading
        uint32_t off = run;
        for (int w = 0; w < wave; ++w) off += wsum[it & 1][w];
        off += (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        if (keep) newid[id] = off;
        run += wsum[it & 1][0] + wsum[it & 1][1] + wsum[it & 1][2] + wsum[it & 1][3];
    }
}

// A record as it is sent: children's index among the exported nodes, or no children at all.
__device__ __forceinline__ QNode let_record(QNode nd, const uint8_t* __restrict__ flags, const uint32_t* __restrict__ newid) {
    if (qn_type(nd) != NBODY_NODE_LEAFBODY) {
        if (qn_count(nd) && flags[nd.first]) {
            nd.first = newid[nd.first];
        } else {
            nd.first = 0;
            nd.count_type = qn_pack(0, qn_type(nd));
        }
    }
    return nd;
}

__global__ void k_let_pack(const QNode* __restrict__ nodes, const float* __restrict__ range, const uint8_t* __restrict__ flags,
                           const uint32_t* __restrict__ newid, uint32_t lo_id, uint32_t n_nodes, QNode* __restrict__ out_nodes,
                           float* __restrict__ out_range, uint32_t cap) {
    const uint32_t id = lo_id + blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= n_nodes || !flags[id]) return;
    const uint32_t o = newid[id];
    if (o >= cap) return;
    out_nodes[o] = let_record(nodes[id], flags, newid);
    out_range[o] = range[id];
}

// k_top_entries into travelling records.  flags / newid given: child indices as exported; null: the
// domain's own numbering (for the GPU that holds the whole domain).
__global__ void k_let_entries(const QNode* __restrict__ nodes, const float4* __restrict__ bmin, const float4* __restrict__ bmax,
                              const float* __restrict__ range, const uint32_t* __restrict__ node_head,
                              const uint32_t* __restrict__ idx_sorted, const float4* __restrict__ posm_k,
                              const uint32_t* __restrict__ lvl_start, int L, uint32_t limit, const uint8_t* __restrict__ flags,
                              const uint32_t* __restrict__ newid, uint32_t* __restrict__ n_entries, uint32_t cap,
                              LetEntry* __restrict__ out, uint32_t* __restrict__ err_flag) {
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= limit) return;
    int d = 0;
    while (d < L && id >= lvl_start[d + 1]) ++d;
    QNode nd = nodes[id];
    const bool leaf = qn_type(nd) == NBODY_NODE_LEAFBODY;
    if (d < L && !leaf) return;
    const uint32_t e = atomicAdd(n_entries, 1u);
    if (e >= cap) {
        *err_flag = 1;
        return;
    }
    LetEntry E;
    E.posm = leaf ? make_float4(nd.cx, nd.cy, nd.cz, nd.mass) : posm_k[idx_sorted[node_head[id]]];  // any body of the cell names it
    E.rec = flags ? let_record(nd, flags, newid) : nd;
    E.bmin = leaf ? make_float4(nd.cx, nd.cy, nd.cz, 0.0f) : bmin[id];
    E.bmax = leaf ? make_float4(nd.cx, nd.cy, nd.cz, 0.0f) : bmax[id];
    E.range = leaf ? 0.0f : range[id];
    E.multi = leaf ? 0u : 1u;
    E.pad[0] = E.pad[1] = 0u;
    out[e] = E;
}

// Received (or own) entries into the arrays the top tree is built from, child indices moved to where the
// sender's nodes were put.
__global__ void k_let_entries_unpack(const LetEntry* __restrict__ in, uint32_t m, uint32_t base, uint32_t e0,
                                     float4* __restrict__ e_posm, uint8_t* __restrict__ e_multi, QNode* __restrict__ e_rec,
                                     float4* __restrict__ e_bmin, float4* __restrict__ e_bmax, float* __restrict__ e_range) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const LetEntry E = in[i];
    QNode nd = E.rec;
    if (qn_type(nd) != NBODY_NODE_LEAFBODY && qn_count(nd)) nd.first += base;
    e_posm[e0 + i] = E.posm;
    e_multi[e0 + i] = (uint8_t)E.multi;
    e_rec[e0 + i] = nd;
    e_bmin[e0 + i] = E.bmin;
    e_bmax[e0 + i] = E.bmax;
    e_range[e0 + i] = E.range;
}

// Exported nodes where they were received (arr + base): child indices become indices of arr.
__global__ void k_let_settle(QNode* arr, uint32_t base, uint32_t count) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    QNode nd = arr[base + i];
    if (qn_type(nd) != NBODY_NODE_LEAFBODY && qn_count(nd)) {
        nd.first += base;
        arr[base + i] = nd;
    }
}

}  // namespace

namespace {
void domain_marks(Octree* T, uint32_t** mark, const uint32_t** dom_off, uint32_t* n_dom) {
    if (T->dom->view >= 0) return;  // (a merged tree assembled for one domain: no marks)
    *mark = T->dom->mark;  // null until a counting traversal has made it
    *dom_off = T->dom->d_off;
    *n_dom = T->dom->G;
}
}  // namespace

static void domains_free(Octree* T) {
    Domains* Dm = T->dom;
    if (!Dm) return;
    for (Octree* t : Dm->tree) tree_free(t);
    tree_free(Dm->top);
    void* ptrs[] = {Dm->ckey, Dm->cell_count, Dm->cell_dom, Dm->dkey[0], Dm->dkey[1], Dm->dval[0], Dm->dval[1], Dm->d_off,
                    Dm->posm_k, Dm->n_entries, Dm->e_posm, Dm->e_multi, Dm->e_rec, Dm->e_bmin, Dm->e_bmax, Dm->e_range,
                    Dm->mark, Dm->imported, Dm->let_flags, Dm->let_newid, Dm->let_blk, Dm->send_nodes, Dm->send_range,
                    Dm->send_entries, Dm->recv_entries, Dm->boxes, Dm->let_counts, Dm->vel_k, Dm->acc_grouped};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    if (Dm->h_off) (void)hipHostFree(Dm->h_off);
    if (Dm->h_entries) (void)hipHostFree(Dm->h_entries);
    if (Dm->h_let_counts) (void)hipHostFree(Dm->h_let_counts);
    for (auto& pair : Dm->ev)
        for (hipEvent_t e : pair)
            if (e) (void)hipEventDestroy(e);
    for (auto& pair : Dm->ev_export)
        for (hipEvent_t e : pair)
            if (e) (void)hipEventDestroy(e);
    delete Dm;
    T->dom = nullptr;
}

int nb_tree_set_domains(nbody_ctx* ctx, uint32_t domains, uint32_t top_levels) {
    Octree* T = ctx->tree;
    NB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    domains_free(T);
    T->have_tree = false;
    T->horder = nullptr;
    if (domains <= 1) return NBODY_OK;
    if (domains > MAX_DOMAINS) return nb_fail(ctx, NBODY_ERR_INVALID, "nbody_bh_set_domains: at most 32 domains");
    if (top_levels < 1 || top_levels > DOMAIN_TOP_LEVELS_MAX || top_levels >= T->D)
        return nb_fail(ctx, NBODY_ERR_INVALID, "nbody_bh_set_domains: top_levels must be in [1, 6] and below max_depth");
    if (T->octree_rs || ctx->cfg.quadrupole)
        return nb_fail(ctx, NBODY_ERR_INVALID, "nbody_bh_set_domains: not with max_depth 0 or with quadrupoles");
    if (ctx->i_count != ctx->n) return nb_fail(ctx, NBODY_ERR_INVALID, "nbody_bh_set_domains: the context must hold all bodies (no shard range)");
    Domains* Dm = new Domains;
    T->dom = Dm;
    Dm->G = domains;
    Dm->L = top_levels;
    Dm->cells = 1u << (3 * top_levels);
    const uint64_t n = T->n_cap;
    auto fail = [&](int code, const char* msg) {
        domains_free(T);
        return nb_fail(ctx, code, msg);
    };
    hipError_t e = hipMalloc(&Dm->ckey, n * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc(&Dm->cell_count, Dm->cells * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc(&Dm->cell_dom, Dm->cells);
    for (int k = 0; k < 2 && e == hipSuccess; ++k) {
        e = hipMalloc(&Dm->dkey[k], n * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMalloc(&Dm->dval[k], n * sizeof(uint32_t));
    }
    if (e == hipSuccess) e = hipMalloc(&Dm->d_off, (MAX_DOMAINS + 1) * sizeof(uint32_t));
    if (e == hipSuccess) e = hipHostMalloc(&Dm->h_off, (MAX_DOMAINS + 1) * sizeof(uint32_t), hipHostMallocDefault);
    if (e == hipSuccess) e = hipMalloc(&Dm->posm_k, n * sizeof(float4));
    // the top tree's entries: every depth-L cell at most once, plus lone bodies above (fewer than cells)
    Dm->top_cap = (uint32_t)std::min<uint64_t>(2ull * Dm->cells + 1024, n + 1024);
    Dm->top_node_cap = 2 * Dm->top_cap + 1024;
    if (e == hipSuccess) e = hipMalloc(&Dm->n_entries, sizeof(uint32_t));
    if (e == hipSuccess) e = hipHostMalloc(&Dm->h_entries, sizeof(uint32_t), hipHostMallocDefault);
    if (e == hipSuccess) e = hipMalloc(&Dm->e_posm, Dm->top_cap * sizeof(float4));
    if (e == hipSuccess) e = hipMalloc(&Dm->e_multi, Dm->top_cap);
    if (e == hipSuccess) e = hipMalloc(&Dm->e_rec, Dm->top_cap * sizeof(QNode));
    if (e == hipSuccess) e = hipMalloc(&Dm->e_bmin, Dm->top_cap * sizeof(float4));
    if (e == hipSuccess) e = hipMalloc(&Dm->e_bmax, Dm->top_cap * sizeof(float4));
    if (e == hipSuccess) e = hipMalloc(&Dm->e_range, Dm->top_cap * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(&Dm->imported, (MAX_DOMAINS + 1) * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMalloc(&Dm->send_entries, Dm->top_cap * sizeof(LetEntry));
    if (e == hipSuccess) e = hipMalloc(&Dm->recv_entries, Dm->top_cap * sizeof(LetEntry));
    if (e == hipSuccess) e = hipMalloc(&Dm->boxes, (size_t)MAX_DOMAINS * LET_BOXES * sizeof(LetBox));
    if (e == hipSuccess) e = hipMalloc(&Dm->let_counts, 2 * MAX_DOMAINS * sizeof(uint32_t));
    if (e == hipSuccess) e = hipHostMalloc(&Dm->h_let_counts, 2 * MAX_DOMAINS * sizeof(uint32_t), hipHostMallocDefault);
    for (auto& pair : Dm->ev)
        for (hipEvent_t& ev : pair)
            if (e == hipSuccess) e = hipEventCreate(&ev);
    for (auto& pair : Dm->ev_export)
        for (hipEvent_t& ev : pair)
            if (e == hipSuccess) e = hipEventCreate(&ev);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail(NBODY_ERR_NOMEM, "nbody_bh_set_domains: out of device memory");
    }
    // the domains' trees are made when a domain is first built here (a rank of a distributed run builds one
    // only), with room for a third more than an even share (domains are runs of whole cells)
    Dm->share = std::min<uint64_t>(n, n / domains + n / (3 * domains) + 4096);
    Dm->top = new Octree;
    if (int r = tree_alloc(ctx, Dm->top, Dm->top_cap, top_levels)) {
        domains_free(T);
        return r;
    }
    return NBODY_OK;
}

// 1. which body goes where (ends with the domains' first bodies on the host)
static int domains_partition(nbody_ctx* ctx, Octree* T, const float4* posm_in) {
    Domains* Dm = T->dom;
    hipStream_t st = ctx->stream;
    const uint32_t n = (uint32_t)T->n_cap, G = Dm->G;
    const int L = (int)Dm->L;
    NB_HIP(ctx, hipEventRecord(Dm->ev[MAX_DOMAINS][0], st));
    NB_HIP(ctx, hipMemsetAsync(Dm->cell_count, 0, Dm->cells * sizeof(uint32_t), st));
    const uint32_t stride = std::max(1u, n / 131072u), sampled = (n + stride - 1) / stride;
    hipLaunchKernelGGL(k_cellkey, dim3((n + 255) / 256), dim3(256), 0, st, posm_in, n, ctx->cfg.world, L, stride, Dm->ckey,
                       Dm->cell_count);
    hipLaunchKernelGGL(k_cell_domains, dim3(1), dim3(1024), 0, st, Dm->cell_count, Dm->cells, sampled, G, Dm->cell_dom);
    hipLaunchKernelGGL(k_body_domain, dim3((n + 255) / 256), dim3(256), 0, st, Dm->ckey, Dm->cell_dom, n, Dm->dkey[0], Dm->dval[0]);
    const uint32_t rs_nb = (n + RS_TILE - 1) / RS_TILE;  // (T's scan workspace is made for n bodies)
    hipLaunchKernelGGL(k_rs_hist<uint32_t>, dim3(rs_nb), dim3(RS_THREADS), 0, st, Dm->dkey[0], n, 0, T->rs_counts, rs_nb);
    hipLaunchKernelGGL(k_rowscan, dim3(256), dim3(256), 0, st, T->rs_counts, rs_nb, T->row_total);
    hipLaunchKernelGGL(k_domain_offsets, dim3(1), dim3(64), 0, st, T->row_total, G, Dm->d_off);
    hipLaunchKernelGGL(k_rs_scatter<uint32_t>, dim3(rs_nb), dim3(RS_THREADS), 0, st, Dm->dkey[0], Dm->dval[0], Dm->dkey[1],
                       Dm->dval[1], n, 0, T->rs_counts, T->row_total, rs_nb);
    NB_HIP(ctx, hipGetLastError());
    NB_HIP(ctx, hipMemcpyAsync(Dm->h_off, Dm->d_off, (G + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    NB_HIP(ctx, hipMemsetAsync(Dm->n_entries, 0, sizeof(uint32_t), st));
    NB_HIP(ctx, hipEventRecord(Dm->ev[MAX_DOMAINS][1], st));
    NB_HIP(ctx, hipStreamSynchronize(st));
    return NBODY_OK;
}

// 2. domain k's own tree, from its own bodies
static int domain_build_one(nbody_ctx* ctx, Octree* T, uint32_t k, const float4* posm_in) {
    Domains* Dm = T->dom;
    hipStream_t st = ctx->stream;
    const uint32_t off = Dm->h_off[k], nk = Dm->h_off[k + 1] - off;
    Octree* Tk = Dm->tree[k];
    Dm->stats.bodies[k] = nk;
    Dm->stats.nodes[k] = 0;
    NB_HIP(ctx, hipEventRecord(Dm->ev[k][0], st));
    if (nk > 0) {
        if (!Tk || nk > Tk->n_cap) {  // first use, or a domain grown past its room: new arrays, with room to spare
            tree_free(Tk);
            Dm->tree[k] = Tk = new Octree;
            if (int r = tree_alloc(ctx, Tk, std::max<uint64_t>(Dm->share, (uint64_t)nk + nk / 4 + 4096), T->D)) return r;
        }
        hipLaunchKernelGGL(k_gather_bodies, dim3((nk + 255) / 256), dim3(256), 0, st, posm_in, Dm->dval[1] + off, nk,
                           Dm->posm_k + off);
        if (int r = build_into(ctx, Tk, Dm->posm_k + off, nk, nullptr)) return r;
        Dm->stats.nodes[k] = Tk->n_nodes;
    }
    NB_HIP(ctx, hipEventRecord(Dm->ev[k][1], st));
    return NBODY_OK;
}

// 3. the top tree over the M entries in Dm->e_*, its deepest nodes being the domains' own; into the merged array
static int domains_top(nbody_ctx* ctx, Octree* T, uint32_t M) {
    Domains* Dm = T->dom;
    hipStream_t st = ctx->stream;
    if (M == 0 || M > Dm->top_cap) return nb_fail(ctx, NBODY_ERR_CAPACITY, "domain build: more top-tree entries than there is room for");
    const TopPatch patch{Dm->e_multi, Dm->e_rec, Dm->e_bmin, Dm->e_bmax, Dm->e_range};
    if (int r = build_into(ctx, Dm->top, Dm->e_posm, M, &patch)) return r;
    if (Dm->top->n_nodes > Dm->top_node_cap) return nb_fail(ctx, NBODY_ERR_CAPACITY, "domain build: the top tree outgrew its room");
    NB_HIP(ctx, hipMemcpyAsync(T->nodes, Dm->top->nodes, Dm->top->n_nodes * sizeof(QNode), hipMemcpyDeviceToDevice, st));
    NB_HIP(ctx, hipMemcpyAsync(T->range, Dm->top->range, Dm->top->n_nodes * sizeof(float), hipMemcpyDeviceToDevice, st));
    Dm->stats.top_entries = M;
    Dm->stats.top_nodes = Dm->top->n_nodes;
    return NBODY_OK;
}

static int domains_build(nbody_ctx* ctx, Octree* T, const float4* posm_in) {
    Domains* Dm = T->dom;
    hipStream_t st = ctx->stream;
    const uint32_t n = (uint32_t)T->n_cap, G = Dm->G;
    const int L = (int)Dm->L;
    T->have_tree = false;
    if (int r = domains_partition(ctx, T, posm_in)) return r;
    // 2. every domain's own tree, and what it hands to the top tree
    Dm->node_base[0] = Dm->top_node_cap;
    uint64_t merged = 0;
    for (uint32_t k = 0; k < G; ++k) {
        if (int r = domain_build_one(ctx, T, k, posm_in)) return r;
        Dm->node_base[k + 1] = Dm->node_base[k] + Dm->stats.nodes[k];
    }
    const uint64_t total = Dm->node_base[G];
    if (total > 0xfffffff0ull) return nb_fail(ctx, NBODY_ERR_CAPACITY, "octree needs more than 2^32 nodes");
    if (int r = ensure_nodes(ctx, T, total)) return r;
    NB_HIP(ctx, hipEventRecord(Dm->ev[MAX_DOMAINS + 1][0], st));
    for (uint32_t k = 0; k < G; ++k) {
        const uint32_t off = Dm->h_off[k], nk = Dm->h_off[k + 1] - off;
        if (!nk) continue;
        Octree* Tk = Dm->tree[k];
        const uint32_t base = (uint32_t)Dm->node_base[k], cnt = (uint32_t)Tk->n_nodes;
        hipLaunchKernelGGL(k_rebase, dim3((cnt + 255) / 256), dim3(256), 0, st, Tk->nodes, Tk->range, cnt, base, T->nodes, T->range);
        hipLaunchKernelGGL(k_global_order, dim3((nk + 255) / 256), dim3(256), 0, st, Tk->keys[Tk->sorted], Tk->idx[Tk->sorted],
                           Dm->dval[1] + off, nk, T->keys[0] + off, T->idx[0] + off);
        const uint32_t limit = Tk->h_lvl_start[L + 1];
        hipLaunchKernelGGL(k_top_entries, dim3((limit + 255) / 256), dim3(256), 0, st, Tk->nodes, Tk->bmin, Tk->bmax, Tk->range,
                           Tk->node_head, Tk->idx[Tk->sorted], Dm->posm_k + off, Tk->lvl_start, L, limit, base, Dm->n_entries,
                           Dm->top_cap, Dm->e_posm, Dm->e_multi, Dm->e_rec, Dm->e_bmin, Dm->e_bmax, Dm->e_range, T->err_flag);
        merged += Tk->n_nodes - limit;  // its nodes below depth L (those at depth L and the lone bodies above reappear in the top tree)
    }
    NB_HIP(ctx, hipGetLastError());
    NB_HIP(ctx, hipMemcpyAsync(Dm->h_entries, Dm->n_entries, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    NB_HIP(ctx, hipStreamSynchronize(st));
    if (int r = domains_top(ctx, T, *Dm->h_entries)) return r;
    NB_HIP(ctx, hipEventRecord(Dm->ev[MAX_DOMAINS + 1][1], st));
    T->sorted = 0;
    T->n = n;
    T->n_nodes = total;
    T->have_tree = true;
    T->built = true;
    T->export_is_upload = false;
    ++T->horder_age;
    Dm->stats.domains = G;
    Dm->stats.top_levels = (uint32_t)L;
    Dm->stats.merged_nodes = merged + Dm->top->n_nodes;
    Dm->stats.view_domain = -1;
    Dm->stats.distributed = 0;
    for (uint32_t k = 0; k < MAX_DOMAINS; ++k) Dm->stats.exported_nodes[k] = 0;
    return NBODY_OK;
}

// ---- essential trees: host side ----------------------------------------------------------------
static int let_reserve(nbody_ctx* ctx, Domains* Dm, uint64_t nodes) {
    if (nodes <= Dm->let_cap) return NBODY_OK;
    NB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    void** arrays[] = {(void**)&Dm->let_flags, (void**)&Dm->let_newid, (void**)&Dm->let_blk, (void**)&Dm->send_nodes, (void**)&Dm->send_range};
    for (void** a : arrays) {
        if (*a) (void)hipFree(*a);
        *a = nullptr;
    }
    Dm->let_cap = Dm->send_cap = 0;
    const uint64_t cap = nodes + nodes / 4 + 1024;
    NB_HIP(ctx, hipMalloc(&Dm->let_flags, cap));
    NB_HIP(ctx, hipMalloc(&Dm->let_newid, cap * sizeof(uint32_t)));
    NB_HIP(ctx, hipMalloc(&Dm->let_blk, ((cap + LET_TILE - 1) / LET_TILE + 1) * sizeof(uint32_t)));
    NB_HIP(ctx, hipMalloc(&Dm->send_nodes, cap * sizeof(QNode)));
    NB_HIP(ctx, hipMalloc(&Dm->send_range, cap * sizeof(float)));
    Dm->let_cap = Dm->send_cap = cap;
    return NBODY_OK;
}

// Domain k's entries of the top tree into `out`, their number into *counter (device).  exported: child
// indices as of the last let_export of this domain; else the domain's own numbering.
static int let_entries(nbody_ctx* ctx, Octree* T, uint32_t k, bool exported, LetEntry* out, uint32_t* counter) {
    Domains* Dm = T->dom;
    Octree* Tk = Dm->tree[k];
    hipStream_t st = ctx->stream;
    const uint32_t off = Dm->h_off[k], limit = Tk->h_lvl_start[Dm->L + 1];
    NB_HIP(ctx, hipMemsetAsync(counter, 0, sizeof(uint32_t), st));
    hipLaunchKernelGGL(k_let_entries, dim3((limit + 255) / 256), dim3(256), 0, st, Tk->nodes, Tk->bmin, Tk->bmax, Tk->range,
                       Tk->node_head, Tk->idx[Tk->sorted], Dm->posm_k + off, Tk->lvl_start, (int)Dm->L, limit,
                       exported ? Dm->let_flags : (const uint8_t*)nullptr, exported ? Dm->let_newid : (const uint32_t*)nullptr,
                       counter, Dm->top_cap, out, T->err_flag);
    NB_HIP(ctx, hipGetLastError());
    return NBODY_OK;
}

// What domain k (its tree built, Dm->boxes filled) hands to the others: nodes into Dm->send_nodes /
// send_range, entries into Dm->send_entries, their numbers into Dm->let_counts[2k], [2k + 1] (device).
static int let_export(nbody_ctx* ctx, Octree* T, uint32_t k) {
    Domains* Dm = T->dom;
    Octree* Tk = Dm->tree[k];
    hipStream_t st = ctx->stream;
    uint32_t* counts = Dm->let_counts + 2 * k;
    NB_HIP(ctx, hipMemsetAsync(counts, 0, 2 * sizeof(uint32_t), st));
    Dm->export_timed[k] = false;
    if (Dm->h_off[k + 1] == Dm->h_off[k]) return NBODY_OK;
    NB_HIP(ctx, hipEventRecord(Dm->ev_export[k][0], st));
    const uint32_t n_nodes = (uint32_t)Tk->n_nodes, mark0 = Tk->h_lvl_start[Dm->L], lo_id = Tk->h_lvl_start[Dm->L + 1];
    if (int r = let_reserve(ctx, Dm, n_nodes)) return r;
    NB_HIP(ctx, hipMemsetAsync(Dm->let_flags, 0, n_nodes, st));
    const float soft = ctx->cfg.law == NBODY_LAW_PLUMMER ? ctx->cfg.eps2 : 0.0f;
    if (n_nodes > mark0)
        hipLaunchKernelGGL(k_let_mark, dim3((n_nodes - mark0 + 255) / 256), dim3(256), 0, st, Tk->nodes, mark0, n_nodes - mark0,
                           Dm->boxes, Dm->G, k, soft, Dm->let_flags);
    if (n_nodes > lo_id) {
        const uint32_t c = n_nodes - lo_id, nb = (c + LET_TILE - 1) / LET_TILE;
        hipLaunchKernelGGL(k_let_count, dim3(nb), dim3(256), 0, st, Dm->let_flags, lo_id, n_nodes, Dm->let_blk);
        hipLaunchKernelGGL(k_rowscan, dim3(1), dim3(256), 0, st, Dm->let_blk, nb, counts);
        hipLaunchKernelGGL(k_let_number, dim3(nb), dim3(256), 0, st, Dm->let_flags, lo_id, n_nodes, Dm->let_blk, Dm->let_newid);
        hipLaunchKernelGGL(k_let_pack, dim3((c + 255) / 256), dim3(256), 0, st, Tk->nodes, Tk->range, Dm->let_flags, Dm->let_newid,
                           lo_id, n_nodes, Dm->send_nodes, Dm->send_range, (uint32_t)Dm->send_cap);
    }
    NB_HIP(ctx, hipGetLastError());
    if (int r = let_entries(ctx, T, k, true, Dm->send_entries, counts + 1)) return r;
    NB_HIP(ctx, hipEventRecord(Dm->ev_export[k][1], st));
    Dm->export_timed[k] = true;
    return NBODY_OK;
}

static int let_box(nbody_ctx* ctx, Octree* T, uint32_t k) {
    Domains* Dm = T->dom;
    Octree* Tk = Dm->tree[k];
    const uint32_t bodies = Tk ? Dm->h_off[k + 1] - Dm->h_off[k] : 0u;  // (no tree: the domain has never held a body here)
    hipLaunchKernelGGL(k_let_box, dim3(1), dim3(128), 0, ctx->stream, Tk ? Tk->nodes : (const QNode*)nullptr,
                       Tk ? Tk->bmin : (const float4*)nullptr, Tk ? Tk->bmax : (const float4*)nullptr,
                       Tk ? Tk->lvl_start : (const uint32_t*)nullptr, bodies, Dm->boxes + (size_t)k * LET_BOXES);
    NB_HIP(ctx, hipGetLastError());
    return NBODY_OK;
}

// `count` nodes that lie at T->nodes + base (as sent) become part of the merged array; the sender's m
// entries in `entries` join the top tree's at e0.
static int let_settle(nbody_ctx* ctx, Octree* T, uint32_t base, uint32_t count, const LetEntry* entries, uint32_t m, uint32_t e0) {
    Domains* Dm = T->dom;
    hipStream_t st = ctx->stream;
    if (count) hipLaunchKernelGGL(k_let_settle, dim3((count + 255) / 256), dim3(256), 0, st, T->nodes, base, count);
    if (m)
        hipLaunchKernelGGL(k_let_entries_unpack, dim3((m + 255) / 256), dim3(256), 0, st, entries, m, base, e0, Dm->e_posm,
                           Dm->e_multi, Dm->e_rec, Dm->e_bmin, Dm->e_bmax, Dm->e_range);
    NB_HIP(ctx, hipGetLastError());
    return NBODY_OK;
}

// The whole tree of domain k (this GPU's own) into the merged array at base, its entries at e0; *m_out: how many.
static int let_own(nbody_ctx* ctx, Octree* T, uint32_t k, uint32_t base, uint32_t e0, uint32_t* m_out) {
    Domains* Dm = T->dom;
    Octree* Tk = Dm->tree[k];
    hipStream_t st = ctx->stream;
    const uint32_t cnt = (uint32_t)Tk->n_nodes;
    hipLaunchKernelGGL(k_rebase, dim3((cnt + 255) / 256), dim3(256), 0, st, Tk->nodes, Tk->range, cnt, base, T->nodes, T->range);
    uint32_t* counter = Dm->let_counts + 2 * k + 1;
    if (int r = let_entries(ctx, T, k, false, Dm->send_entries, counter)) return r;
    NB_HIP(ctx, hipMemcpyAsync(Dm->h_let_counts + 2 * k + 1, counter, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    NB_HIP(ctx, hipStreamSynchronize(st));
    const uint32_t m = Dm->h_let_counts[2 * k + 1];
    if (e0 + (uint64_t)m > Dm->top_cap) return nb_fail(ctx, NBODY_ERR_CAPACITY, "domain build: more top-tree entries than there is room for");
    *m_out = m;
    return let_settle(ctx, T, base, 0, Dm->send_entries, m, e0);
}

// The merged tree as the GPU of domain `view` would hold it, on one GPU: all domains' trees are built here,
// but the others' reach the merged array only through what they would send.  Only the targets of domain
// `view` are served by it.
static int domains_build_view(nbody_ctx* ctx, Octree* T, const float4* posm_in) {
    Domains* Dm = T->dom;
    hipStream_t st = ctx->stream;
    const uint32_t n = (uint32_t)T->n_cap, G = Dm->G, view = (uint32_t)Dm->view;
    T->have_tree = false;
    if (int r = domains_partition(ctx, T, posm_in)) return r;
    uint64_t bound = Dm->top_node_cap;
    for (uint32_t k = 0; k < G; ++k) {
        if (int r = domain_build_one(ctx, T, k, posm_in)) return r;
        if (int r = let_box(ctx, T, k)) return r;
        bound += Dm->stats.nodes[k];
    }
    if (bound > 0xfffffff0ull) return nb_fail(ctx, NBODY_ERR_CAPACITY, "octree needs more than 2^32 nodes");
    if (int r = ensure_nodes(ctx, T, bound)) return r;
    NB_HIP(ctx, hipEventRecord(Dm->ev[MAX_DOMAINS + 1][0], st));
    uint32_t base = Dm->top_node_cap, e0 = 0;
    for (uint32_t k = 0; k < G; ++k) {
        const uint32_t off = Dm->h_off[k], nk = Dm->h_off[k + 1] - off;
        Dm->node_base[k] = base;
        Dm->stats.exported_nodes[k] = 0;
        if (!nk) continue;
        Octree* Tk = Dm->tree[k];
        hipLaunchKernelGGL(k_global_order, dim3((nk + 255) / 256), dim3(256), 0, st, Tk->keys[Tk->sorted], Tk->idx[Tk->sorted],
                           Dm->dval[1] + off, nk, T->keys[0] + off, T->idx[0] + off);
        if (k == view) {
            uint32_t m = 0;
            if (int r = let_own(ctx, T, k, base, e0, &m)) return r;
            base += (uint32_t)Tk->n_nodes;
            e0 += m;
            continue;
        }
        if (int r = let_export(ctx, T, k)) return r;
        NB_HIP(ctx, hipMemcpyAsync(Dm->h_let_counts + 2 * k, Dm->let_counts + 2 * k, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        NB_HIP(ctx, hipStreamSynchronize(st));
        const uint32_t c = Dm->h_let_counts[2 * k], m = Dm->h_let_counts[2 * k + 1];
        if (c > Tk->n_nodes || e0 + (uint64_t)m > Dm->top_cap) return nb_fail(ctx, NBODY_ERR_CAPACITY, "domain build: export beyond its room");
        if (c) {
            NB_HIP(ctx, hipMemcpyAsync(T->nodes + base, Dm->send_nodes, (size_t)c * sizeof(QNode), hipMemcpyDeviceToDevice, st));
            NB_HIP(ctx, hipMemcpyAsync(T->range + base, Dm->send_range, (size_t)c * sizeof(float), hipMemcpyDeviceToDevice, st));
        }
        if (int r = let_settle(ctx, T, base, c, Dm->send_entries, m, e0)) return r;
        NB_HIP(ctx, hipStreamSynchronize(st));  // (the send buffers are reused by the next domain)
        Dm->stats.exported_nodes[k] = c;
        base += c;
        e0 += m;
    }
    Dm->node_base[G] = base;
    if (int r = domains_top(ctx, T, e0)) return r;
    NB_HIP(ctx, hipEventRecord(Dm->ev[MAX_DOMAINS + 1][1], st));
    T->sorted = 0;
    T->n = n;
    T->n_nodes = base;
    T->have_tree = true;
    T->built = true;
    T->export_is_upload = false;
    ++T->horder_age;
    Dm->stats.domains = G;
    Dm->stats.top_levels = Dm->L;
    Dm->stats.merged_nodes = base - Dm->top_node_cap + Dm->top->n_nodes;
    Dm->stats.view_domain = (int32_t)view;
    Dm->stats.distributed = 0;
    return NBODY_OK;
}

static int walk_launch(nbody_ctx* ctx, Octree* T, const uint32_t* order, uint32_t m, const float4* posm_in,
                       const float4* vel_in, float4* acc_out, uint32_t i_begin, int out_by_rank);

bool nb_tree_distributed(const nbody_ctx* ctx) { return ctx->comm && ctx->tree && ctx->tree->dom; }

// One Barnes-Hut force evaluation by as many ranks as there are domains (one GPU each), every rank holding
// all bodies' state.  All ranks deal the bodies to the domains alike (the partition is a function of the
// positions); rank r builds the tree of domain r alone, tells the others where its bodies are (boxes), is
// told where theirs are, and exports what their targets can reach of its tree; after one exchange of the
// counts, every rank's nodes, ranges and top-tree entries travel as one group of broadcasts, straight to
// their places in the receivers' merged arrays.  Each rank then builds the (small) top tree, traverses for
// the bodies of its own domain, and the accelerations are exchanged domain by domain.
int nb_tree_forces_distributed(nbody_ctx* ctx, const float4* posm_in, const float4* vel_in, float4* acc_body_out) {
    Octree* T = ctx->tree;
    Domains* Dm = T->dom;
    hipStream_t st = ctx->stream;
    const uint32_t n = (uint32_t)T->n_cap, G = Dm->G, self = (uint32_t)nb_comm_rank(ctx);
    if ((uint32_t)nb_comm_world(ctx) != G)
        return nb_fail(ctx, NBODY_ERR_INVALID, "nbody_step_bh: with a communicator there must be as many domains as ranks (nbody_bh_set_domains)");
    if (Dm->view >= 0) return nb_fail(ctx, NBODY_ERR_INVALID, "nbody_step_bh: nbody_bh_set_domain_view is a one-GPU rehearsal; switch it off for a distributed step");
    T->have_tree = false;
    if (!Dm->acc_grouped) NB_HIP(ctx, hipMalloc(&Dm->acc_grouped, (size_t)n * sizeof(float4)));
    const bool need_vel = ctx->cfg.law == NBODY_LAW_WGSL_BIAS;
    if (need_vel && !Dm->vel_k) NB_HIP(ctx, hipMalloc(&Dm->vel_k, (size_t)n * sizeof(float4)));
    struct BuildTimer {  // the build's time ends where the build ends, also when it fails
        nbody_ctx* ctx;
        bool open = true;
        void end() {
            if (open) nb_time_end(ctx, 2);
            open = false;
        }
        ~BuildTimer() { end(); }
    } timer{ctx};
    nb_time_begin(ctx, 2);
    if (int r = domains_partition(ctx, T, posm_in)) return r;
    for (uint32_t k = 0; k < G; ++k) {
        Dm->stats.bodies[k] = Dm->h_off[k + 1] - Dm->h_off[k];
        Dm->stats.nodes[k] = 0;
    }
    const uint32_t off = Dm->h_off[self], nk = Dm->h_off[self + 1] - off;
    if (int r = domain_build_one(ctx, T, self, posm_in)) return r;
    Octree* Ts = Dm->tree[self];
    NB_HIP(ctx, hipEventRecord(Dm->ev[MAX_DOMAINS + 1][0], st));
    // where everybody's bodies are, what everybody exports
    if (int r = let_box(ctx, T, self)) return r;
    if (int r = nb_comm_allgather_inplace(ctx, Dm->boxes, (size_t)LET_BOXES * sizeof(LetBox))) return r;
    if (int r = let_export(ctx, T, self)) return r;
    if (int r = nb_comm_allgather_inplace(ctx, Dm->let_counts, 2 * sizeof(uint32_t))) return r;
    NB_HIP(ctx, hipMemcpyAsync(Dm->h_let_counts, Dm->let_counts, 2 * G * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    NB_HIP(ctx, hipStreamSynchronize(st));
    uint64_t base[MAX_DOMAINS + 1], e0[MAX_DOMAINS + 1];
    base[0] = Dm->top_node_cap;
    e0[0] = 0;
    for (uint32_t k = 0; k < G; ++k) {
        const uint32_t c = Dm->h_let_counts[2 * k], m = Dm->h_let_counts[2 * k + 1];
        base[k + 1] = base[k] + (k == self ? (nk ? Ts->n_nodes : 0) : c);
        e0[k + 1] = e0[k] + m;
        Dm->node_base[k] = base[k];
        Dm->stats.exported_nodes[k] = c;
    }
    Dm->node_base[G] = base[G];
    if (base[G] > 0xfffffff0ull) return nb_fail(ctx, NBODY_ERR_CAPACITY, "octree needs more than 2^32 nodes");
    if (e0[G] == 0 || e0[G] > Dm->top_cap) return nb_fail(ctx, NBODY_ERR_CAPACITY, "domain build: more top-tree entries than there is room for");
    if (Dm->h_let_counts[2 * self] > Dm->send_cap && nk) return nb_fail(ctx, NBODY_ERR_CAPACITY, "domain build: export beyond its room");
    if (int r = ensure_nodes(ctx, T, base[G])) return r;
    std::vector<NbBroadcast> items;
    for (uint32_t k = 0; k < G; ++k) {
        const size_t c = Dm->h_let_counts[2 * k], m = Dm->h_let_counts[2 * k + 1];
        const bool mine = k == self;
        items.push_back({Dm->send_nodes, mine ? (void*)Dm->send_nodes : (void*)(T->nodes + base[k]), c * sizeof(QNode), (int)k});
        items.push_back({Dm->send_range, mine ? (void*)Dm->send_range : (void*)(T->range + base[k]), c * sizeof(float), (int)k});
        items.push_back({Dm->send_entries, mine ? (void*)Dm->send_entries : (void*)(Dm->recv_entries + e0[k]), m * sizeof(LetEntry), (int)k});
    }
    if (int r = nb_comm_broadcasts(ctx, items.data(), (int)items.size())) return r;
    for (uint32_t k = 0; k < G; ++k) {
        if (k == self) continue;
        if (int r = let_settle(ctx, T, (uint32_t)base[k], Dm->h_let_counts[2 * k], Dm->recv_entries + e0[k], Dm->h_let_counts[2 * k + 1],
                               (uint32_t)e0[k]))
            return r;
    }
    if (nk) {  // (after the broadcasts have read the send buffers: the entries are made again, in the tree's own numbering)
        uint32_t m = 0;
        const uint32_t sent = Dm->h_let_counts[2 * self + 1];
        if (int r = let_own(ctx, T, self, (uint32_t)base[self], (uint32_t)e0[self], &m)) return r;
        if (m != sent) return nb_fail(ctx, NBODY_ERR_STATE, "domain build: a domain's entries changed between export and use");
    }
    if (int r = domains_top(ctx, T, (uint32_t)e0[G])) return r;
    NB_HIP(ctx, hipEventRecord(Dm->ev[MAX_DOMAINS + 1][1], st));
    if (int r = check_err_flag(ctx, T, "distributed octree build")) return r;  // (the host has just waited for the top tree's level sizes anyway)
    timer.end();
    Dm->stats.domains = G;
    Dm->stats.top_levels = Dm->L;
    Dm->stats.merged_nodes = base[G] - Dm->top_node_cap + Dm->top->n_nodes;
    Dm->stats.view_domain = -1;
    Dm->stats.distributed = 1;
    ctx->counters.octree_nodes = base[G];
    // this rank's targets: the bodies of its domain, in the sorted order of its own tree
    if (nk) {
        if (need_vel) hipLaunchKernelGGL(k_gather_bodies, dim3((nk + 255) / 256), dim3(256), 0, st, vel_in, Dm->dval[1] + off, nk, Dm->vel_k + off);
        T->n = n;
        T->n_nodes = base[G];
        T->have_tree = true;
        T->built = false;  // (a merged array for this rank's targets only: no keys, no order of all bodies)
        const int r = walk_launch(ctx, T, Ts->idx[Ts->sorted], nk, Dm->posm_k + off, need_vel ? Dm->vel_k + off : nullptr,
                                  Dm->acc_grouped + off, 0u, 0);
        T->have_tree = false;
        if (r) return r;
    }
    items.clear();
    for (uint32_t k = 0; k < G; ++k) {
        float4* part = Dm->acc_grouped + Dm->h_off[k];
        items.push_back({part, part, (size_t)(Dm->h_off[k + 1] - Dm->h_off[k]) * sizeof(float4), (int)k});
    }
    if (int r = nb_comm_broadcasts(ctx, items.data(), (int)items.size())) return r;
    hipLaunchKernelGGL(k_unrank, dim3((n + 255) / 256), dim3(256), 0, st, Dm->dval[1], Dm->acc_grouped, n, acc_body_out);
    NB_HIP(ctx, hipGetLastError());
    return NBODY_OK;
}

int nb_tree_set_domain_view(nbody_ctx* ctx, int domain) {
    Octree* T = ctx->tree;
    Domains* Dm = T->dom;
    if (!Dm) return nb_fail(ctx, NBODY_ERR_STATE, "nbody_bh_set_domain_view: no domains set (nbody_bh_set_domains)");
    if (domain >= (int)Dm->G) return nb_fail(ctx, NBODY_ERR_INVALID, "nbody_bh_set_domain_view: no such domain");
    NB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    Dm->view = domain < 0 ? -1 : domain;
    T->have_tree = false;
    return NBODY_OK;
}

int nb_tree_domain_stats(nbody_ctx* ctx, nbody_domain_stats_t* out) {
    Octree* T = ctx->tree;
    Domains* Dm = T->dom;
    if (!Dm) return nb_fail(ctx, NBODY_ERR_STATE, "nbody_bh_domain_stats: no domains set (nbody_bh_set_domains)");
    if (!T->have_tree && !Dm->stats.distributed) return nb_fail(ctx, NBODY_ERR_STATE, "nbody_bh_domain_stats: no tree built yet");
    NB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    auto ms = [&](int slot) {
        float t = 0.0f;
        if (hipEventElapsedTime(&t, Dm->ev[slot][0], Dm->ev[slot][1]) == hipSuccess) return (double)t;
        (void)hipGetLastError();  // (events of a part that did not run in the last build, e.g. another rank's domain)
        return 0.0;
    };
    for (uint32_t k = 0; k < Dm->G; ++k) {
        Dm->stats.build_ms[k] = ms((int)k);
        float t = 0.0f;
        const bool timed = (Dm->stats.view_domain >= 0 || Dm->stats.distributed) && Dm->export_timed[k] &&
                           hipEventElapsedTime(&t, Dm->ev_export[k][0], Dm->ev_export[k][1]) == hipSuccess;
        if (!timed) (void)hipGetLastError();
        Dm->stats.export_ms[k] = timed ? (double)t : 0.0;
    }
    Dm->stats.partition_ms = ms(MAX_DOMAINS);
    Dm->stats.merge_ms = ms(MAX_DOMAINS + 1);
    *out = Dm->stats;
    return NBODY_OK;
}

int nb_tree_build(nbody_ctx* ctx, const float4* posm_in) {
    Octree* T = ctx->tree;
    nb_time_begin(ctx, 2);
    const int r = !T->dom ? build_into(ctx, T, posm_in, (uint32_t)T->n_cap, nullptr)
                  : T->dom->view >= 0 ? domains_build_view(ctx, T, posm_in) : domains_build(ctx, T, posm_in);
    nb_time_end(ctx, 2);
    if (r) return r;
    ctx->counters.octree_nodes = T->n_nodes;
    return NBODY_OK;
}

static int walk_launch(nbody_ctx* ctx, Octree* T, const uint32_t* order, uint32_t m, const float4* posm_in,
                       const float4* vel_in, float4* acc_out, uint32_t i_begin, int out_by_rank) {
    hipStream_t st = ctx->stream;
    if (ctx->counting) NB_HIP(ctx, hipMemsetAsync(T->walk_counters, 0, NWALK_COUNTERS * sizeof(uint64_t), st));
    Domains* Dm = (ctx->counting && T->built && T->dom && T->dom->view < 0) ? T->dom : nullptr;
    if (Dm) {  // the marks of which domains' targets visit which nodes
        if (Dm->mark_cap < T->cap_nodes) {
            if (Dm->mark) NB_HIP(ctx, hipFree(Dm->mark));
            Dm->mark = nullptr;
            Dm->mark_cap = 0;
            NB_HIP(ctx, hipMalloc(&Dm->mark, T->cap_nodes * sizeof(uint32_t)));
            Dm->mark_cap = T->cap_nodes;
        }
        NB_HIP(ctx, hipMemsetAsync(Dm->mark, 0, T->n_nodes * sizeof(uint32_t), st));
        NB_HIP(ctx, hipMemsetAsync(Dm->imported, 0, (MAX_DOMAINS + 1) * sizeof(unsigned long long), st));
    }
    nb_time_begin(ctx, 1);
    if (ctx->cfg.law == NBODY_LAW_WGSL_BIAS)
        launch_walk<NBODY_LAW_WGSL_BIAS>(ctx, T, order, m, posm_in, vel_in, acc_out, i_begin, out_by_rank);
    else
        launch_walk<NBODY_LAW_PLUMMER>(ctx, T, order, m, posm_in, vel_in, acc_out, i_begin, out_by_rank);
    nb_time_end(ctx, 1);
    NB_HIP(ctx, hipGetLastError());
    if (ctx->counting) {
        uint64_t h[NWALK_COUNTERS];
        NB_HIP(ctx, hipMemcpyAsync(h, T->walk_counters, sizeof h, hipMemcpyDeviceToHost, st));
        NB_HIP(ctx, hipStreamSynchronize(st));
        if (const char* dump = std::getenv("NBODY_WALK_HIST")) {  // measurement aid: the 16 bins, to stderr
            if (*dump == '1') {
                std::fprintf(stderr, "walk lanes-per-visit histogram (bins of 4 lanes):");
                for (int b = 0; b < 16; ++b) std::fprintf(stderr, " %llu", (unsigned long long)h[8 + b]);
                std::fprintf(stderr, "\n");
            }
        }
        ctx->counters.bh_node_visits = h[0];
        ctx->counters.bh_leaf_interactions = h[1];
        ctx->counters.bh_aggregate_interactions = h[2];
        ctx->counters.bh_wave_node_visits = h[3];
        ctx->counters.bh_max_stack = h[4];
        if (Dm) {
            const uint32_t G = Dm->G;
            for (uint32_t k = 0; k <= G; ++k) {  // slot G: the top tree
                const uint32_t first = k < G ? (uint32_t)Dm->node_base[k] : 0u;
                const uint32_t cnt = k < G ? (uint32_t)(Dm->node_base[k + 1] - Dm->node_base[k]) : (uint32_t)Dm->top->n_nodes;
                if (cnt) hipLaunchKernelGGL(k_imported, dim3((cnt + 255) / 256), dim3(256), 0, st, Dm->mark, first, cnt, k, G, Dm->imported);
            }
            unsigned long long imp[MAX_DOMAINS + 1];
            NB_HIP(ctx, hipMemcpyAsync(imp, Dm->imported, sizeof imp, hipMemcpyDeviceToHost, st));
            NB_HIP(ctx, hipStreamSynchronize(st));
            for (uint32_t k = 0; k < G; ++k) Dm->stats.imported_nodes[k] = imp[k];
            Dm->stats.top_nodes_visited = imp[G];
        }
        return check_err_flag(ctx, T, "traversal");
    }
    return NBODY_OK;
}

// Body indices along the Hilbert curve (stable among the bodies of one depth-HILBERT_LEVELS cell, which
// keep their Morton order), from the keys and order of the present build.
static int make_hilbert_order(nbody_ctx* ctx, Octree* T) {
    const uint32_t n = (uint32_t)T->n;
    hipStream_t st = ctx->stream;
    if (!T->hkeys[0]) {
        for (int k = 0; k < 2; ++k) {
            NB_HIP(ctx, hipMalloc(&T->hkeys[k], (size_t)n * sizeof(uint32_t)));
            NB_HIP(ctx, hipMalloc(&T->hidx[k], (size_t)n * sizeof(uint32_t)));
        }
    }
    const int D = (int)T->D, L = std::min(D, HILBERT_LEVELS);
    hipLaunchKernelGGL(k_hilbert, dim3((n + 255) / 256), dim3(256), 0, st, T->keys[T->sorted], T->idx[T->sorted], n, D, L,
                       T->hkeys[0], T->hidx[0]);
    int cur = 0;
    for (int shift = 0; shift < 3 * L; shift += 8) {
        hipLaunchKernelGGL(k_rs_hist<uint32_t>, dim3(T->rs_nb), dim3(RS_THREADS), 0, st, T->hkeys[cur], n, shift, T->rs_counts, T->rs_nb);
        hipLaunchKernelGGL(k_rowscan, dim3(256), dim3(256), 0, st, T->rs_counts, T->rs_nb, T->row_total);
        hipLaunchKernelGGL(k_rs_scatter<uint32_t>, dim3(T->rs_nb), dim3(RS_THREADS), 0, st, T->hkeys[cur], T->hidx[cur],
                           T->hkeys[cur ^ 1], T->hidx[cur ^ 1], n, shift, T->rs_counts, T->row_total, T->rs_nb);
        cur ^= 1;
    }
    NB_HIP(ctx, hipGetLastError());
    T->horder = T->hidx[cur];
    T->horder_age = 0;
    T->perm_nblk = 0;  // the wavefronts' costs belonged to the previous grouping
    return NBODY_OK;
}

void nb_tree_forget_order(nbody_ctx* ctx) {
    if (ctx->tree) ctx->tree->horder = nullptr;
}

int nb_tree_walk(nbody_ctx* ctx, const float4* posm_in, const float4* vel_in, float4* acc_out) {
    Octree* T = ctx->tree;
    if (!T->have_tree) return nb_fail(ctx, NBODY_ERR_STATE, "no octree: call nbody_build_octree or nbody_upload_octree first");
    hipStream_t st = ctx->stream;
    const uint32_t m = (uint32_t)ctx->i_count;
    const uint32_t* order = nullptr;
    if (T->built) {
        if (ctx->i_count == ctx->n && T->user_order) {
            return walk_launch(ctx, T, T->user_order, (uint32_t)T->user_m, posm_in, vel_in, acc_out, 0u, 0);
        } else if (ctx->i_count == ctx->n) {
            // all bodies: grouped along the Hilbert curve, the order kept over several builds
            static const bool hilbert = env_int_or("NBODY_WALK_HILBERT", 1) != 0;
            order = T->idx[T->sorted];
            if (hilbert && T->n >= HILBERT_MIN_N && !(T->dom && ctx->counting)) {
                if (!T->horder || T->horder_age >= HILBERT_KEEP) {
                    int r = make_hilbert_order(ctx, T);
                    if (r) return r;
                }
                order = T->horder;
            }
        } else {
            const uint32_t nb = ((uint32_t)T->n + CO_TILE - 1) / CO_TILE;
            const uint32_t b0 = (uint32_t)ctx->i_begin, b1 = (uint32_t)(ctx->i_begin + ctx->i_count);
            hipLaunchKernelGGL(k_compact_count, dim3(nb), dim3(256), 0, st, T->idx[T->sorted], (uint32_t)T->n, b0, b1, T->shard_blk);
            hipLaunchKernelGGL(k_rowscan, dim3(1), dim3(256), 0, st, T->shard_blk, nb, T->shard_blk + nb);
            hipLaunchKernelGGL(k_compact_scatter, dim3(nb), dim3(256), 0, st, T->idx[T->sorted], (uint32_t)T->n, b0, b1,
                               T->shard_blk, T->shard_order, m);
            order = T->shard_order;
        }
    }
    return walk_launch(ctx, T, order, m, posm_in, vel_in, acc_out, (uint32_t)ctx->i_begin, 0);
}

// Errors raised on the device by the octree kernels since the last check (capacity, traversal
// stack); for the blocking entry points, so that a truncated traversal cannot pass unnoticed.
int nb_tree_check(nbody_ctx* ctx) {
    Octree* T = ctx->tree;
    if (!T || !T->have_tree) return NBODY_OK;
    return check_err_flag(ctx, T, "Barnes-Hut step");
}

// Targets = the bodies at sorted ranks [rank_begin, rank_begin + rank_count) of the built tree;
// acc_ranked[rank] receives their accelerations.
int nb_tree_walk_ranked(nbody_ctx* ctx, const float4* posm_in, const float4* vel_in, float4* acc_ranked,
                        uint64_t rank_begin, uint64_t rank_count) {
    Octree* T = ctx->tree;
    if (!T->have_tree || !T->built) return nb_fail(ctx, NBODY_ERR_STATE, "ranked traversal needs an octree built from the bodies");
    if (rank_begin + rank_count > T->n) return nb_fail(ctx, NBODY_ERR_INVALID, "rank range outside the bodies");
    if (rank_count == 0) return NBODY_OK;
    return walk_launch(ctx, T, T->idx[T->sorted] + rank_begin, (uint32_t)rank_count, posm_in, vel_in, acc_ranked,
                       (uint32_t)rank_begin, 1);
}

// acc_out[body at rank r] = acc_ranked[r] for every rank of the built tree.
int nb_tree_unrank(nbody_ctx* ctx, const float4* acc_ranked, float4* acc_out) {
    Octree* T = ctx->tree;
    if (!T->have_tree || !T->built) return nb_fail(ctx, NBODY_ERR_STATE, "no octree built from the bodies");
    const uint32_t n = (uint32_t)T->n;
    hipLaunchKernelGGL(k_unrank, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, T->idx[T->sorted], acc_ranked, n, acc_out);
    NB_HIP(ctx, hipGetLastError());
    return NBODY_OK;
}

int nb_tree_download(nbody_ctx* ctx, nbody_octree_node* host_nodes, uint64_t cap, uint64_t* n_nodes) {
    Octree* T = ctx->tree;
    if (!T->have_tree) return nb_fail(ctx, NBODY_ERR_STATE, "no octree to download");
    if (n_nodes) *n_nodes = T->n_nodes;
    if (!host_nodes) return NBODY_OK;
    if (cap < T->n_nodes) return nb_fail(ctx, NBODY_ERR_CAPACITY, "nbody_download_octree: host buffer too small");
    void* dev = nullptr;
    size_t bytes = 0;
    int r = nb_tree_device_nodes(ctx, &dev, &bytes);
    if (r) return r;
    NB_HIP(ctx, hipMemcpyAsync(host_nodes, dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
    NB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return check_err_flag(ctx, T, "octree build");
}

int nb_tree_device_nodes(nbody_ctx* ctx, void** ptr, size_t* bytes) {
    Octree* T = ctx->tree;
    if (!T->have_tree) return nb_fail(ctx, NBODY_ERR_STATE, "no octree");
    if (T->dom && !T->export_is_upload)
        return nb_fail(ctx, NBODY_ERR_STATE, "a domain-decomposed tree is not exported in the 96-byte layout (nbody_bh_set_domains(ctx, 0, 0) first)");
    if (T->export_cap < T->n_nodes) {
        if (T->export_is_upload) return nb_fail(ctx, NBODY_ERR_STATE, "octree export buffer lost");
        if (T->export_buf) NB_HIP(ctx, hipFree(T->export_buf));
        T->export_buf = nullptr;
        T->export_cap = 0;
        NB_HIP(ctx, hipMalloc(&T->export_buf, T->cap_nodes * sizeof(nbody_octree_node)));
        T->export_cap = T->cap_nodes;
    }
    const uint32_t cnt = (uint32_t)T->n_nodes;
    if (!T->export_is_upload) {  // an uploaded tree is kept as it came (renumbered); a built one is exported
        hipLaunchKernelGGL(k_export, dim3((cnt + 255) / 256), dim3(256), 0, ctx->stream, T->nodes, T->bmin, T->bmax,
                           T->range, T->node_head, T->keys[T->sorted], T->lvl_start, (int)T->D, cnt, T->octree_rs ? 1 : 0,
                           T->export_buf);
        NB_HIP(ctx, hipGetLastError());
    }
    *ptr = T->export_buf;
    if (bytes) *bytes = T->n_nodes * sizeof(nbody_octree_node);
    return NBODY_OK;
}

int nb_tree_upload(nbody_ctx* ctx, const nbody_octree_node* host_nodes, uint64_t n_nodes) {
    Octree* T = ctx->tree;
    // The traversal wants the children of a node (and the members of a leaf list) consecutive.
    // Renumber breadth-first on the host, validating the indices on the way: the reference's
    // allocation order (src/octree_maxdepth.rs:188-194) does not have that property.
    std::vector<uint32_t> new_id(n_nodes, NO_NODE);
    std::vector<uint32_t> bfs;  // bfs[new] = old
    bfs.reserve(n_nodes);
    new_id[0] = 0;
    bfs.push_back(0);
    for (size_t head = 0; head < bfs.size(); ++head) {
        const nbody_octree_node& nd = host_nodes[bfs[head]];
        if (nd.node_type > NBODY_NODE_INTERIOR) return nb_fail(ctx, NBODY_ERR_INVALID, "nbody_upload_octree: bad node_type");
        if (nd.node_type == NBODY_NODE_INTERIOR) {
            size_t children = 0;
            for (int k = 0; k < 8; ++k) {
                const uint32_t c = nd.child_indices[k];
                if (c == 0) continue;
                if (c >= n_nodes) return nb_fail(ctx, NBODY_ERR_INVALID, "nbody_upload_octree: child index outside the node array");
                if (new_id[c] != NO_NODE) return nb_fail(ctx, NBODY_ERR_INVALID, "nbody_upload_octree: a node has two parents");
                new_id[c] = (uint32_t)bfs.size();
                bfs.push_back(c);
                ++children;
            }
            // (the builders never make one: src/octree_maxdepth.rs:139-184 gives an interior node the new body's child)
            if (!children) return nb_fail(ctx, NBODY_ERR_INVALID, "nbody_upload_octree: an INTERIOR node has no children");
        } else if (nd.node_type == NBODY_NODE_LEAFLIST) {
            const uint64_t first = nd.child_indices[0], cnt = std::max<uint32_t>(nd.child_indices[1], 1u);
            if (first + cnt > n_nodes) return nb_fail(ctx, NBODY_ERR_INVALID, "nbody_upload_octree: leaf list outside the node array");
            for (uint64_t c = first; c < first + cnt; ++c) {
                if (c == 0 || new_id[c] != NO_NODE) return nb_fail(ctx, NBODY_ERR_INVALID, "nbody_upload_octree: a node has two parents");
                if (host_nodes[c].node_type != NBODY_NODE_LEAFBODY)
                    return nb_fail(ctx, NBODY_ERR_INVALID, "nbody_upload_octree: leaf-list members must be LEAFBODY nodes");
                new_id[c] = (uint32_t)bfs.size();
                bfs.push_back((uint32_t)c);
            }
        }
    }
    const uint64_t kept = bfs.size();
    std::vector<nbody_octree_node> renum(kept);
    for (uint64_t i = 0; i < kept; ++i) {
        nbody_octree_node nd = host_nodes[bfs[i]];
        if (nd.node_type == NBODY_NODE_INTERIOR) {
            for (int k = 0; k < 8; ++k)
                if (nd.child_indices[k]) nd.child_indices[k] = new_id[nd.child_indices[k]];
        } else if (nd.node_type == NBODY_NODE_LEAFLIST) {
            nd.child_indices[0] = new_id[nd.child_indices[0]];
        }
        renum[i] = nd;
    }
    int r = ensure_nodes(ctx, T, kept);
    if (r) return r;
    if (T->export_cap < kept) {
        if (T->export_buf) NB_HIP(ctx, hipFree(T->export_buf));
        T->export_buf = nullptr;
        T->export_cap = 0;
        NB_HIP(ctx, hipMalloc(&T->export_buf, T->cap_nodes * sizeof(nbody_octree_node)));
        T->export_cap = T->cap_nodes;
    }
    NB_HIP(ctx, hipMemcpyAsync(T->export_buf, renum.data(), kept * sizeof(nbody_octree_node), hipMemcpyHostToDevice, ctx->stream));
    const uint32_t cnt = (uint32_t)kept;
    hipLaunchKernelGGL(k_import, dim3((cnt + 255) / 256), dim3(256), 0, ctx->stream, T->export_buf, cnt, T->nodes, T->bmin,
                       T->bmax, T->range, mac_scale(ctx), T->err_flag);
    NB_HIP(ctx, hipGetLastError());
    NB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    T->n_nodes = kept;
    T->have_tree = true;
    T->built = false;  // no sorted order for this tree: targets are walked in index order
    T->export_is_upload = true;
    ctx->counters.octree_nodes = kept;
    return check_err_flag(ctx, T, "nbody_upload_octree");
}

// The grouping of the traversal's targets into wavefronts (64 consecutive entries each): body
// indices, every body exactly once, NO_NODE for a lane left empty.  Null: back to the sorted order.
int nb_tree_set_target_order(nbody_ctx* ctx, const uint32_t* host_order, uint64_t count) {
    Octree* T = ctx->tree;
    if (T->user_order) NB_HIP(ctx, hipFree(T->user_order));
    T->user_order = nullptr;
    T->user_m = 0;
    T->perm_nblk = 0;
    if (!host_order) return NBODY_OK;
    if (count < T->n || count > 4 * T->n + 64) return nb_fail(ctx, NBODY_ERR_INVALID, "nbody_bh_set_target_order: count must be in [n_bodies, 4 n_bodies + 64]");
    std::vector<uint8_t> seen(T->n, 0);
    uint64_t bodies = 0;
    for (uint64_t r = 0; r < count; ++r) {
        const uint32_t b = host_order[r];
        if (b == NO_NODE) continue;
        if (b >= T->n || seen[b]) return nb_fail(ctx, NBODY_ERR_INVALID, "nbody_bh_set_target_order: not a permutation of the body indices");
        seen[b] = 1;
        ++bodies;
    }
    if (bodies != T->n) return nb_fail(ctx, NBODY_ERR_INVALID, "nbody_bh_set_target_order: every body must appear once");
    NB_HIP(ctx, hipMalloc(&T->user_order, count * sizeof(uint32_t)));
    NB_HIP(ctx, hipMemcpyAsync(T->user_order, host_order, count * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    NB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    T->user_m = count;
    return NBODY_OK;
}

int nb_tree_keys(nbody_ctx* ctx, uint64_t* keys, uint8_t* depth, uint32_t* order) {
    Octree* T = ctx->tree;
    if (!T->have_tree || !T->built) return nb_fail(ctx, NBODY_ERR_STATE, "nbody_morton_keys needs a tree built by nbody_build_octree");
    if (T->dom && depth) return nb_fail(ctx, NBODY_ERR_STATE, "leaf depths are not kept for a domain-decomposed tree (pass NULL)");
    const uint64_t n = T->n;
    std::vector<uint64_t> k(n);
    std::vector<uint32_t> id(n);
    std::vector<uint8_t> tp(n), fl(n);
    NB_HIP(ctx, hipMemcpyAsync(k.data(), T->keys[T->sorted], n * 8, hipMemcpyDeviceToHost, ctx->stream));
    NB_HIP(ctx, hipMemcpyAsync(id.data(), T->idx[T->sorted], n * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (depth) {
        NB_HIP(ctx, hipMemcpyAsync(tp.data(), T->top, n, hipMemcpyDeviceToHost, ctx->stream));
        NB_HIP(ctx, hipMemcpyAsync(fl.data(), T->flags, n, hipMemcpyDeviceToHost, ctx->stream));
    }
    NB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (uint64_t r = 0; r < n; ++r) {
        const uint32_t b = id[r];
        if (keys) keys[b] = k[r];
        if (depth) depth[b] = (fl[r] & 4u) ? (uint8_t)T->D : tp[r];
        if (order) order[r] = b;
    }
    return NBODY_OK;
}
