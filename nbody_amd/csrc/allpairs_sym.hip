// All-pairs PLUMMER forces evaluated once per UNORDERED pair (Newton's third law).
//
// The plain kernel (allpairs.hip) spends 12 packed lane-ops + 1 v_rsq_f32 per ordered pair.  The
// distance, r^2 and r^-3 of a pair serve both bodies; what differs is the mass factor and the sign.
// Bodies are grouped in superblocks of 256 * P (one workgroup: 4 wavefronts x 64 lanes x P targets
// held in registers, two per packed register pair; P = 16 by default).  A work item is
// (superblock a, a run of 64-body source tiles that lie in LATER superblocks).  Per source tile
// every lane meets each of the 64 sources once: at step t lane l reads source (l + t) mod 64 from
// LDS (conflict-free), updates its P targets' accelerations in registers, and adds the P
// reactions on that source to the wavefront's own LDS accumulator by a plain read-add-write
// (lane-distinct addresses, one wavefront per accumulator, LDS operations of a wavefront execute
// in order; ds_add_f32 was measured an order of magnitude slower here).  After the tile the four
// wavefronts' accumulators are summed in a fixed order and written as that item's partial row; a
// final kernel adds, per body and in a fixed order, the partial rows of all earlier superblocks,
// the target-side partials of its own superblock's items and the within-superblock term.  No
// atomics reach global memory, so results are reproducible bit for bit.
//
// Per step (P pair-interactions = 2 P ordered interactions) a lane issues 7 P v_pk_*_f32,
// P v_rsq_f32 and 7 instructions that do not depend on P (the reaction's adds): about 18 cycles
// per ordered interaction at P = 16 against 34 for the plain kernel.
#include "ctx.h"
#include <algorithm>
#include <cstdlib>
#include <vector>

namespace {

typedef float f2 __attribute__((ext_vector_type(2)));

constexpr int TILE = 64;  // sources per tile

struct SymItem {
    uint32_t a;           // target superblock
    uint32_t tile_begin;  // source tiles [tile_begin, tile_end)
    uint32_t tile_end;
    uint32_t pad;
};

// Which part evaluates a work item: group g = item / nparts deals its nparts items out starting
// at part g mod nparts, so that no part always gets the same position within a group.
__device__ __forceinline__ uint32_t sym_owner(uint32_t item, uint32_t nparts) {
    return (item % nparts + nparts - (item / nparts) % nparts) % nparts;
}

// UNIFORM: every body has the same mass, which is then applied once per body by the reduction
// instead of twice per pair here.
// P: targets per lane (superblock = 256 * P bodies), two per packed register pair.
template <bool UNIFORM, int P>
__global__ __launch_bounds__(256) void k_ap_sym(const float4* __restrict__ posm, int n, const SymItem* __restrict__ items,
                                                const unsigned long long* __restrict__ rowoff,
                                                float4* __restrict__ part_j, float4* __restrict__ part_i, float eps2,
                                                uint32_t part, uint32_t nparts, uint32_t n_items,
                                                unsigned long long* __restrict__ clk) {
    // One workgroup in the middle of the dispatch stamps the shader-cycle counter and the
    // 100 MHz real-time counter around its work item: their ratio is the clock the chip held.
    const bool stamp = clk != nullptr && blockIdx.x == gridDim.x / 2 && threadIdx.x == 0;
    unsigned long long c0 = 0, r0 = 0;
    if (stamp) {
        c0 = __builtin_amdgcn_s_memtime();
        r0 = __builtin_amdgcn_s_memrealtime();
    }
    // sources are stored twice (entries k and k+64) so that lane + t never needs a modulo
    __shared__ float4 tile[2][2 * TILE];
    __shared__ float4 accj[3][4][2 * TILE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int SB = 256 * P, TILES_PER_SB = SB / TILE, H = P / 2;
    // with nparts > 1 (one part per GPU) this launch takes one work item of every nparts
    // consecutive ones, rotating which (see sym_owner)
    const uint32_t item_id = blockIdx.x * nparts + (part + blockIdx.x) % nparts;
    if (item_id >= n_items) return;
    const SymItem it = items[item_id];
    const int ibase = (int)it.a * SB + wave * (64 * P) + lane;

    f2 px[H], py[H], pz[H], pm[H], ax[H], ay[H], az[H];
#pragma unroll
    for (int h = 0; h < H; ++h) {
        const int i0 = ibase + (2 * h) * 64, i1 = i0 + 64;
        // padding bodies: zero mass and so far away that r^-3 underflows to exactly 0
        const float4 far = make_float4(1e18f, 1e18f, 1e18f, 0.0f);
        float4 p0 = posm[min(i0, n - 1)], p1 = posm[min(i1, n - 1)];
        if (i0 >= n) p0 = far;
        if (i1 >= n) p1 = far;
        px[h] = f2{p0.x, p1.x};
        py[h] = f2{p0.y, p1.y};
        pz[h] = f2{p0.z, p1.z};
        pm[h] = f2{-p0.w, -p1.w};  // negated: the reaction on a source is -m_i r^-3 d
        ax[h] = ay[h] = az[h] = f2{0.0f, 0.0f};
    }

    auto load_tile = [&](uint32_t t) {
        const int j = (int)t * TILE + lane;
        float4 s = posm[min(j, n - 1)];
        if (j >= n) s = make_float4(1e18f, 1e18f, 1e18f, 0.0f);  // padding source, as above
        return s;
    };
    auto zero_acc = [&](int b) {
        accj[b][wave][lane] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        accj[b][wave][lane + TILE] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    };
    auto flush = [&](uint32_t t, int b) {
        // one wavefront sums the four accumulators (fixed order) and writes the tile's partial row
        float sx = 0.0f, sy = 0.0f, sz = 0.0f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float4 u = accj[b][w][lane], v = accj[b][w][lane + TILE];
            sx += u.x + v.x;
            sy += u.y + v.y;
            sz += u.z + v.z;
        }
        const unsigned long long row = rowoff[it.a] + ((unsigned long long)t - (unsigned long long)(it.a + 1) * TILES_PER_SB) * TILE;
        part_j[row + lane] = make_float4(sx, sy, sz, 0.0f);
    };

    float4 pre = make_float4(0, 0, 0, 0);
    if (wave == 0) pre = load_tile(it.tile_begin);
    int tb = 0;  // tile buffer
    int ab = 0;  // accumulator buffer (three deep: one being filled, one being flushed, one being zeroed)
    for (uint32_t t = it.tile_begin; t < it.tile_end; ++t) {
        if (wave == 0) {
            tile[tb][lane] = pre;
            tile[tb][lane + TILE] = pre;
        }
        zero_acc(ab);
        __syncthreads();
        if (wave == 0 && t + 1 < it.tile_end) pre = load_tile(t + 1);
        if (t > it.tile_begin && wave == (int)((t - 1) & 3u)) flush(t - 1, (ab + 2) % 3);

        const float4* tp = &tile[tb][lane];
        float4* ap = &accj[ab][wave][lane];
        float4 nxt = tp[0];
#pragma unroll 8
        for (int s = 0; s < TILE; ++s) {
            const float4 src = nxt;
            nxt = tp[s + 1];  // read ahead of this step's fence (entry lane+64 exists: the tile is stored twice)
            f2 cx, cy, cz;
#pragma unroll
            for (int h = 0; h < H; ++h) {
                const f2 dx = src.x - px[h], dy = src.y - py[h], dz = src.z - pz[h];
                const f2 r2 = __builtin_elementwise_fma(dx, dx, __builtin_elementwise_fma(dy, dy, __builtin_elementwise_fma(dz, dz, (f2)eps2)));
                f2 rinv;
                rinv.x = __builtin_amdgcn_rsqf(r2.x);
                rinv.y = __builtin_amdgcn_rsqf(r2.y);
                const f2 r3 = (rinv * rinv) * rinv;
                const f2 si = UNIFORM ? r3 : src.w * r3;   // on the targets: m_j r^-3 d
                const f2 sj = UNIFORM ? -r3 : pm[h] * r3;  // on the source: -m_i r^-3 d
                ax[h] = __builtin_elementwise_fma(si, dx, ax[h]);
                ay[h] = __builtin_elementwise_fma(si, dy, ay[h]);
                az[h] = __builtin_elementwise_fma(si, dz, az[h]);
                if (h == 0) {
                    cx = sj * dx;
                    cy = sj * dy;
                    cz = sj * dz;
                } else {
                    cx = __builtin_elementwise_fma(sj, dx, cx);
                    cy = __builtin_elementwise_fma(sj, dy, cy);
                    cz = __builtin_elementwise_fma(sj, dz, cz);
                }
            }
            float4 aj = ap[s];
            aj.x += cx.x + cx.y;
            aj.y += cy.x + cy.y;
            aj.z += cz.x + cz.y;
            ap[s] = aj;
            // the entry this lane reads at step s+1 is the one its neighbour has just written:
            // keep the wavefront's LDS accesses in program order across steps
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        tb ^= 1;
        ab = (ab + 1) % 3;
    }
    __syncthreads();
    if (it.tile_end > it.tile_begin && wave == (int)((it.tile_end - 1) & 3u)) flush(it.tile_end - 1, (ab + 2) % 3);

    float4* pi = part_i + (size_t)item_id * SB + wave * (64 * P) + lane;
#pragma unroll
    for (int h = 0; h < H; ++h) {
        pi[(2 * h) * 64] = make_float4(ax[h].x, ay[h].x, az[h].x, 0.0f);
        pi[(2 * h + 1) * 64] = make_float4(ax[h].y, ay[h].y, az[h].y, 0.0f);
    }
    if (stamp) {
        clk[0] = __builtin_amdgcn_s_memtime() - c0;
        clk[1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
}

// acc_out[j] = G * ( diag[j] + sum over earlier superblocks a of part_j[a][j]
//                             + sum over the items of j's superblock of part_i[item][j] ), in f64.
template <bool SINGLE>  // SINGLE: nparts == 1, every row counts and the loads are issued in batches
__global__ void k_ap_sym_reduce(const float4* __restrict__ diag, const float4* __restrict__ part_j,
                                const float4* __restrict__ part_i, const unsigned long long* __restrict__ rowoff,
                                const uint32_t* __restrict__ item_start, int n, int SB,
                                const uint32_t* __restrict__ seg_of, float G,
                                float pair_scale, uint32_t part, uint32_t nparts, int own_begin, int own_end,
                                int diag_slices, float4* __restrict__ acc_out) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const int b = j / SB;
    double x = 0.0, y = 0.0, z = 0.0;
    constexpr int U = SINGLE ? 4 : 1;
    int a = 0;
    if (SINGLE) {
        for (; a + U <= b; a += U) {
            float4 p[U];
#pragma unroll
            for (int u = 0; u < U; ++u) p[u] = part_j[rowoff[a + u] + (unsigned long long)(j - (a + u + 1) * SB)];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                x += p[u].x;
                y += p[u].y;
                z += p[u].z;
            }
        }
    }
    for (; a < b; ++a) {
        // the work item that covered (a, j's tile); with nparts > 1 only this part's items count
        const uint32_t item = item_start[a] + (uint32_t)((j - (a + 1) * SB) / TILE) / seg_of[a];
        if (!SINGLE && sym_owner(item, nparts) != part) continue;
        const float4 p = part_j[rowoff[a] + (unsigned long long)(j - (a + 1) * SB)];
        x += p.x;
        y += p.y;
        z += p.z;
    }
    const uint32_t i0 = item_start[b], i1 = item_start[b + 1];
    uint32_t i = i0;
    if (SINGLE) {
        for (; i + U <= i1; i += U) {
            float4 p[U];
#pragma unroll
            for (int u = 0; u < U; ++u) p[u] = part_i[(size_t)(i + u) * SB + (j - b * SB)];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                x += p[u].x;
                y += p[u].y;
                z += p[u].z;
            }
        }
    }
    for (; i < i1; ++i) {
        if (!SINGLE && sym_owner(i, nparts) != part) continue;
        const float4 p = part_i[(size_t)i * SB + (j - b * SB)];
        x += p.x;
        y += p.y;
        z += p.z;
    }
    // pair_scale: the common mass when the pair kernel left the masses out; the within-superblock
    // term always carries its masses; it is added by the part that owns the body
    x *= (double)pair_scale;
    y *= (double)pair_scale;
    z *= (double)pair_scale;
    if (j >= own_begin && j < own_end) {
        for (int s = 0; s < diag_slices; ++s) {
            const float4 d = diag[(size_t)s * n + j];
            x += d.x;
            y += d.y;
            z += d.z;
        }
    }
    acc_out[j] = make_float4(G * (float)x, G * (float)y, G * (float)z, 0.0f);
}

}  // namespace

struct SymPlan {
    uint64_t n = 0;
    uint32_t sb = 0;   // bodies per superblock
    uint32_t seg = 0;  // source tiles per work item of the first superblocks (later ones: fewer, see sym_seg_of)
    uint32_t* seg_of = nullptr;  // [superblocks] source tiles per work item
    uint32_t n_items = 0;
    SymItem* items = nullptr;
    unsigned long long* rowoff = nullptr;
    uint32_t* item_start = nullptr;
    float4* part_j = nullptr;
    float4* part_i = nullptr;
    float4* diag = nullptr;
};

static void sym_free(SymPlan* P);

void nb_sym_destroy(nbody_ctx* ctx) {
    sym_free(ctx->sym);
    ctx->sym = nullptr;
}

// Source tiles per work item for every superblock.  Work items are dispatched in superblock
// order and the GPU holds only ~1000 of them at a time, so the last ones decide how long the
// kernel's tail is: superblocks that start after 70 / 85 / 95 % of all the pair work get items of
// 1/2, 1/4, 1/8 the size.
static std::vector<uint32_t> sym_seg_of(uint64_t n, uint32_t seg_tiles, uint32_t SB) {
    const uint64_t nsb = (n + SB - 1) / SB, nt = (n + TILE - 1) / TILE, TPS = SB / TILE;
    std::vector<uint32_t> seg(nsb, seg_tiles);
    uint64_t total = 0;
    for (uint64_t a = 0; a < nsb; ++a) total += nt > (a + 1) * TPS ? nt - (a + 1) * TPS : 0;
    uint64_t done = 0;
    for (uint64_t a = 0; a < nsb; ++a) {
        const double frac = total ? (double)done / (double)total : 0.0;
        static const bool taper = !getenv("NBODY_AP_NO_TAPER");  // (for measurements)
        const uint32_t div = !taper ? 1u : frac < 0.70 ? 1u : frac < 0.85 ? 2u : frac < 0.95 ? 4u : 8u;
        seg[a] = std::max(4u, seg_tiles / div);
        done += nt > (a + 1) * TPS ? nt - (a + 1) * TPS : 0;
    }
    return seg;
}

// Bytes of scratch the symmetric path needs for n bodies (0 if it does not apply).
uint64_t nb_sym_scratch_bytes(uint64_t n, uint32_t seg_tiles, uint32_t SB) {
    const uint64_t nsb = (n + SB - 1) / SB, nt = (n + TILE - 1) / TILE, TILES_PER_SB = SB / TILE;
    if (nsb < 2) return 0;
    const std::vector<uint32_t> seg = sym_seg_of(n, seg_tiles, SB);
    uint64_t rows = 0, items = 0;
    for (uint64_t a = 0; a + 1 < nsb; ++a) {
        const uint64_t tiles = nt - (a + 1) * TILES_PER_SB;
        rows += tiles * TILE;
        items += (tiles + seg[a] - 1) / seg[a];
    }
    return (rows + items * SB + (SB / NB_DIAG_TILE) * n) * sizeof(float4);
}

static void sym_free(SymPlan* P) {
    if (!P) return;
    void* ptrs[] = {P->items, P->rowoff, P->item_start, P->seg_of, P->part_j, P->part_i, P->diag};
    for (void* p : ptrs)
        if (p) hipFree(p);
    delete P;
}

// The plan is attached to the context only once every buffer of it exists: a failed allocation
// leaves the context without a plan, so the next call plans (and fails, or succeeds) afresh.
static int sym_prepare(nbody_ctx* ctx, uint32_t seg_tiles, uint32_t SB) {
    if (ctx->sym && ctx->sym->n == ctx->n && ctx->sym->sb == SB && ctx->sym->seg == seg_tiles) return NBODY_OK;
    nb_sym_destroy(ctx);
    SymPlan* P = new SymPlan;
    P->n = ctx->n;
    P->sb = SB;
    P->seg = seg_tiles;
    const uint32_t TILES_PER_SB = SB / TILE;
    const uint64_t n = ctx->n;
    const uint32_t nsb = (uint32_t)((n + SB - 1) / SB), nt = (uint32_t)((n + TILE - 1) / TILE);
    std::vector<SymItem> items;
    const std::vector<uint32_t> seg_of = sym_seg_of(n, seg_tiles, SB);
    std::vector<unsigned long long> rowoff(nsb + 1, 0);
    std::vector<uint32_t> item_start(nsb + 1, 0);
    unsigned long long rows = 0;
    for (uint32_t a = 0; a < nsb; ++a) {
        rowoff[a] = rows;
        item_start[a] = (uint32_t)items.size();
        const uint32_t first = (a + 1) * TILES_PER_SB;
        if (first < nt) {
            for (uint32_t t = first; t < nt; t += seg_of[a]) items.push_back(SymItem{a, t, std::min(nt, t + seg_of[a]), 0});
            rows += (unsigned long long)(nt - first) * TILE;
        }
    }
    rowoff[nsb] = rows;
    item_start[nsb] = (uint32_t)items.size();
    P->n_items = (uint32_t)items.size();
    hipError_t e = hipMalloc(&P->items, std::max<size_t>(1, items.size()) * sizeof(SymItem));
    if (e == hipSuccess) e = hipMalloc(&P->rowoff, rowoff.size() * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMalloc(&P->item_start, item_start.size() * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc(&P->seg_of, seg_of.size() * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc(&P->part_j, std::max<unsigned long long>(1, rows) * sizeof(float4));
    if (e == hipSuccess) e = hipMalloc(&P->part_i, std::max<size_t>(1, items.size()) * SB * sizeof(float4));
    if (e == hipSuccess) e = hipMalloc(&P->diag, (size_t)(SB / NB_DIAG_TILE) * n * sizeof(float4));
    if (e != hipSuccess) {
        (void)hipGetLastError();
        sym_free(P);
        return nb_fail(ctx, NBODY_ERR_NOMEM, std::string("pair kernel scratch: ") + hipGetErrorString(e));
    }
    e = hipMemcpyAsync(P->seg_of, seg_of.data(), seg_of.size() * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(P->items, items.data(), items.size() * sizeof(SymItem), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess)
        e = hipMemcpyAsync(P->rowoff, rowoff.data(), rowoff.size() * sizeof(unsigned long long), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess)
        e = hipMemcpyAsync(P->item_start, item_start.data(), item_start.size() * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);  // the host vectors go out of scope
    if (e != hipSuccess) {
        sym_free(P);
        return nb_fail(ctx, NBODY_ERR_HIP, std::string("pair kernel plan upload: ") + hipGetErrorString(e));
    }
    ctx->sym = P;
    return NBODY_OK;
}

// PLUMMER forces on all n bodies.  tpl: targets per lane (4, 8 or 16).  part/nparts: evaluate every
// nparts-th work item only; acc_out then holds this part's contribution to EVERY body (the
// within-superblock term only for the context's own targets) and the parts must be summed.
int nb_allpairs_forces_sym(nbody_ctx* ctx, const float4* posm_in, float4* acc_out, uint32_t seg_tiles, int tpl,
                           uint32_t part, uint32_t nparts) {
    tpl = (tpl == 16 || tpl == 8) ? tpl : 4;
    const uint32_t SB = 256u * (uint32_t)tpl;
    int r = sym_prepare(ctx, seg_tiles, SB);
    if (r) return r;
    SymPlan* P = ctx->sym;
    const int n = (int)ctx->n;
    nb_time_begin(ctx, 0);
    const bool uniform = ctx->uniform_mass;
    if (P->n_items) {
#define NB_SYM_LAUNCH(U, T)                                                                                          \
    hipLaunchKernelGGL((k_ap_sym<U, T>), dim3((P->n_items + nparts - 1) / nparts), dim3(256), 0, ctx->stream, posm_in, n, \
                       P->items, P->rowoff, P->part_j, P->part_i, ctx->cfg.eps2, part, nparts, P->n_items, ctx->clk_dev)
        if (uniform && tpl == 16) NB_SYM_LAUNCH(true, 16);
        else if (tpl == 16) NB_SYM_LAUNCH(false, 16);
        else if (uniform && tpl == 8) NB_SYM_LAUNCH(true, 8);
        else if (uniform) NB_SYM_LAUNCH(true, 4);
        else if (tpl == 8) NB_SYM_LAUNCH(false, 8);
        else NB_SYM_LAUNCH(false, 4);
#undef NB_SYM_LAUNCH
    }
    nb_time_end(ctx, 0);
    NB_HIP(ctx, hipGetLastError());
    r = nb_allpairs_diag(ctx, posm_in, P->diag, tpl);
    if (r) return r;
    if (nparts == 1)
        hipLaunchKernelGGL(k_ap_sym_reduce<true>, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, P->diag, P->part_j, P->part_i,
                           P->rowoff, P->item_start, n, (int)SB, P->seg_of, ctx->cfg.G, uniform ? ctx->mass0 : 1.0f, part, nparts,
                           (int)ctx->i_begin, (int)(ctx->i_begin + ctx->i_count), (int)(SB / NB_DIAG_TILE), acc_out);
    else
        hipLaunchKernelGGL(k_ap_sym_reduce<false>, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, P->diag, P->part_j, P->part_i,
                           P->rowoff, P->item_start, n, (int)SB, P->seg_of, ctx->cfg.G, uniform ? ctx->mass0 : 1.0f, part, nparts,
                           (int)ctx->i_begin, (int)(ctx->i_begin + ctx->i_count), (int)(SB / NB_DIAG_TILE), acc_out);
    NB_HIP(ctx, hipGetLastError());
    ctx->counters.interactions_allpairs += ctx->n * ctx->n / nparts;
    ctx->counters.allpairs_path = 2;
    return NBODY_OK;
}
