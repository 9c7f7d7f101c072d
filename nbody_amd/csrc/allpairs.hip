// All-pairs gravity on gfx950: LDS-tiled force kernel, finalize/integrate kernels, energy.
//
// Replaces the compute pass of src/nbody_gpu.rs:418-451 running src/nbody.wgsl:10-139, and the
// rayon loop of src/nbody_cpu.rs:324-349.  The reference streams every source from global memory
// in every thread; here a workgroup stages 512-source tiles {x,y,z,m} in LDS with coalesced
// 16-byte loads (double buffered, one barrier per tile), every lane keeps IPT targets in
// registers, and the LDS read of a source is a wave-wide broadcast.
//
// FP32 VALU bound.  Per PLUMMER interaction the inner loop needs 12 VALU lane-operations
// (3 sub, 3 fma, 3 mul, 3 fma) and one quarter-rate v_rsq_f32; the PLUMMER kernel keeps two
// targets per packed register pair so that all twelve are v_pk_*_f32 (measured on gfx950:
// 6 v_pk_fma_f32 + 1 v_rsq_f32 take 34 cycles, 12 v_fma_f32 + 1 v_rsq_f32 take 44).
#include "ctx.h"
#include "laws.h"
#include <algorithm>
#include <cmath>
#include <cstdlib>

namespace {

constexpr int BLOCK = 256;  // 4 wavefronts
constexpr int TJ = 512;     // sources per LDS tile (8 KiB), two tiles per workgroup
constexpr int JL = TJ / BLOCK;

typedef float f2 __attribute__((ext_vector_type(2)));

// Two targets per packed register pair against one source: every VALU instruction of the
// PLUMMER term is a v_pk_*_f32 with the source operand broadcast, plus two v_rsq_f32.
// UNIFORM: every body has the same mass, which is then applied once per target when the slices are
// summed instead of once per term here (eleven packed instructions per pair of terms instead of twelve).
template <bool UNIFORM = false>
__device__ __forceinline__ void interact_plummer_pk(const f2 px, const f2 py, const f2 pz, const float4 s, float eps2,
                                                    f2& ax, f2& ay, f2& az) {
    f2 dx = s.x - px, dy = s.y - py, dz = s.z - pz;
    f2 r2 = __builtin_elementwise_fma(dx, dx, __builtin_elementwise_fma(dy, dy, __builtin_elementwise_fma(dz, dz, (f2)eps2)));
    f2 rinv;
    rinv.x = __builtin_amdgcn_rsqf(r2.x);
    rinv.y = __builtin_amdgcn_rsqf(r2.y);
    f2 sc = UNIFORM ? rinv * (rinv * rinv) : (s.w * rinv) * (rinv * rinv);
    ax = __builtin_elementwise_fma(sc, dx, ax);
    ay = __builtin_elementwise_fma(sc, dy, ay);
    az = __builtin_elementwise_fma(sc, dz, az);
}

// WGSL_BIAS for two targets per packed register pair: src/nbody.wgsl:56-81 with every lane-op
// that has a packed form packed; the six transcendentals per target stay scalar.
__device__ __forceinline__ void interact_bias_pk(const f2 px, const f2 py, const f2 pz, const f2 vx, const f2 vy,
                                                 const f2 vz, const f2 vn, const float4 s, float eps2, f2& ax, f2& ay,
                                                 f2& az) {
    const f2 dx = s.x - px, dy = s.y - py, dz = s.z - pz;
    const f2 r2u = __builtin_elementwise_fma(dx, dx, __builtin_elementwise_fma(dy, dy, dz * dz));
    const f2 r2s = r2u + eps2;
    f2 dist, rinv;
    dist.x = __builtin_amdgcn_sqrtf(r2u.x);
    dist.y = __builtin_amdgcn_sqrtf(r2u.y);
    rinv.x = __builtin_amdgcn_rsqf(r2s.x);
    rinv.y = __builtin_amdgcn_rsqf(r2s.y);
    f2 sc = (s.w * rinv) * (rinv * rinv);
    const f2 vd = __builtin_elementwise_fma(vx, dx, __builtin_elementwise_fma(vy, dy, vz * dz));
    const f2 den = __builtin_elementwise_fma(dist, vn, (f2)1.0f);
    f2 rden;
    rden.x = __builtin_amdgcn_rcpf(den.x);
    rden.y = __builtin_amdgcn_rcpf(den.y);
    const f2 q = vd * rden;
    // acos(q) = sqrt(1-|q|) * P7(|q|), reflected for q < 0 (see fast_acos)
    f2 a;
    a.x = fminf(fabsf(q.x), 1.0f);
    a.y = fminf(fabsf(q.y), 1.0f);
    f2 p = (f2)-0.0012624911f;
    p = __builtin_elementwise_fma(p, a, (f2)0.0066700901f);
    p = __builtin_elementwise_fma(p, a, (f2)-0.0170881256f);
    p = __builtin_elementwise_fma(p, a, (f2)0.0308918810f);
    p = __builtin_elementwise_fma(p, a, (f2)-0.0501743046f);
    p = __builtin_elementwise_fma(p, a, (f2)0.0889789874f);
    p = __builtin_elementwise_fma(p, a, (f2)-0.2145988016f);
    p = __builtin_elementwise_fma(p, a, (f2)1.5707963050f);
    const f2 om = 1.0f - a;
    f2 sq;
    sq.x = __builtin_amdgcn_sqrtf(om.x);
    sq.y = __builtin_amdgcn_sqrtf(om.y);
    const f2 r = sq * p;
    f2 b;
    b.x = q.x < 0.0f ? 3.14159265358979f - r.x : r.x;
    b.y = q.y < 0.0f ? 3.14159265358979f - r.y : r.y;
    f2 lg;
    lg.x = __builtin_amdgcn_logf(b.x);
    lg.y = __builtin_amdgcn_logf(b.y);
    lg = lg * 0.15f;
    f2 bias;
    bias.x = __builtin_amdgcn_exp2f(lg.x);
    bias.y = __builtin_amdgcn_exp2f(lg.y);
    sc = sc * bias;
    ax = __builtin_elementwise_fma(sc, dx, ax);
    ay = __builtin_elementwise_fma(sc, dy, ay);
    az = __builtin_elementwise_fma(sc, dz, az);
}

// grid.x: target blocks of BLOCK*IPT bodies; grid.y: source slices of j_chunk bodies.
// acc_part[slice][i - i_begin] receives the un-scaled (no G) partial sum of the slice.
template <int LAW, int IPT>
__global__ __launch_bounds__(BLOCK) void k_allpairs(const float4* __restrict__ posm, const float4* __restrict__ vel,
                                                     float4* __restrict__ acc_part, int n, int i_begin, int i_count,
                                                     int j_chunk, float eps2) {
    __shared__ float4 tile[2][TJ];
    const int tid = threadIdx.x;
    const int ib = i_begin + blockIdx.x * (BLOCK * IPT);
    const int i_end = i_begin + i_count;

    Tgt t[IPT];
#pragma unroll
    for (int k = 0; k < IPT; ++k) {
        int ig = ib + k * BLOCK + tid;
        int ic = min(ig, i_end - 1);
        float4 p = posm[ic];
        t[k].x = p.x;
        t[k].y = p.y;
        t[k].z = p.z;
        t[k].idx = ig;
        if (LAW == NBODY_LAW_WGSL_BIAS) {
            float4 v = vel[ic];
            t[k].vx = v.x;
            t[k].vy = v.y;
            t[k].vz = v.z;
            // distance(vec3(0), vel), src/nbody.wgsl:75
            t[k].vn = sqrtf(fmaf(v.x, v.x, fmaf(v.y, v.y, v.z * v.z)));
        }
    }

    const int j0 = blockIdx.y * j_chunk;
    const int j1 = min(n, j0 + j_chunk);
    const int ntiles = (j1 - j0 + TJ - 1) / TJ;

    // running sum + Kahan compensation per target; a tile's terms are summed separately first
    float sx[IPT], sy[IPT], sz[IPT], cx[IPT], cy[IPT], cz[IPT];
#pragma unroll
    for (int k = 0; k < IPT; ++k) sx[k] = sy[k] = sz[k] = cx[k] = cy[k] = cz[k] = 0.0f;

    float4 pre[JL];
#pragma unroll
    for (int l = 0; l < JL; ++l) pre[l] = posm[min(j0 + l * BLOCK + tid, j1 - 1)];
#pragma unroll
    for (int l = 0; l < JL; ++l) tile[0][l * BLOCK + tid] = pre[l];
    __syncthreads();

    int buf = 0;
    for (int tl = 0; tl < ntiles; ++tl) {
        const int jb = j0 + tl * TJ;
        // prefetch the next tile into registers (index clamped: the last trip re-reads valid data)
#pragma unroll
        for (int l = 0; l < JL; ++l) pre[l] = posm[min(jb + TJ + l * BLOCK + tid, j1 - 1)];
        const int cnt = min(TJ, j1 - jb);
        float ax[IPT], ay[IPT], az[IPT];
#pragma unroll
        for (int k = 0; k < IPT; ++k) ax[k] = ay[k] = az[k] = 0.0f;

        const bool overlap = (LAW == NBODY_LAW_CPU_R6) && (jb < ib + BLOCK * IPT) && (jb + TJ > ib);
        if (cnt == TJ && !overlap) {
#pragma unroll 8
            for (int jj = 0; jj < TJ; ++jj) {
                const float4 s = tile[buf][jj];
#pragma unroll
                for (int k = 0; k < IPT; ++k) interact<LAW, false>(t[k], s, 0, eps2, ax[k], ay[k], az[k]);
            }
        } else {
            for (int jj = 0; jj < cnt; ++jj) {
                const float4 s = tile[buf][jj];
#pragma unroll
                for (int k = 0; k < IPT; ++k) interact<LAW, true>(t[k], s, jb + jj, eps2, ax[k], ay[k], az[k]);
            }
        }
#pragma unroll
        for (int k = 0; k < IPT; ++k) {
            float y, u;
            y = ax[k] - cx[k]; u = sx[k] + y; cx[k] = (u - sx[k]) - y; sx[k] = u;
            y = ay[k] - cy[k]; u = sy[k] + y; cy[k] = (u - sy[k]) - y; sy[k] = u;
            y = az[k] - cz[k]; u = sz[k] + y; cz[k] = (u - sz[k]) - y; sz[k] = u;
        }
#pragma unroll
        for (int l = 0; l < JL; ++l) tile[buf ^ 1][l * BLOCK + tid] = pre[l];
        __syncthreads();
        buf ^= 1;
    }

#pragma unroll
    for (int k = 0; k < IPT; ++k) {
        int ig = ib + k * BLOCK + tid;
        if (ig < i_end) acc_part[(size_t)blockIdx.y * i_count + (ig - i_begin)] = make_float4(sx[k], sy[k], sz[k], 0.0f);
    }
}

// PLUMMER law with two targets per packed register pair (IPT even): same tiling, staging and
// summation scheme as k_allpairs.
template <int IPT, bool BIAS = false, bool UNIFORM = false>
__global__ __launch_bounds__(BLOCK) void k_allpairs_plummer_pk(const float4* __restrict__ posm,
                                                                float4* __restrict__ acc_part, int n, int i_begin,
                                                                int i_count, int j_chunk, float eps2, int diag,
                                                                const float4* __restrict__ vel = nullptr,
                                                                int slice_stride = 0) {
    static_assert(IPT % 2 == 0, "packed kernel needs an even number of targets per lane");
    constexpr int H = IPT / 2;
    __shared__ float4 tile[2][TJ];
    const int tid = threadIdx.x;
    const int ib = i_begin + blockIdx.x * (BLOCK * IPT);
    const int i_end = i_begin + i_count;

    f2 px[H], py[H], pz[H], vx[H], vy[H], vz[H], vn[H];
#pragma unroll
    for (int k = 0; k < H; ++k) {
        const int i0 = min(ib + (2 * k) * BLOCK + tid, i_end - 1), i1 = min(ib + (2 * k + 1) * BLOCK + tid, i_end - 1);
        float4 p0 = posm[i0];
        float4 p1 = posm[i1];
        px[k] = f2{p0.x, p1.x};
        py[k] = f2{p0.y, p1.y};
        pz[k] = f2{p0.z, p1.z};
        if (BIAS) {
            const float4 v0 = vel[i0], v1 = vel[i1];
            vx[k] = f2{v0.x, v1.x};
            vy[k] = f2{v0.y, v1.y};
            vz[k] = f2{v0.z, v1.z};
            // distance(vec3(0), vel), src/nbody.wgsl:75
            vn[k] = f2{sqrtf(fmaf(v0.x, v0.x, fmaf(v0.y, v0.y, v0.z * v0.z))),
                       sqrtf(fmaf(v1.x, v1.x, fmaf(v1.y, v1.y, v1.z * v1.z)))};
        }
    }
    // diag > 0: the sources are the `diag` bodies of the superblock that holds the workgroup's
    // targets (the within-superblock term of the symmetric path), slice blockIdx.y of them; diag
    // is a multiple of BLOCK*IPT and of j_chunk
    const int j0 = (diag ? (ib / diag) * diag : 0) + blockIdx.y * j_chunk;
    const int j1 = min(n, j0 + j_chunk);
    const int ntiles = (j1 - j0 + TJ - 1) / TJ;

    f2 sx[H], sy[H], sz[H], cx[H], cy[H], cz[H];
#pragma unroll
    for (int k = 0; k < H; ++k) sx[k] = sy[k] = sz[k] = cx[k] = cy[k] = cz[k] = f2{0.0f, 0.0f};

    float4 pre[JL];
#pragma unroll
    for (int l = 0; l < JL; ++l) pre[l] = posm[min(j0 + l * BLOCK + tid, j1 - 1)];
#pragma unroll
    for (int l = 0; l < JL; ++l) tile[0][l * BLOCK + tid] = pre[l];
    __syncthreads();

    int buf = 0;
    for (int tl = 0; tl < ntiles; ++tl) {
        const int jb = j0 + tl * TJ;
#pragma unroll
        for (int l = 0; l < JL; ++l) pre[l] = posm[min(jb + TJ + l * BLOCK + tid, j1 - 1)];
        const int cnt = min(TJ, j1 - jb);
        f2 ax[H], ay[H], az[H];
#pragma unroll
        for (int k = 0; k < H; ++k) ax[k] = ay[k] = az[k] = f2{0.0f, 0.0f};
        if (cnt == TJ) {
#pragma unroll 4
            for (int jj = 0; jj < TJ; ++jj) {
                const float4 s = tile[buf][jj];
#pragma unroll
                for (int k = 0; k < H; ++k) {
                    if (BIAS) interact_bias_pk(px[k], py[k], pz[k], vx[k], vy[k], vz[k], vn[k], s, eps2, ax[k], ay[k], az[k]);
                    else interact_plummer_pk<UNIFORM>(px[k], py[k], pz[k], s, eps2, ax[k], ay[k], az[k]);
                }
            }
        } else {
            for (int jj = 0; jj < cnt; ++jj) {
                const float4 s = tile[buf][jj];
#pragma unroll
                for (int k = 0; k < H; ++k) {
                    if (BIAS) interact_bias_pk(px[k], py[k], pz[k], vx[k], vy[k], vz[k], vn[k], s, eps2, ax[k], ay[k], az[k]);
                    else interact_plummer_pk<UNIFORM>(px[k], py[k], pz[k], s, eps2, ax[k], ay[k], az[k]);
                }
            }
        }
#pragma unroll
        for (int k = 0; k < H; ++k) {
            f2 y, u;
            y = ax[k] - cx[k]; u = sx[k] + y; cx[k] = (u - sx[k]) - y; sx[k] = u;
            y = ay[k] - cy[k]; u = sy[k] + y; cy[k] = (u - sy[k]) - y; sy[k] = u;
            y = az[k] - cz[k]; u = sz[k] + y; cz[k] = (u - sz[k]) - y; sz[k] = u;
        }
#pragma unroll
        for (int l = 0; l < JL; ++l) tile[buf ^ 1][l * BLOCK + tid] = pre[l];
        __syncthreads();
        buf ^= 1;
    }
#pragma unroll
    for (int k = 0; k < H; ++k) {
        int ig0 = ib + (2 * k) * BLOCK + tid, ig1 = ig0 + BLOCK;
        float4* dst = acc_part + (size_t)blockIdx.y * (slice_stride ? slice_stride : i_count);
        if (ig0 < i_end) dst[ig0 - i_begin] = make_float4(sx[k].x, sy[k].x, sz[k].x, 0.0f);
        if (ig1 < i_end) dst[ig1 - i_begin] = make_float4(sx[k].y, sy[k].y, sz[k].y, 0.0f);
    }
}

// acc_out[i] = G * sum over slices (f64), for i in [i_begin, i_begin + i_count)
__global__ void k_finalize(const float4* __restrict__ acc_part, int slices, int i_begin, int i_count, float G,
                           float4* __restrict__ acc_out) {
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= i_count) return;
    double x = 0, y = 0, z = 0;
    for (int s = 0; s < slices; ++s) {
        float4 p = acc_part[(size_t)s * i_count + r];
        x += p.x;
        y += p.y;
        z += p.z;
    }
    acc_out[i_begin + r] = make_float4(G * (float)x, G * (float)y, G * (float)z, 0.0f);
}

// src/nbody.wgsl:132-138 / src/nbodybh.wgsl:163-169, evaluated in the reference's operation
// order with no contraction so that equal accelerations give bit-equal states.
__global__ void k_integrate_verlet(const float4* __restrict__ posm_in, const float4* __restrict__ vel_in,
                                   const float4* __restrict__ acc_in, const float4* __restrict__ acc_new,
                                   float4* __restrict__ posm_out, float4* __restrict__ vel_out,
                                   float4* __restrict__ acc_out, int i_begin, int i_count, float dt, float dt2,
                                   float clamp_world) {
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= i_count) return;
    int i = i_begin + r;
    float4 p = posm_in[i], v = vel_in[i], a0 = acc_in[i], a1 = acc_new[i];
    float px = __fadd_rn(p.x, __fadd_rn(__fmul_rn(v.x, dt), __fmul_rn(__fmul_rn(0.5f, a0.x), dt2)));
    float py = __fadd_rn(p.y, __fadd_rn(__fmul_rn(v.y, dt), __fmul_rn(__fmul_rn(0.5f, a0.y), dt2)));
    float pz = __fadd_rn(p.z, __fadd_rn(__fmul_rn(v.z, dt), __fmul_rn(__fmul_rn(0.5f, a0.z), dt2)));
    float vx = __fadd_rn(v.x, __fmul_rn(__fmul_rn(0.5f, __fadd_rn(a0.x, a1.x)), dt));
    float vy = __fadd_rn(v.y, __fmul_rn(__fmul_rn(0.5f, __fadd_rn(a0.y, a1.y)), dt));
    float vz = __fadd_rn(v.z, __fmul_rn(__fmul_rn(0.5f, __fadd_rn(a0.z, a1.z)), dt));
    if (clamp_world > 0.0f) {
        px = fminf(fmaxf(px, 0.0f), clamp_world);
        py = fminf(fmaxf(py, 0.0f), clamp_world);
        pz = fminf(fmaxf(pz, 0.0f), clamp_world);
    }
    posm_out[i] = make_float4(px, py, pz, p.w);
    vel_out[i] = make_float4(vx, vy, vz, 0.0f);
    acc_out[i] = make_float4(a1.x, a1.y, a1.z, 0.0f);
}

// Kick-drift leapfrog (no counterpart in the reference): v += a dt_kick; x += v dt.
__global__ void k_integrate_leapfrog(const float4* __restrict__ posm_in, const float4* __restrict__ vel_in,
                                     const float4* __restrict__ acc_new, float4* __restrict__ posm_out,
                                     float4* __restrict__ vel_out, float4* __restrict__ acc_out, int i_begin, int i_count,
                                     float dt, float dt_kick, float clamp_world) {
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= i_count) return;
    int i = i_begin + r;
    const float4 p = posm_in[i], v = vel_in[i], a = acc_new[i];
    const float vx = __fadd_rn(v.x, __fmul_rn(a.x, dt_kick));
    const float vy = __fadd_rn(v.y, __fmul_rn(a.y, dt_kick));
    const float vz = __fadd_rn(v.z, __fmul_rn(a.z, dt_kick));
    float px = __fadd_rn(p.x, __fmul_rn(vx, dt));
    float py = __fadd_rn(p.y, __fmul_rn(vy, dt));
    float pz = __fadd_rn(p.z, __fmul_rn(vz, dt));
    if (clamp_world > 0.0f) {
        px = fminf(fmaxf(px, 0.0f), clamp_world);
        py = fminf(fmaxf(py, 0.0f), clamp_world);
        pz = fminf(fmaxf(pz, 0.0f), clamp_world);
    }
    posm_out[i] = make_float4(px, py, pz, p.w);
    vel_out[i] = make_float4(vx, vy, vz, 0.0f);
    acc_out[i] = make_float4(a.x, a.y, a.z, 0.0f);
}

// src/nbody_cpu.rs:345-348: the OTHER buffer set is advanced in place.
__global__ void k_integrate_cpu_xbuf(const float4* __restrict__ acc_new, float4* __restrict__ posm_alt,
                                     float4* __restrict__ vel_alt, float4* __restrict__ acc_alt, int i_begin,
                                     int i_count) {
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= i_count) return;
    int i = i_begin + r;
    float4 a = acc_new[i], v = vel_alt[i], p = posm_alt[i];
    acc_alt[i] = make_float4(a.x, a.y, a.z, 0.0f);
    v.x = __fadd_rn(v.x, a.x);
    v.y = __fadd_rn(v.y, a.y);
    v.z = __fadd_rn(v.z, a.z);
    vel_alt[i] = make_float4(v.x, v.y, v.z, 0.0f);
    posm_alt[i] = make_float4(__fadd_rn(p.x, v.x), __fadd_rn(p.y, v.y), __fadd_rn(p.z, v.z), p.w);
}

// K = sum 0.5 m v^2 ; U = -G/2 sum_i sum_{j != i} m_i m_j / sqrt(r^2 + eps2), in f64.
__global__ void k_energy(const float4* __restrict__ posm, const float4* __restrict__ vel, int n, double eps2,
                         double* __restrict__ out) {
    __shared__ double sk[256], su[256];
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    double k = 0, u = 0;
    if (i < n) {
        float4 p = posm[i], v = vel[i];
        k = 0.5 * (double)p.w * ((double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z);
        for (int j = 0; j < n; ++j) {
            if (j == i) continue;
            float4 q = posm[j];
            double dx = (double)q.x - p.x, dy = (double)q.y - p.y, dz = (double)q.z - p.z;
            u += (double)q.w / sqrt(dx * dx + dy * dy + dz * dz + eps2);
        }
        u *= (double)p.w;
    }
    sk[threadIdx.x] = k;
    su[threadIdx.x] = u;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            sk[threadIdx.x] += sk[threadIdx.x + s];
            su[threadIdx.x] += su[threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        atomicAdd(&out[0], sk[0]);
        atomicAdd(&out[1], su[0]);
    }
}

int env_int(const char* name, int dflt) {
    const char* s = getenv(name);
    return s ? atoi(s) : dflt;
}

// targets per lane of the pair kernel (superblock = 256 of them)
int sym_tpl() {
    // 16: the per-step work that does not depend on the target count (the reaction's LDS
    // read-add-write and its adds) is spread over twice the pairs; measured 5 % faster than 8 at
    // 2 wavefronts per SIMD instead of 4, while 32 (1 wavefront, spills) is slower
    const int t = env_int("NBODY_AP_SYM_TPL", 16);
    return t == 4 || t == 8 ? t : 16;
}

// source tiles per work item of the pair kernel: aim at ~32K items (measured best at N = 1M)
uint32_t sym_seg_tiles(const nbody_ctx* ctx, int tpl) {
    if (env_int("NBODY_AP_SEG", 0) > 0) return (uint32_t)env_int("NBODY_AP_SEG", 0);
    const uint64_t nsb = (ctx->n + 256 * tpl - 1) / (256 * tpl), nt = (ctx->n + 63) / 64;
    return (uint32_t)std::min<uint64_t>(512, std::max<uint64_t>(4, nsb * nt / 2 / 32768));
}

// Everything a launch of the plain (ordered-pair) kernels needs; no context behind it.
struct ApLaunch {
    hipStream_t stream;
    const float4* posm;
    const float4* vel;
    float4* acc_part;  // [slices][i_count]
    int n, i_begin, i_count;
    float eps2;
    float uniform_mass;  // > 0: every body has this mass (PLUMMER: applied once per target, in ap_finalize)
};

template <int LAW, int IPT>
void launch_ap(const ApLaunch& a, int slices, int j_chunk) {
    dim3 grid((unsigned)((a.i_count + BLOCK * IPT - 1) / (BLOCK * IPT)), (unsigned)slices);
    hipLaunchKernelGGL((k_allpairs<LAW, IPT>), grid, dim3(BLOCK), 0, a.stream, a.posm, a.vel, a.acc_part, a.n, a.i_begin,
                       a.i_count, j_chunk, a.eps2);
}

template <int IPT>
void launch_ap_pk(const ApLaunch& a, int slices, int j_chunk) {
    dim3 grid((unsigned)((a.i_count + BLOCK * IPT - 1) / (BLOCK * IPT)), (unsigned)slices);
    if (a.uniform_mass > 0.0f)
        hipLaunchKernelGGL((k_allpairs_plummer_pk<IPT, false, true>), grid, dim3(BLOCK), 0, a.stream, a.posm, a.acc_part, a.n,
                           a.i_begin, a.i_count, j_chunk, a.eps2, 0);
    else
        hipLaunchKernelGGL((k_allpairs_plummer_pk<IPT>), grid, dim3(BLOCK), 0, a.stream, a.posm, a.acc_part, a.n, a.i_begin,
                           a.i_count, j_chunk, a.eps2, 0);
}

template <int IPT>
void launch_ap_bias_pk(const ApLaunch& a, int slices, int j_chunk) {
    dim3 grid((unsigned)((a.i_count + BLOCK * IPT - 1) / (BLOCK * IPT)), (unsigned)slices);
    hipLaunchKernelGGL((k_allpairs_plummer_pk<IPT, true>), grid, dim3(BLOCK), 0, a.stream, a.posm, a.acc_part, a.n,
                       a.i_begin, a.i_count, j_chunk, a.eps2, 0, a.vel);
}

template <int LAW>
void launch_ap_ipt(const ApLaunch& a, int ipt, int slices, int j_chunk) {
    if (LAW == NBODY_LAW_WGSL_BIAS && ipt >= 2 && !env_int("NBODY_AP_NOPACK", 0)) {
        if (ipt == 2) launch_ap_bias_pk<2>(a, slices, j_chunk);
        else launch_ap_bias_pk<4>(a, slices, j_chunk);
        return;
    }
    if (LAW == NBODY_LAW_PLUMMER && ipt >= 2 && !env_int("NBODY_AP_NOPACK", 0)) {
        switch (ipt) {
            case 2: launch_ap_pk<2>(a, slices, j_chunk); return;
            case 8: launch_ap_pk<8>(a, slices, j_chunk); return;
            default: launch_ap_pk<4>(a, slices, j_chunk); return;
        }
    }
    switch (ipt) {
        case 1: launch_ap<LAW, 1>(a, slices, j_chunk); break;
        case 2: launch_ap<LAW, 2>(a, slices, j_chunk); break;
        case 8: launch_ap<LAW, 8>(a, slices, j_chunk); break;
        default: launch_ap<LAW, 4>(a, slices, j_chunk); break;
    }
}

// Targets per lane, source slices and slice length of the plain kernels for i_count targets
// against n sources; at most max_slices slices (what the partial-sum buffer can hold).
struct ApPlan {
    int ipt, slices, j_chunk;
};
// Whether launch_ap_ipt<PLUMMER> takes the packed kernel for this plan (which leaves a uniform mass to ap_finalize).
bool ap_uses_packed_plummer(const ApPlan& pl) { return pl.ipt >= 2 && !env_int("NBODY_AP_NOPACK", 0); }
ApPlan ap_plan(uint64_t n, uint64_t i_count, int num_cus, uint64_t max_slices_fit) {
    ApPlan pl;
    pl.ipt = env_int("NBODY_AP_IPT", 0);
    if (pl.ipt != 1 && pl.ipt != 2 && pl.ipt != 4 && pl.ipt != 8) {
        // four targets per lane measured best down to N = 65,536 (source slices fill the chip);
        // fewer only when there are not even that many targets
        pl.ipt = i_count >= 16384 ? 4 : i_count >= 4096 ? 2 : 1;
    }
    const uint64_t tblocks = (i_count + BLOCK * pl.ipt - 1) / (BLOCK * pl.ipt);
    int slices = env_int("NBODY_AP_SLICES", 0);
    const int max_slices = (int)std::max<uint64_t>(1, (n + TJ - 1) / TJ);
    if (slices <= 0) {
        // aim at >= 16 workgroups per CU in total (measured best at N = 65,536 and 1,048,576)
        uint64_t want = (uint64_t)num_cus * 16;
        slices = (int)std::min<uint64_t>(64, (want + tblocks - 1) / tblocks);
    }
    slices = std::max(1, std::min(slices, max_slices));
    slices = (int)std::min<uint64_t>((uint64_t)slices, std::max<uint64_t>(1, max_slices_fit));
    // slice length: a whole number of tiles
    uint64_t tiles = (n + TJ - 1) / TJ;
    uint64_t tiles_per_slice = (tiles + slices - 1) / slices;
    pl.j_chunk = (int)(tiles_per_slice * TJ);
    pl.slices = (int)((n + pl.j_chunk - 1) / pl.j_chunk);
    return pl;
}

// The force kernel of `law` over the plan's slices; ap_finalize then writes their f64 sums times G.
void ap_forces_plain(const ApLaunch& a, int law, const ApPlan& pl) {
    switch (law) {
        case NBODY_LAW_CPU_R6: launch_ap_ipt<NBODY_LAW_CPU_R6>(a, pl.ipt, pl.slices, pl.j_chunk); break;
        case NBODY_LAW_WGSL_BIAS: launch_ap_ipt<NBODY_LAW_WGSL_BIAS>(a, pl.ipt, pl.slices, pl.j_chunk); break;
        default: launch_ap_ipt<NBODY_LAW_PLUMMER>(a, pl.ipt, pl.slices, pl.j_chunk); break;
    }
}
void ap_finalize(const ApLaunch& a, int law, const ApPlan& pl, float G, float4* acc_out) {
    float g = (law == NBODY_LAW_CPU_R6) ? 0.0066743f : G;
    // the packed PLUMMER kernel left the (common) mass out of its terms
    if (law == NBODY_LAW_PLUMMER && a.uniform_mass > 0.0f && ap_uses_packed_plummer(pl)) g *= a.uniform_mass;
    unsigned fb = (unsigned)((a.i_count + 255) / 256);
    hipLaunchKernelGGL(k_finalize, dim3(fb), dim3(256), 0, a.stream, a.acc_part, pl.slices, a.i_begin, a.i_count, g, acc_out);
}

}  // namespace

// Kernel timing with HIP events on the context's stream.  The events come from a pool that
// nbody_set_kernel_timing(on) fills, so a timed launch creates nothing; the pool only grows (by
// one pair) when more launches than it holds are enqueued between two collections.
int nb_time_reserve(nbody_ctx* ctx, size_t pairs) {
    for (auto& v : ctx->ev) {
        while (v.size() < 2 * pairs) {
            hipEvent_t e = nullptr;
            NB_HIP(ctx, hipEventCreate(&e));
            v.push_back(e);
        }
    }
    return NBODY_OK;
}
void nb_time_begin(nbody_ctx* ctx, int which) {
    if (!ctx->timing) return;
    auto& v = ctx->ev[which];
    size_t& used = ctx->ev_used[which];
    if (used + 2 > v.size()) {
        if (v.size() >= 4096) nb_time_collect(ctx, which);
        else {
            hipEvent_t e0 = nullptr, e1 = nullptr;
            if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return;
            v.push_back(e0);
            v.push_back(e1);
        }
    }
    hipEventRecord(v[used], ctx->stream);
    used += 2;
}
void nb_time_end(nbody_ctx* ctx, int which) {
    if (!ctx->timing) return;
    const size_t used = ctx->ev_used[which];
    if (used >= 2) hipEventRecord(ctx->ev[which][used - 1], ctx->stream);
}
int nb_time_collect(nbody_ctx* ctx, int which) {
    auto& v = ctx->ev[which];
    size_t& used = ctx->ev_used[which];
    if (used == 0) return NBODY_OK;
    NB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (size_t k = 0; k + 1 < used; k += 2) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, v[k], v[k + 1]) == hipSuccess) {
            ctx->time_acc_ms[which] += ms;
            ctx->time_launches[which] += 1;
        }
    }
    used = 0;
    return NBODY_OK;
}

// Un-scaled PLUMMER sums of the context's own targets over the sources of their superblock
// (256 * tpl bodies), one partial sum per NB_DIAG_TILE sources: diag_out[slice * n + body].  Two
// targets per lane and sliced sources: enough workgroups to keep the chip busy with few superblocks.
int nb_allpairs_diag(nbody_ctx* ctx, const float4* posm_in, float4* diag_out, int tpl) {
    constexpr int IPT = 2, WG = BLOCK * IPT;
    const int sb = BLOCK * tpl;
    static_assert(NB_DIAG_TILE == TJ, "one slice per LDS tile");
    const int first = (int)(ctx->i_begin / WG) * WG;  // whole workgroups; the few extra bodies are ignored by the caller
    const int last = (int)std::min<uint64_t>(ctx->n, (ctx->i_begin + ctx->i_count + WG - 1) / WG * WG);
    dim3 grid((unsigned)((last - first + WG - 1) / WG), (unsigned)(sb / TJ));
    hipLaunchKernelGGL((k_allpairs_plummer_pk<IPT>), grid, dim3(BLOCK), 0, ctx->stream, posm_in, diag_out + first, (int)ctx->n,
                       first, last - first, TJ, ctx->cfg.eps2, sb, (const float4*)nullptr, (int)ctx->n);
    NB_HIP(ctx, hipGetLastError());
    return NBODY_OK;
}

// True when the pair kernel's scratch exists already or fits in half of the free device memory.
static bool sym_scratch_fits(nbody_ctx* ctx, uint32_t seg, int tpl, uint64_t* need_out) {
    const uint64_t need = nb_sym_scratch_bytes(ctx->n, seg, 256u * tpl);
    if (need_out) *need_out = need;
    if (ctx->sym != nullptr) return true;
    size_t free_b = 0, total_b = 0;
    return hipMemGetInfo(&free_b, &total_b) == hipSuccess && need < free_b / 2;
}

// Whether this GPU can hold the pair kernel's partial rows (a communicator asks every rank).
bool nb_allpairs_pair_fits(nbody_ctx* ctx) {
    const int tpl = sym_tpl();
    return sym_scratch_fits(ctx, sym_seg_tiles(ctx, tpl), tpl, nullptr);
}

// One part's share of the pair tiles of ALL bodies (PLUMMER): acc_tmp[j] for every j.
int nb_allpairs_partial(nbody_ctx* ctx, uint32_t part, uint32_t nparts) {
    const int tpl = sym_tpl();
    const uint32_t seg = sym_seg_tiles(ctx, tpl);
    uint64_t need = 0;
    if (!sym_scratch_fits(ctx, seg, tpl, &need))
        return nb_fail(ctx, NBODY_ERR_NOMEM, "nbody_allpairs_partial: the pair kernel needs " + std::to_string(need >> 20) +
                                                 " MiB of partial rows, more than half of the free device memory");
    return nb_allpairs_forces_sym(ctx, ctx->posm[ctx->cur], ctx->acc_tmp, seg, tpl, part, nparts);
}

// Forces on this context's targets from all n sources in posm_in; acc_out[i] for i in the range.
int nb_allpairs_forces(nbody_ctx* ctx, const float4* posm_in, const float4* vel_in, float4* acc_out) {
    const int law = ctx->cfg.law;
    // NBODY_AP_SYM: 0 never, 1 whenever there are two superblocks; unset: from N = 65,536 up, where
    // the pair kernel's partial rows and their reduction no longer outweigh what it saves
    const int sym_env = env_int("NBODY_AP_SYM", -1);
    if (law == NBODY_LAW_PLUMMER && ctx->i_count == ctx->n && sym_env != 0 &&
        ctx->n >= (sym_env == 1 ? 4097u : 65536u)) {
        // one evaluation per unordered pair; needs (superblocks x n / 2) x 16 B of partial rows
        const int tpl = sym_tpl();
        const uint32_t seg = sym_seg_tiles(ctx, tpl);
        if (sym_scratch_fits(ctx, seg, tpl, nullptr)) return nb_allpairs_forces_sym(ctx, posm_in, acc_out, seg, tpl, 0, 1);
        // otherwise the plain kernel below; nbody_counters().allpairs_path tells which one ran
    }
    const ApPlan pl = ap_plan(ctx->n, ctx->i_count, ctx->num_cus, ~0ull);
    if ((uint64_t)pl.slices > ctx->acc_part_slices) {
        if (ctx->acc_part) NB_HIP(ctx, hipFree(ctx->acc_part));
        ctx->acc_part = nullptr;
        ctx->acc_part_slices = 0;
        if (hipMalloc(&ctx->acc_part, (size_t)pl.slices * ctx->i_count * sizeof(float4)) != hipSuccess) {
            ctx->acc_part = nullptr;
            (void)hipGetLastError();
            return nb_fail(ctx, NBODY_ERR_NOMEM, "all-pairs partial sums: out of device memory");
        }
        ctx->acc_part_slices = pl.slices;
    }
    const ApLaunch a{ctx->stream, posm_in, vel_in, ctx->acc_part, (int)ctx->n, (int)ctx->i_begin, (int)ctx->i_count, ctx->cfg.eps2,
                     (ctx->uniform_mass && ctx->mass0 > 0.0f) ? ctx->mass0 : 0.0f};
    nb_time_begin(ctx, 0);
    ap_forces_plain(a, law, pl);
    nb_time_end(ctx, 0);
    NB_HIP(ctx, hipGetLastError());
    ap_finalize(a, law, pl, ctx->cfg.G, acc_out);
    NB_HIP(ctx, hipGetLastError());
    ctx->counters.interactions_allpairs +=
        (law == NBODY_LAW_CPU_R6) ? ctx->i_count * (ctx->n - 1) : ctx->i_count * ctx->n;
    ctx->counters.allpairs_path = 1;
    return NBODY_OK;
}

// Advance the state with the accelerations acc_new (indexed by global body) and swap the sets.
// The first leapfrog step after an upload kicks by dt/2: uploaded velocities are on the step.
static void launch_leapfrog(nbody_ctx* ctx, const float4* acc_new, float clamp_world) {
    unsigned fb = (unsigned)((ctx->i_count + 255) / 256);
    const int c = ctx->cur, o = ctx->cur ^ 1;
    const float dt = ctx->cfg.dt;
    hipLaunchKernelGGL(k_integrate_leapfrog, dim3(fb), dim3(256), 0, ctx->stream, ctx->posm[c], ctx->vel[c], acc_new,
                       ctx->posm[o], ctx->vel[o], ctx->acc[o], (int)ctx->i_begin, (int)ctx->i_count, dt,
                       ctx->leapfrog_started ? dt : 0.5f * dt, clamp_world);
    ctx->leapfrog_started = true;
}

// clamp_world > 0: positions are clamped to [0, clamp_world] (src/nbodybh.wgsl:167).
static int integrate(nbody_ctx* ctx, const float4* acc_new, float clamp_world) {
    unsigned fb = (unsigned)((ctx->i_count + 255) / 256);
    const int c = ctx->cur, o = ctx->cur ^ 1;
    if (ctx->cfg.integrator == NBODY_INT_CPU_EULER_XBUF) {
        hipLaunchKernelGGL(k_integrate_cpu_xbuf, dim3(fb), dim3(256), 0, ctx->stream, acc_new, ctx->posm[o], ctx->vel[o],
                           ctx->acc[o], (int)ctx->i_begin, (int)ctx->i_count);
    } else if (ctx->cfg.integrator == NBODY_INT_LEAPFROG) {
        launch_leapfrog(ctx, acc_new, clamp_world);
    } else {
        const float dt = ctx->cfg.dt;
        const float dt2 = std::pow(dt, 2.0f);  // pow(time_step, 2.0), src/nbody.wgsl:132
        hipLaunchKernelGGL(k_integrate_verlet, dim3(fb), dim3(256), 0, ctx->stream, ctx->posm[c], ctx->vel[c],
                           ctx->acc[c], acc_new, ctx->posm[o], ctx->vel[o], ctx->acc[o], (int)ctx->i_begin,
                           (int)ctx->i_count, dt, dt2, clamp_world);
    }
    NB_HIP(ctx, hipGetLastError());
    ctx->cur = o;
    return NBODY_OK;
}

int nb_integrate(nbody_ctx* ctx, const float4* acc_new) { return integrate(ctx, acc_new, 0.0f); }

// The Barnes-Hut step; the caller has refused the cross-buffer integrator, which has no clamp.
int nb_integrate_clamped(nbody_ctx* ctx, const float4* acc_new) { return integrate(ctx, acc_new, ctx->cfg.world); }

int nb_energy(nbody_ctx* ctx, double* kinetic, double* potential) {
    if (!ctx->energy_dev) NB_HIP(ctx, hipMalloc(&ctx->energy_dev, 2 * sizeof(double)));
    NB_HIP(ctx, hipMemsetAsync(ctx->energy_dev, 0, 2 * sizeof(double), ctx->stream));
    unsigned fb = (unsigned)((ctx->n + 255) / 256);
    hipLaunchKernelGGL(k_energy, dim3(fb), dim3(256), 0, ctx->stream, ctx->posm[ctx->cur], ctx->vel[ctx->cur],
                       (int)ctx->n, (double)ctx->cfg.eps2, ctx->energy_dev);
    NB_HIP(ctx, hipGetLastError());
    double h[2];
    NB_HIP(ctx, hipMemcpyAsync(h, ctx->energy_dev, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    NB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (kinetic) *kinetic = h[0];
    if (potential) *potential = -0.5 * (double)ctx->cfg.G * h[1];
    return NBODY_OK;
}

// ---- all-pairs accelerations without any allocation ---------------------------------------------
// For callers that cannot allocate device memory (HIP graph capture, tight serving loops, harnesses
// that forbid it): the same plain kernels as nbody_accel_allpairs, with the per-slice partial sums
// in a fixed device array instead of a buffer grown on demand.  Every GPU the library runs on
// holds its own copy of that array until the process ends.
namespace {
constexpr uint64_t NOALLOC_MAX_BODIES = 131072;
constexpr size_t NOALLOC_SCRATCH_BYTES = 16u << 20;  // 16 MiB: 8 slices at the largest N, 64 up to N = 16,384
constexpr int NOALLOC_MAX_DEVICES = 64;
}  // namespace
static __device__ __align__(256) unsigned char scratch[NOALLOC_SCRATCH_BYTES];
namespace {
// Device address of `scratch` on each GPU (the array's name is only a host-side handle); looked up
// on the first call with that GPU current.  Plain loads and stores of an aligned pointer; two
// threads racing on the first call both store the same value.
void* g_scratch_ptr[NOALLOC_MAX_DEVICES] = {};
int g_scratch_cus[NOALLOC_MAX_DEVICES] = {};
thread_local const char* g_noalloc_error = "";
int noalloc_fail(int code, const char* msg) {
    g_noalloc_error = msg;
    return code;
}
}  // namespace

const char* nb_noalloc_last_error() { return g_noalloc_error; }
void nb_noalloc_clear_error() { g_noalloc_error = ""; }

extern "C" int nbody_accel_allpairs_noalloc(const float* dev_posm, const float* dev_vel16, uint64_t n_bodies, int law, float G,
                                            float eps2, float* dev_acc16_out, void* stream) {
    g_noalloc_error = "";  // the message belongs to the last call made without a context
    if (!dev_posm || !dev_acc16_out) return noalloc_fail(NBODY_ERR_INVALID, "nbody_accel_allpairs_noalloc: null buffer");
    if (law < 0 || law > 2) return noalloc_fail(NBODY_ERR_INVALID, "nbody_accel_allpairs_noalloc: unknown law");
    if (law == NBODY_LAW_WGSL_BIAS && !dev_vel16)
        return noalloc_fail(NBODY_ERR_INVALID, "nbody_accel_allpairs_noalloc: the WGSL_BIAS law needs the velocities");
    if (n_bodies == 0 || n_bodies > NOALLOC_MAX_BODIES)
        return noalloc_fail(NBODY_ERR_CAPACITY, "nbody_accel_allpairs_noalloc: n_bodies must be in [1, 131072]");
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return noalloc_fail(NBODY_ERR_NO_DEVICE, "nbody_accel_allpairs_noalloc: no current HIP device");
    if (dev < 0 || dev >= NOALLOC_MAX_DEVICES)
        return noalloc_fail(NBODY_ERR_NO_DEVICE, "nbody_accel_allpairs_noalloc: device ordinal out of range");
    if (!g_scratch_ptr[dev]) {
        void* ptr = nullptr;
        if (hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)) != hipSuccess || !ptr)
            return noalloc_fail(NBODY_ERR_HIP, "nbody_accel_allpairs_noalloc: hipGetSymbolAddress(scratch) failed");
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        g_scratch_cus[dev] = cus;
        g_scratch_ptr[dev] = ptr;
    }
    float4* part = reinterpret_cast<float4*>(g_scratch_ptr[dev]);
    const ApPlan pl = ap_plan(n_bodies, n_bodies, g_scratch_cus[dev], NOALLOC_SCRATCH_BYTES / (n_bodies * sizeof(float4)));
    if ((uint64_t)pl.slices * n_bodies * sizeof(float4) > NOALLOC_SCRATCH_BYTES)
        return noalloc_fail(NBODY_ERR_CAPACITY, "nbody_accel_allpairs_noalloc: partial sums do not fit the fixed scratch");
    const ApLaunch a{(hipStream_t)stream, reinterpret_cast<const float4*>(dev_posm), reinterpret_cast<const float4*>(dev_vel16),
                     part, (int)n_bodies, 0, (int)n_bodies, eps2, 0.0f};  // (masses unknown here: taken per term)
    // every slice of every target is written by the force kernel before the sum reads it: nothing
    // of the previous call's contents is read, so the array needs no clearing
    (void)hipGetLastError();  // an error left behind by an earlier, unrelated call is not this call's
    ap_forces_plain(a, law, pl);
    if (hipGetLastError() != hipSuccess) return noalloc_fail(NBODY_ERR_HIP, "nbody_accel_allpairs_noalloc: force kernel launch failed");
    ap_finalize(a, law, pl, G, reinterpret_cast<float4*>(dev_acc16_out));
    if (hipGetLastError() != hipSuccess) return noalloc_fail(NBODY_ERR_HIP, "nbody_accel_allpairs_noalloc: sum kernel launch failed");
    return NBODY_OK;
}
