// Force laws of the reference as device functions shared by the all-pairs and Barnes-Hut kernels.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/nbody_hip.h"

struct Tgt {
    float x, y, z;
    float vx, vy, vz, vn;  // WGSL_BIAS only: target velocity and its norm
    int idx;               // global body index (CPU_R6 self test)
};

// acos(q) for q in [-1,1]: sqrt(1-|q|) * P7(|q|), reflected for q < 0 (Abramowitz & Stegun
// 4.4.46, |error| <= 2e-8 before f32 rounding).
__device__ __forceinline__ float fast_acos(float q) {
    float a = fminf(fabsf(q), 1.0f);
    float p = -0.0012624911f;
    p = fmaf(p, a, 0.0066700901f);
    p = fmaf(p, a, -0.0170881256f);
    p = fmaf(p, a, 0.0308918810f);
    p = fmaf(p, a, -0.0501743046f);
    p = fmaf(p, a, 0.0889789874f);
    p = fmaf(p, a, -0.2145988016f);
    p = fmaf(p, a, 1.5707963050f);
    float r = __builtin_amdgcn_sqrtf(1.0f - a) * p;
    return q < 0.0f ? 3.14159265358979f - r : r;
}

// One source term added to (ax,ay,az).  G is applied once per target afterwards.
// SELF: compare the source's global index with the target's (CPU_R6 skips n2 == n,
// src/nbody_cpu.rs:335).
// v_rsq_f32 is good to ~1 ulp; one Newton step brings x^-1/2 to f32 rounding.  Used where a
// body's sum is short enough (tree walks) that per-term error does not average out.
__device__ __forceinline__ float rsq_refined(float x) {
    float y = __builtin_amdgcn_rsqf(x);
    float h = 0.5f * y;
    float e = fmaf(-x * y, h, 0.5f);  // 0.5 - 0.5*x*y*y
    return fmaf(y, e, y);
}

// The factor sc of one source term sc * d for the WGSL laws, from the separation
// d = source - target and its squared length r2u (no softening in it).
template <int LAW, bool PRECISE>
__device__ __forceinline__ float pair_scale(const Tgt& t, float dx, float dy, float dz, float r2u, float mass, float eps2) {
    static_assert(LAW != NBODY_LAW_CPU_R6, "WGSL laws only");
    const float rinv = PRECISE ? rsq_refined(r2u + eps2) : __builtin_amdgcn_rsqf(r2u + eps2);
    float sc = (mass * rinv) * (rinv * rinv);  // src/nbody.wgsl:56-65
    if (LAW == NBODY_LAW_WGSL_BIAS) {          // src/nbody.wgsl:74-79
        const float dist = __builtin_amdgcn_sqrtf(r2u);
        const float vd = fmaf(t.vx, dx, fmaf(t.vy, dy, t.vz * dz));
        const float q = vd * __builtin_amdgcn_rcpf(fmaf(dist, t.vn, 1.0f));
        sc *= __builtin_amdgcn_exp2f(0.15f * __builtin_amdgcn_logf(fast_acos(q)));
    }
    return sc;
}

template <int LAW, bool SELF, bool PRECISE = false>
__device__ __forceinline__ void interact_d(const Tgt& t, float dx, float dy, float dz, float mass, int jglob, float eps2,
                                           float& ax, float& ay, float& az);

template <int LAW, bool SELF, bool PRECISE = false>
__device__ __forceinline__ void interact(const Tgt& t, const float4 s, int jglob, float eps2, float& ax, float& ay,
                                         float& az) {
    interact_d<LAW, SELF, PRECISE>(t, s.x - t.x, s.y - t.y, s.z - t.z, s.w, jglob, eps2, ax, ay, az);
}

// The same with the separation d = source - target already formed.
template <int LAW, bool SELF, bool PRECISE>
__device__ __forceinline__ void interact_d(const Tgt& t, float dx, float dy, float dz, float mass, int jglob, float eps2,
                                           float& ax, float& ay, float& az) {
    const struct { float w; } s = {mass};
    float sc;
    if (LAW == NBODY_LAW_CPU_R6) {
        // src/nbody_cpu.rs:337-340: m*d / (d.d)^3
        float r2 = fmaf(dx, dx, fmaf(dy, dy, dz * dz));
        float r6 = r2 * r2 * r2;
        sc = s.w * __builtin_amdgcn_rcpf(r6);
        if (SELF) sc = (jglob == t.idx) ? 0.0f : sc;
    } else if (LAW == NBODY_LAW_PLUMMER) {
        // src/nbody.wgsl:56-65 without :74-79
        float r2 = fmaf(dx, dx, fmaf(dy, dy, fmaf(dz, dz, eps2)));
        float rinv = PRECISE ? rsq_refined(r2) : __builtin_amdgcn_rsqf(r2);
        sc = (s.w * rinv) * (rinv * rinv);
    } else {
        // src/nbody.wgsl:56-79
        float r2u = fmaf(dx, dx, fmaf(dy, dy, dz * dz));
        float dist = __builtin_amdgcn_sqrtf(r2u);
        float rinv = PRECISE ? rsq_refined(r2u + eps2) : __builtin_amdgcn_rsqf(r2u + eps2);
        sc = (s.w * rinv) * (rinv * rinv);
        float vd = fmaf(t.vx, dx, fmaf(t.vy, dy, t.vz * dz));
        float q = vd * __builtin_amdgcn_rcpf(fmaf(dist, t.vn, 1.0f));
        float b = fast_acos(q);
        // pow(b, 0.15) = exp2(0.15*log2(b)); b == 0 gives exp2(-inf) = 0
        sc *= __builtin_amdgcn_exp2f(0.15f * __builtin_amdgcn_logf(b));
    }
    ax = fmaf(sc, dx, ax);
    ay = fmaf(sc, dy, ay);
    az = fmaf(sc, dz, az);
}

