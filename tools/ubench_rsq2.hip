// Micro-benchmark: orderings of the traversal's per-record vector sequence around its one v_rsq_f32
// (no memory accesses; cycles per record per SIMD at 4, 7 and 8 resident wavefronts).
//   D   the sequence as compiled (7 distance ops, rsq, 3 mul, select, 3 fmac)
//   V   D with the record in vector registers instead of scalar ones
//   X   D with the select replaced by running the three accumulations under exec = taking lanes
//   P   software-pipelined: rsq of record k, then the distance part of record k+1, then the rest of k
//   PX  P with the exec form of X
//   T   D with v_rcp_f32 in place of v_rsq_f32 (is it the instruction or the unit?)
//   S   14 v_fma_f32 with a scalar multiplicand, no transcendental
//   N   D with the rsq result written to another register and two s_nop behind it
//   hipcc --offload-arch=gfx950 -O3 -o ubench_rsq2 ubench_rsq2.hip
#include <hip/hip_runtime.h>
#include <cstdio>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// operands: %0..%2 accumulators, %3..%5 target x,y,z (v); %6 cx %7 cy %8 cz %9 mass %10 threshold %11 eps2 (s);
// %12 lane mask (s, 64-bit); %13..%18 the record again in vector registers (cx cy cz mass threshold eps2)
// register sets: A = v11 v12 v13 v14 (dx dy dz r2), compare in s[20:21]; B = v16 v17 v18 v19, s[24:25]; v15 temporary
#define DIST(dx, dy, dz, r2, cm) \
    "v_sub_f32 " dz ", %8, %5\n\tv_sub_f32 " dy ", %7, %4\n\tv_fma_f32 " r2 ", " dz ", " dz ", %11\n\tv_sub_f32 " dx ", %6, %3\n\t" \
    "v_fmac_f32 " r2 ", " dy ", " dy "\n\tv_fmac_f32 " r2 ", " dx ", " dx "\n\tv_cmp_lt_f32 " cm ", %10, " r2 "\n\t"
#define DISTV(dx, dy, dz, r2, cm) \
    "v_sub_f32 " dz ", %15, %5\n\tv_sub_f32 " dy ", %14, %4\n\tv_fma_f32 " r2 ", " dz ", " dz ", %18\n\tv_sub_f32 " dx ", %13, %3\n\t" \
    "v_fmac_f32 " r2 ", " dy ", " dy "\n\tv_fmac_f32 " r2 ", " dx ", " dx "\n\tv_cmp_lt_f32 " cm ", %17, " r2 "\n\t"
#define RSQ(r2) "v_rsq_f32 " r2 ", " r2 "\n\t"
#define RCP(r2) "v_rcp_f32 " r2 ", " r2 "\n\t"
#define TAIL(dx, dy, dz, r2, cm) \
    "v_mul_f32 v15, %9, " r2 "\n\tv_mul_f32 " r2 ", " r2 ", " r2 "\n\tv_mul_f32 " r2 ", v15, " r2 "\n\t" \
    "v_cndmask_b32 " r2 ", 0, " r2 ", " cm "\n\tv_fmac_f32 %0, " r2 ", " dx "\n\tv_fmac_f32 %1, " r2 ", " dy "\n\tv_fmac_f32 %2, " r2 ", " dz "\n\t"
#define TAILV(dx, dy, dz, r2, cm) \
    "v_mul_f32 v15, %16, " r2 "\n\tv_mul_f32 " r2 ", " r2 ", " r2 "\n\tv_mul_f32 " r2 ", v15, " r2 "\n\t" \
    "v_cndmask_b32 " r2 ", 0, " r2 ", " cm "\n\tv_fmac_f32 %0, " r2 ", " dx "\n\tv_fmac_f32 %1, " r2 ", " dy "\n\tv_fmac_f32 %2, " r2 ", " dz "\n\t"
#define TAILX(dx, dy, dz, r2, cm) \
    "v_mul_f32 v15, %9, " r2 "\n\tv_mul_f32 " r2 ", " r2 ", " r2 "\n\tv_mul_f32 " r2 ", v15, " r2 "\n\t" \
    "s_mov_b64 exec, " cm "\n\tv_fmac_f32 %0, " r2 ", " dx "\n\tv_fmac_f32 %1, " r2 ", " dy "\n\tv_fmac_f32 %2, " r2 ", " dz "\n\ts_mov_b64 exec, -1\n\t"
#define SETA "v11", "v12", "v13", "v14", "s[20:21]"
#define SETB "v16", "v17", "v18", "v19", "s[24:25]"
#define AP(M) M("v11", "v12", "v13", "v14", "s[20:21]")
#define BP(M) M("v16", "v17", "v18", "v19", "s[24:25]")
#define REC_D AP(DIST) RSQ("v14") AP(TAIL)
#define REC_V AP(DISTV) RSQ("v14") AP(TAILV)
#define REC_X AP(DIST) RSQ("v14") AP(TAILX)
#define REC_T AP(DIST) RCP("v14") AP(TAIL)
#define REC_N AP(DIST) "v_rsq_f32 v19, v14\n\ts_nop 1\n\t" \
    "v_mul_f32 v15, %9, v19\n\tv_mul_f32 v19, v19, v19\n\tv_mul_f32 v19, v15, v19\n\t" \
    "v_cndmask_b32 v19, 0, v19, s[20:21]\n\tv_fmac_f32 %0, v19, v11\n\tv_fmac_f32 %1, v19, v12\n\tv_fmac_f32 %2, v19, v13\n\t"
// two records per trip of the pipelined forms (set A's distance part is live on entry)
#define PAIR_P RSQ("v14") BP(DIST) AP(TAIL) RSQ("v19") AP(DIST) BP(TAIL)
#define PAIR_PX RSQ("v14") BP(DIST) AP(TAILX) RSQ("v19") AP(DIST) BP(TAILX)
// rsq early in the NEXT record's distance part, its uses at the end of it
#define PAIR_Q RSQ("v14") BP(DIST) RSQ("v19") AP(TAIL) AP(DIST) BP(TAIL)
#define FS "v_fma_f32 %0, %0, %9, %3\n\tv_fma_f32 %1, %1, %9, %4\n\tv_fma_f32 %2, %2, %9, %5\n\tv_fma_f32 v11, v11, %9, %3\n\t"
#define FS2 "v_fma_f32 %0, %0, %9, %3\n\tv_fma_f32 %1, %1, %9, %4\n\t"
#define REC_S FS FS FS FS2

#define TAILM(dx, dy, dz, r2, cm) \
    "v_mul_f32 v15, %9, " r2 "\n\tv_mul_f32 " r2 ", " r2 ", " r2 "\n\tv_mul_f32 " r2 ", v15, " r2 "\n\t" \
    "v_mul_f32 " r2 ", %16, " r2 "\n\tv_fmac_f32 %0, " r2 ", " dx "\n\tv_fmac_f32 %1, " r2 ", " dy "\n\tv_fmac_f32 %2, " r2 ", " dz "\n\t"
#define TAILS(dx, dy, dz, r2, cm) \
    "v_mul_f32 v15, %9, " r2 "\n\tv_mul_f32 " r2 ", " r2 ", " r2 "\n\tv_mul_f32 " r2 ", v15, " r2 "\n\t" \
    "v_cndmask_b32 " r2 ", 0, " r2 ", %12\n\tv_fmac_f32 %0, " r2 ", " dx "\n\tv_fmac_f32 %1, " r2 ", " dy "\n\tv_fmac_f32 %2, " r2 ", " dz "\n\t"
#define DISTNC(dx, dy, dz, r2, cm) \
    "v_sub_f32 " dz ", %8, %5\n\tv_sub_f32 " dy ", %7, %4\n\tv_fma_f32 " r2 ", " dz ", " dz ", %11\n\tv_sub_f32 " dx ", %6, %3\n\t" \
    "v_fmac_f32 " r2 ", " dy ", " dy "\n\tv_fmac_f32 " r2 ", " dx ", " dx "\n\tv_max_f32 v15, %10, " r2 "\n\t"
#define DISTVCC(dx, dy, dz, r2, cm) \
    "v_sub_f32 " dz ", %8, %5\n\tv_sub_f32 " dy ", %7, %4\n\tv_fma_f32 " r2 ", " dz ", " dz ", %11\n\tv_sub_f32 " dx ", %6, %3\n\t" \
    "v_fmac_f32 " r2 ", " dy ", " dy "\n\tv_fmac_f32 " r2 ", " dx ", " dx "\n\tv_cmp_lt_f32 vcc, %10, " r2 "\n\t"
#define TAILVCC(dx, dy, dz, r2, cm) \
    "v_mul_f32 v15, %9, " r2 "\n\tv_mul_f32 " r2 ", " r2 ", " r2 "\n\tv_mul_f32 " r2 ", v15, " r2 "\n\t" \
    "v_cndmask_b32 " r2 ", 0, " r2 ", vcc\n\tv_fmac_f32 %0, " r2 ", " dx "\n\tv_fmac_f32 %1, " r2 ", " dy "\n\tv_fmac_f32 %2, " r2 ", " dz "\n\t"
#define REC_N1 AP(DIST) RSQ("v14") "s_nop 1\n\t" AP(TAIL)
#define REC_N3 AP(DIST) RSQ("v14") "s_nop 0\n\t" AP(TAIL)
#define REC_N2 AP(DIST) "v_rsq_f32 v19, v14\n\t" \
    "v_mul_f32 v15, %9, v19\n\tv_mul_f32 v19, v19, v19\n\tv_mul_f32 v19, v15, v19\n\t" \
    "v_cndmask_b32 v19, 0, v19, s[20:21]\n\tv_fmac_f32 %0, v19, v11\n\tv_fmac_f32 %1, v19, v12\n\tv_fmac_f32 %2, v19, v13\n\t"
#define REC_XN AP(DIST) RSQ("v14") "s_nop 1\n\t" AP(TAILX)
#define REC_C1 AP(DIST) RSQ("v14") AP(TAILM)
#define REC_C2 AP(DISTVCC) RSQ("v14") AP(TAILVCC)
#define REC_C3 AP(DISTNC) RSQ("v14") AP(TAILS)
#define REC_C4 AP(DIST) RSQ("v14") AP(TAILS)
#define REC_E2 AP(DIST) "v_mul_f32 v14, v14, v14\n\t" AP(TAILX)
#define X8(R) R R R R R R R R
#define X4(R) R R R R

template <int KIND>
__global__ __launch_bounds__(64) void k_mix(float* out, int trips, float seed, float cx, float cy, float cz, float m, float thr,
                                            float eps2) {
    float ax = 0, ay = 0, az = 0;
    const float tx = seed + threadIdx.x, ty = seed * 2 + threadIdx.x, tz = seed * 3;
    const float vcx = cx + seed, vcy = cy + seed, vcz = cz + seed, vm = m * seed, vthr = thr * seed, veps = eps2 * seed;
    const unsigned long long mask = __ballot(threadIdx.x < 40);
    for (int t = 0; t < trips; ++t) {
#define OPS : "+v"(ax), "+v"(ay), "+v"(az) : "v"(tx), "v"(ty), "v"(tz), "s"(cx), "s"(cy), "s"(cz), "s"(m), "s"(thr), "s"(eps2), "s"(mask), \
              "v"(vcx), "v"(vcy), "v"(vcz), "v"(vm), "v"(vthr), "v"(veps) \
            : "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "s20", "s21", "s22", "s23", "s24", "s25", "scc", "vcc"
        if (KIND == 0) asm volatile(X8(REC_D) OPS);
        if (KIND == 1) asm volatile(X8(REC_V) OPS);
        if (KIND == 2) asm volatile(X8(REC_X) OPS);
        if (KIND == 3) asm volatile(X4(PAIR_P) OPS);
        if (KIND == 4) asm volatile(X4(PAIR_PX) OPS);
        if (KIND == 5) asm volatile(X8(REC_T) OPS);
        if (KIND == 6) asm volatile(X8(REC_S) OPS);
        if (KIND == 7) asm volatile(X8(REC_N) OPS);
        if (KIND == 8) asm volatile(X4(PAIR_Q) OPS);
        if (KIND == 9) asm volatile(X8(REC_N1) OPS);
        if (KIND == 10) asm volatile(X8(REC_N2) OPS);
        if (KIND == 11) asm volatile(X8(REC_N3) OPS);
        if (KIND == 12) asm volatile(X8(REC_XN) OPS);
        if (KIND == 13) asm volatile(X8(REC_C1) OPS);
        if (KIND == 14) asm volatile(X8(REC_C2) OPS);
        if (KIND == 15) asm volatile(X8(REC_C3) OPS);
        if (KIND == 16) asm volatile(X8(REC_C4) OPS);
        if (KIND == 17) asm volatile(X8(REC_E2) OPS);
    }
    if (ax + ay + az == 123.456f) out[0] = ax;
}

template <int KIND>
int run(const char* name) {
    float* out;
    CHECK(hipMalloc(&out, 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int simds = prop.multiProcessorCount * 4;
    for (int wps : {1, 4, 7, 8}) {
        const int blocks = simds * wps, trips = 40000 / wps;
        const size_t lds = (size_t)(160 * 1024) / (4 * wps) - 512;
        k_mix<KIND><<<blocks, 64, lds>>>(out, 10, 1.0f, 3.f, 4.f, 5.f, 1.f, 0.5f, 1.f);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        k_mix<KIND><<<blocks, 64, lds>>>(out, trips, 1.0f, 3.f, 4.f, 5.f, 1.f, 0.5f, 1.f);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double cyc = ms * 1e-3 * 2.4e9 / ((double)wps * trips * 8);
        printf("%-58s waves/SIMD=%d  %8.3f ms  %6.1f cycles per record per SIMD (at 2.4 GHz)\n", name, wps, ms, cyc);
    }
    CHECK(hipFree(out));
    return 0;
}

int main() {
    run<0>("D  as compiled");
    run<1>("V  record in vector registers");
    run<2>("X  accumulate under exec (14 VALU + 2 SALU)");
    run<3>("P  pipelined: rsq k | distance k+1 | rest of k");
    run<4>("PX pipelined + accumulate under exec");
    run<5>("T  v_rcp_f32 instead of v_rsq_f32");
    run<6>("S  14 v_fma_f32 with a scalar operand, no transcendental");
    run<7>("N  rsq to another register, s_nop 1 behind it");
    run<8>("Q  rsq k | distance k+1 | rsq k+1 | rest of k | ...");
    run<9>("N1 D + s_nop 1 behind the rsq");
    run<10>("N2 D with the rsq result in another register");
    run<11>("N3 D + s_nop 0 behind the rsq");
    run<12>("XN X + s_nop 1 behind the rsq");
    run<13>("C1 select replaced by v_mul with a vector register");
    run<14>("C2 compare into vcc, select on vcc");
    run<15>("C3 no compare (v_max instead), select on a constant mask");
    run<16>("C4 compare kept, select on a constant scalar mask");
    run<17>("E2 X with v_mul in place of the rsq");
    return 0;
}
