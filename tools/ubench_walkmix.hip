// Micro-benchmark: the traversal's per-record instruction sequence without its memory accesses, to
// see what the instruction mix alone costs per record on one SIMD at 4..8 resident wavefronts.
//   A: the sequence the compiler emits for the common case (all lanes of the group accept)
//   B: the same vector work with ONE scalar instruction and one untaken branch per record
//   C: B without the force term (separation, r^2 and compare only)
//   D: the vector work alone
//   E: D with the v_rsq_f32 replaced by a v_mul_f32 (what the one transcendental costs in this mix)
//   F, G: D for four / eight records at a time, pass by pass (independent chains interleaved, the
//         transcendentals back to back)
//   H..L: two records per packed instruction (record fields of a pair in consecutive scalar registers)
//   hipcc --offload-arch=gfx950 -O3 -o ubench_walkmix ubench_walkmix.hip
#include <hip/hip_runtime.h>
#include <cstdio>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// operands: %0..%2 accumulators, %3..%5 target x,y,z (v); %6 cx %7 cy %8 cz %9 mass %10 threshold %11 eps2 (s);
// %12 lane mask (s, 64-bit)
#define REC_V1 \
    "v_sub_f32 v13, %8, %5\n\tv_sub_f32 v12, %7, %4\n\tv_fma_f32 v14, v13, v13, %11\n\tv_sub_f32 v11, %6, %3\n\t" \
    "v_fmac_f32 v14, v12, v12\n\tv_fmac_f32 v14, v11, v11\n\tv_cmp_lt_f32 s[20:21], %10, v14\n\t"
#define REC_V2 \
    "v_rsq_f32 v14, v14\n\tv_mul_f32 v15, %9, v14\n\tv_mul_f32 v14, v14, v14\n\tv_mul_f32 v14, v15, v14\n\t" \
    "v_cndmask_b32 v14, 0, v14, s[22:23]\n\tv_fmac_f32 %0, v14, v11\n\tv_fmac_f32 %1, v14, v12\n\tv_fmac_f32 %2, v14, v13\n\t"
#define REC_A REC_V1 "s_and_b64 s[22:23], s[20:21], %12\n\ts_cmp_lg_u64 s[22:23], %12\n\ts_cbranch_scc0 1f\n\ts_nop 0\n\t1:\n\ts_mov_b64 s[22:23], %12\n\t" \
    "s_add_u32 s24, s24, 1\n\ts_cmp_lt_u32 s24, 2\n\t" REC_V2 "s_cbranch_scc1 2f\n\t2:\n\t"
#define REC_B REC_V1 "s_xor_b64 s[22:23], s[20:21], %12\n\ts_cbranch_scc0 1f\n\t1:\n\t" REC_V2
#define REC_C REC_V1 "s_xor_b64 s[22:23], s[20:21], %12\n\ts_cbranch_scc0 1f\n\t1:\n\t"
#define REC_D REC_V1 REC_V2
#define REC_E REC_V1 \
    "v_mul_f32 v14, v14, v14\n\tv_mul_f32 v15, %9, v14\n\tv_mul_f32 v14, v14, v14\n\tv_mul_f32 v14, v15, v14\n\t" \
    "v_cndmask_b32 v14, 0, v14, s[20:21]\n\tv_fmac_f32 %0, v14, v11\n\tv_fmac_f32 %1, v14, v12\n\tv_fmac_f32 %2, v14, v13\n\t"
#define PH4 \
    "v_sub_f32 v22, %8, %5\n\tv_sub_f32 v28, %8, %5\n\tv_sub_f32 v34, %8, %5\n\tv_sub_f32 v40, %8, %5\n\t" \
    "v_sub_f32 v21, %7, %4\n\tv_sub_f32 v27, %7, %4\n\tv_sub_f32 v33, %7, %4\n\tv_sub_f32 v39, %7, %4\n\t" \
    "v_sub_f32 v20, %6, %3\n\tv_sub_f32 v26, %6, %3\n\tv_sub_f32 v32, %6, %3\n\tv_sub_f32 v38, %6, %3\n\t" \
    "v_fma_f32 v23, v22, v22, %11\n\tv_fma_f32 v29, v28, v28, %11\n\tv_fma_f32 v35, v34, v34, %11\n\tv_fma_f32 v41, v40, v40, %11\n\t" \
    "v_fmac_f32 v23, v21, v21\n\tv_fmac_f32 v29, v27, v27\n\tv_fmac_f32 v35, v33, v33\n\tv_fmac_f32 v41, v39, v39\n\t" \
    "v_fmac_f32 v23, v20, v20\n\tv_fmac_f32 v29, v26, v26\n\tv_fmac_f32 v35, v32, v32\n\tv_fmac_f32 v41, v38, v38\n\t" \
    "v_cmp_lt_f32 s[26:27], %10, v23\n\tv_cmp_lt_f32 s[28:29], %10, v29\n\tv_cmp_lt_f32 s[30:31], %10, v35\n\tv_cmp_lt_f32 s[32:33], %10, v41\n\t" \
    "v_rsq_f32 v23, v23\n\tv_rsq_f32 v29, v29\n\tv_rsq_f32 v35, v35\n\tv_rsq_f32 v41, v41\n\t" \
    "v_mul_f32 v24, %9, v23\n\tv_mul_f32 v30, %9, v29\n\tv_mul_f32 v36, %9, v35\n\tv_mul_f32 v42, %9, v41\n\t" \
    "v_mul_f32 v23, v23, v23\n\tv_mul_f32 v29, v29, v29\n\tv_mul_f32 v35, v35, v35\n\tv_mul_f32 v41, v41, v41\n\t" \
    "v_mul_f32 v23, v24, v23\n\tv_mul_f32 v29, v30, v29\n\tv_mul_f32 v35, v36, v35\n\tv_mul_f32 v41, v42, v41\n\t" \
    "v_cndmask_b32 v23, 0, v23, s[26:27]\n\tv_cndmask_b32 v29, 0, v29, s[28:29]\n\tv_cndmask_b32 v35, 0, v35, s[30:31]\n\tv_cndmask_b32 v41, 0, v41, s[32:33]\n\t" \
    "v_fmac_f32 %0, v23, v20\n\tv_fmac_f32 %1, v23, v21\n\tv_fmac_f32 %2, v23, v22\n\tv_fmac_f32 %0, v29, v26\n\t" \
    "v_fmac_f32 %1, v29, v27\n\tv_fmac_f32 %2, v29, v28\n\tv_fmac_f32 %0, v35, v32\n\tv_fmac_f32 %1, v35, v33\n\t" \
    "v_fmac_f32 %2, v35, v34\n\tv_fmac_f32 %0, v41, v38\n\tv_fmac_f32 %1, v41, v39\n\tv_fmac_f32 %2, v41, v40\n\t"
#define PH8 \
    "v_sub_f32 v22, %8, %5\n\tv_sub_f32 v28, %8, %5\n\tv_sub_f32 v34, %8, %5\n\tv_sub_f32 v40, %8, %5\n\t" \
    "v_sub_f32 v46, %8, %5\n\tv_sub_f32 v52, %8, %5\n\tv_sub_f32 v58, %8, %5\n\tv_sub_f32 v64, %8, %5\n\t" \
    "v_sub_f32 v21, %7, %4\n\tv_sub_f32 v27, %7, %4\n\tv_sub_f32 v33, %7, %4\n\tv_sub_f32 v39, %7, %4\n\t" \
    "v_sub_f32 v45, %7, %4\n\tv_sub_f32 v51, %7, %4\n\tv_sub_f32 v57, %7, %4\n\tv_sub_f32 v63, %7, %4\n\t" \
    "v_sub_f32 v20, %6, %3\n\tv_sub_f32 v26, %6, %3\n\tv_sub_f32 v32, %6, %3\n\tv_sub_f32 v38, %6, %3\n\t" \
    "v_sub_f32 v44, %6, %3\n\tv_sub_f32 v50, %6, %3\n\tv_sub_f32 v56, %6, %3\n\tv_sub_f32 v62, %6, %3\n\t" \
    "v_fma_f32 v23, v22, v22, %11\n\tv_fma_f32 v29, v28, v28, %11\n\tv_fma_f32 v35, v34, v34, %11\n\tv_fma_f32 v41, v40, v40, %11\n\t" \
    "v_fma_f32 v47, v46, v46, %11\n\tv_fma_f32 v53, v52, v52, %11\n\tv_fma_f32 v59, v58, v58, %11\n\tv_fma_f32 v65, v64, v64, %11\n\t" \
    "v_fmac_f32 v23, v21, v21\n\tv_fmac_f32 v29, v27, v27\n\tv_fmac_f32 v35, v33, v33\n\tv_fmac_f32 v41, v39, v39\n\t" \
    "v_fmac_f32 v47, v45, v45\n\tv_fmac_f32 v53, v51, v51\n\tv_fmac_f32 v59, v57, v57\n\tv_fmac_f32 v65, v63, v63\n\t" \
    "v_fmac_f32 v23, v20, v20\n\tv_fmac_f32 v29, v26, v26\n\tv_fmac_f32 v35, v32, v32\n\tv_fmac_f32 v41, v38, v38\n\t" \
    "v_fmac_f32 v47, v44, v44\n\tv_fmac_f32 v53, v50, v50\n\tv_fmac_f32 v59, v56, v56\n\tv_fmac_f32 v65, v62, v62\n\t" \
    "v_cmp_lt_f32 s[26:27], %10, v23\n\tv_cmp_lt_f32 s[28:29], %10, v29\n\tv_cmp_lt_f32 s[30:31], %10, v35\n\tv_cmp_lt_f32 s[32:33], %10, v41\n\t" \
    "v_cmp_lt_f32 s[34:35], %10, v47\n\tv_cmp_lt_f32 s[36:37], %10, v53\n\tv_cmp_lt_f32 s[38:39], %10, v59\n\tv_cmp_lt_f32 s[40:41], %10, v65\n\t" \
    "v_rsq_f32 v23, v23\n\tv_rsq_f32 v29, v29\n\tv_rsq_f32 v35, v35\n\tv_rsq_f32 v41, v41\n\t" \
    "v_rsq_f32 v47, v47\n\tv_rsq_f32 v53, v53\n\tv_rsq_f32 v59, v59\n\tv_rsq_f32 v65, v65\n\t" \
    "v_mul_f32 v24, %9, v23\n\tv_mul_f32 v30, %9, v29\n\tv_mul_f32 v36, %9, v35\n\tv_mul_f32 v42, %9, v41\n\t" \
    "v_mul_f32 v48, %9, v47\n\tv_mul_f32 v54, %9, v53\n\tv_mul_f32 v60, %9, v59\n\tv_mul_f32 v66, %9, v65\n\t" \
    "v_mul_f32 v23, v23, v23\n\tv_mul_f32 v29, v29, v29\n\tv_mul_f32 v35, v35, v35\n\tv_mul_f32 v41, v41, v41\n\t" \
    "v_mul_f32 v47, v47, v47\n\tv_mul_f32 v53, v53, v53\n\tv_mul_f32 v59, v59, v59\n\tv_mul_f32 v65, v65, v65\n\t" \
    "v_mul_f32 v23, v24, v23\n\tv_mul_f32 v29, v30, v29\n\tv_mul_f32 v35, v36, v35\n\tv_mul_f32 v41, v42, v41\n\t" \
    "v_mul_f32 v47, v48, v47\n\tv_mul_f32 v53, v54, v53\n\tv_mul_f32 v59, v60, v59\n\tv_mul_f32 v65, v66, v65\n\t" \
    "v_cndmask_b32 v23, 0, v23, s[26:27]\n\tv_cndmask_b32 v29, 0, v29, s[28:29]\n\tv_cndmask_b32 v35, 0, v35, s[30:31]\n\tv_cndmask_b32 v41, 0, v41, s[32:33]\n\t" \
    "v_cndmask_b32 v47, 0, v47, s[34:35]\n\tv_cndmask_b32 v53, 0, v53, s[36:37]\n\tv_cndmask_b32 v59, 0, v59, s[38:39]\n\tv_cndmask_b32 v65, 0, v65, s[40:41]\n\t" \
    "v_fmac_f32 %0, v23, v20\n\tv_fmac_f32 %1, v23, v21\n\tv_fmac_f32 %2, v23, v22\n\tv_fmac_f32 %0, v29, v26\n\t" \
    "v_fmac_f32 %1, v29, v27\n\tv_fmac_f32 %2, v29, v28\n\tv_fmac_f32 %0, v35, v32\n\tv_fmac_f32 %1, v35, v33\n\t" \
    "v_fmac_f32 %2, v35, v34\n\tv_fmac_f32 %0, v41, v38\n\tv_fmac_f32 %1, v41, v39\n\tv_fmac_f32 %2, v41, v40\n\t" \
    "v_fmac_f32 %0, v47, v44\n\tv_fmac_f32 %1, v47, v45\n\tv_fmac_f32 %2, v47, v46\n\tv_fmac_f32 %0, v53, v50\n\t" \
    "v_fmac_f32 %1, v53, v51\n\tv_fmac_f32 %2, v53, v52\n\tv_fmac_f32 %0, v59, v56\n\tv_fmac_f32 %1, v59, v57\n\t" \
    "v_fmac_f32 %2, v59, v58\n\tv_fmac_f32 %0, v65, v62\n\tv_fmac_f32 %1, v65, v63\n\tv_fmac_f32 %2, v65, v64\n\t"
#define X8(R) R R R R R R R R

template <int KIND>
__global__ __launch_bounds__(64) void k_mix(float* out, int trips, float seed, float cx, float cy, float cz, float m, float thr,
                                            float eps2) {
    float ax = 0, ay = 0, az = 0;
    const float tx = seed + threadIdx.x, ty = seed * 2 + threadIdx.x, tz = seed * 3;
    const unsigned long long mask = __ballot(threadIdx.x < 40);
    for (int t = 0; t < trips; ++t) {
#define OPS : "+v"(ax), "+v"(ay), "+v"(az) : "v"(tx), "v"(ty), "v"(tz), "s"(cx), "s"(cy), "s"(cz), "s"(m), "s"(thr), "s"(eps2), "s"(mask) \
            : "v11", "v12", "v13", "v14", "v15", "s20", "s21", "s22", "s23", "s24", "scc", \
              "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "s26", "s27", "s28", "s29", "s30", "s31", "s32", "s33", "s34", "s35", "s36", "s37", "s38", "s39", "s40", "s41"
        if (KIND == 0) asm volatile(X8(REC_A) OPS);
        if (KIND == 1) asm volatile(X8(REC_B) OPS);
        if (KIND == 2) asm volatile(X8(REC_C) OPS);
        if (KIND == 3) asm volatile(X8(REC_D) OPS);
        if (KIND == 4) asm volatile(X8(REC_E) OPS);
        if (KIND == 5) asm volatile(PH4 PH4 OPS);
        if (KIND == 6) asm volatile(PH8 OPS);
    }
    if (ax + ay + az == 123.456f) out[0] = ax;
}


// Two records per instruction: operands %0..%2 packed accumulators, %3 {tx,ty}, %4 {tz,eps2} (v pairs);
// %5 cx pair, %6 cy pair, %7 cz pair, %8 mass pair (s pairs), %9 %10 thresholds (s)
#define PAIR_V1(a) \
    "v_pk_add_f32 v[" #a "0:" #a "1], %5, %3 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t" \
    "v_pk_add_f32 v[" #a "2:" #a "3], %6, %3 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]\n\t" \
    "v_pk_add_f32 v[" #a "4:" #a "5], %7, %4 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t" \
    "v_pk_fma_f32 v[" #a "6:" #a "7], v[" #a "4:" #a "5], v[" #a "4:" #a "5], %4 op_sel:[0,0,1] op_sel_hi:[1,1,1]\n\t" \
    "v_pk_fma_f32 v[" #a "6:" #a "7], v[" #a "2:" #a "3], v[" #a "2:" #a "3], v[" #a "6:" #a "7]\n\t" \
    "v_pk_fma_f32 v[" #a "6:" #a "7], v[" #a "0:" #a "1], v[" #a "0:" #a "1], v[" #a "6:" #a "7]\n\t"
#define PAIR_CMP(a, sa, sb) \
    "v_cmp_lt_f32 " sa ", %9, v" #a "6\n\tv_cmp_lt_f32 " sb ", %10, v" #a "7\n\t"
#define PAIR_RSQ(a) "v_rsq_f32 v" #a "6, v" #a "6\n\tv_rsq_f32 v" #a "7, v" #a "7\n\t"
#define PAIR_SC(a) \
    "v_pk_mul_f32 v[" #a "8:" #a "9], %8, v[" #a "6:" #a "7]\n\t" \
    "v_pk_mul_f32 v[" #a "6:" #a "7], v[" #a "6:" #a "7], v[" #a "6:" #a "7]\n\t" \
    "v_pk_mul_f32 v[" #a "6:" #a "7], v[" #a "8:" #a "9], v[" #a "6:" #a "7]\n\t"
#define PAIR_SEL(a, sa, sb) "v_cndmask_b32 v" #a "6, 0, v" #a "6, " sa "\n\tv_cndmask_b32 v" #a "7, 0, v" #a "7, " sb "\n\t"
#define PAIR_ACC(a) \
    "v_pk_fma_f32 %0, v[" #a "6:" #a "7], v[" #a "0:" #a "1], %0\n\t" \
    "v_pk_fma_f32 %1, v[" #a "6:" #a "7], v[" #a "2:" #a "3], %1\n\t" \
    "v_pk_fma_f32 %2, v[" #a "6:" #a "7], v[" #a "4:" #a "5], %2\n\t"
#define PAIR_H PAIR_V1(2) PAIR_CMP(2, "s[20:21]", "s[22:23]") PAIR_RSQ(2) PAIR_SC(2) PAIR_ACC(2)
#define PAIR_I PAIR_V1(2) PAIR_CMP(2, "s[20:21]", "s[22:23]") PAIR_RSQ(2) PAIR_SC(2) PAIR_SEL(2, "s[20:21]", "s[22:23]") PAIR_ACC(2)
#define PAIR_J PAIR_V1(2) PAIR_V1(3) PAIR_CMP(2, "s[20:21]", "s[22:23]") PAIR_CMP(3, "s[26:27]", "s[28:29]") PAIR_RSQ(2) PAIR_RSQ(3) \
    PAIR_SC(2) PAIR_SC(3) PAIR_ACC(2) PAIR_ACC(3)
#define PAIR_K PAIR_V1(2) PAIR_V1(3) PAIR_V1(4) PAIR_V1(5) PAIR_CMP(2, "s[20:21]", "s[22:23]") PAIR_CMP(3, "s[26:27]", "s[28:29]") \
    PAIR_CMP(4, "s[30:31]", "s[32:33]") PAIR_CMP(5, "s[34:35]", "s[36:37]") PAIR_RSQ(2) PAIR_RSQ(3) PAIR_RSQ(4) PAIR_RSQ(5) \
    PAIR_SC(2) PAIR_SC(3) PAIR_SC(4) PAIR_SC(5) PAIR_ACC(2) PAIR_ACC(3) PAIR_ACC(4) PAIR_ACC(5)
// H without the transcendental (what the two v_rsq_f32 cost in the packed mix)
#define PAIR_L PAIR_V1(2) PAIR_CMP(2, "s[20:21]", "s[22:23]") "v_mul_f32 v26, v26, v26\n\tv_mul_f32 v27, v27, v27\n\t" PAIR_SC(2) PAIR_ACC(2)
typedef float f2 __attribute__((ext_vector_type(2)));
template <int KIND>
__global__ __launch_bounds__(64) void k_pair(float* out, int trips, float seed, f2 cx, f2 cy, f2 cz, f2 m, float thr0, float thr1, float eps2) {
    f2 ax = {0, 0}, ay = {0, 0}, az = {0, 0};
    const f2 txy = {seed + threadIdx.x, seed * 2 + threadIdx.x}, tze = {seed * 3, eps2};
    for (int t = 0; t < trips; ++t) {
#define POPS : "+v"(ax), "+v"(ay), "+v"(az) : "v"(txy), "v"(tze), "s"(cx), "s"(cy), "s"(cz), "s"(m), "s"(thr0), "s"(thr1) \
            : "s20", "s21", "s22", "s23", "s26", "s27", "s28", "s29", "s30", "s31", "s32", "s33", "s34", "s35", "s36", "s37", \
              "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", \
              "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59"
        if (KIND == 0) asm volatile(PAIR_H PAIR_H PAIR_H PAIR_H POPS);
        if (KIND == 1) asm volatile(PAIR_I PAIR_I PAIR_I PAIR_I POPS);
        if (KIND == 2) asm volatile(PAIR_J PAIR_J POPS);
        if (KIND == 3) asm volatile(PAIR_K POPS);
        if (KIND == 4) asm volatile(PAIR_L PAIR_L PAIR_L PAIR_L POPS);
    }
    if (ax.x + ay.x + az.x + ax.y + ay.y + az.y == 123.456f) out[0] = ax.x;
}
template <int KIND>
int run_pair(const char* name) {
    float* out;
    CHECK(hipMalloc(&out, 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int simds = prop.multiProcessorCount * 4;
    const f2 cx = {3.f, 3.5f}, cy = {4.f, 4.5f}, cz = {5.f, 5.5f}, m = {1.f, 2.f};
    for (int wps : {1, 2, 4, 6, 7, 8}) {
        const int blocks = simds * wps, trips = 40000 / wps;
        const size_t lds = (size_t)(160 * 1024) / (4 * wps) - 512;
        k_pair<KIND><<<blocks, 64, lds>>>(out, 10, 1.0f, cx, cy, cz, m, 0.5f, 0.6f, 1.f);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        k_pair<KIND><<<blocks, 64, lds>>>(out, trips, 1.0f, cx, cy, cz, m, 0.5f, 0.6f, 1.f);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double cyc = ms * 1e-3 * 2.4e9 / ((double)wps * trips * 8);
        printf("%-52s waves/SIMD=%d  %8.3f ms  %6.1f cycles per record per SIMD (at 2.4 GHz)\n", name, wps, ms, cyc);
    }
    CHECK(hipFree(out));
    return 0;
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
    for (int wps : {1, 2, 4, 6, 7, 8}) {
        // one-wavefront workgroups, wps per SIMD; dynamic LDS keeps more from becoming resident
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
        printf("%-52s waves/SIMD=%d  %8.3f ms  %6.1f cycles per record per SIMD (at 2.4 GHz)\n", name, wps, ms, cyc);
    }
    CHECK(hipFree(out));
    return 0;
}

int main() {
    run<0>("A compiler's sequence (15 VALU, 5 SALU, 2 branches)");
    run<1>("B 15 VALU, 1 SALU, 1 untaken branch");
    run<2>("C 7 VALU, 1 SALU, 1 untaken branch (no force term)");
    run<3>("D 15 VALU");
    run<4>("E 15 VALU, v_rsq_f32 replaced by v_mul_f32");
    run<5>("F 15 VALU, four records pass by pass");
    run<6>("G 15 VALU, eight records pass by pass");
    run_pair<0>("H two records packed (12 pk, 2 cmp, 2 rsq)");
    run_pair<1>("I H with two selects");
    run_pair<2>("J H for two pairs pass by pass");
    run_pair<3>("K H for four pairs pass by pass");
    run_pair<4>("L H with v_rsq_f32 replaced by v_mul_f32");
    return 0;
}
