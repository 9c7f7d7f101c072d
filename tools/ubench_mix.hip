// Micro-benchmark 2: shader clock under load, VALU+transcendental mixes, and whether f32 MFMA
// instructions co-execute with VALU work of the same wave on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f2 __attribute__((ext_vector_type(2)));

constexpr int U = 12;

// KIND: 0: 12 fma + 1 rsq (spread)   1: 24 fma + 2 rsq adjacent   2: 12 mul + 1 rsq
//       3: 12 fma only               4: 12 fma + 1 mfma4x4x1      5: 8 fma + 1 mfma4x4x1
//       6: mfma4x4x1 only            7: 12 fma + 1 mfma16x16x4 per 4 groups   8: 6 pk_fma + 1 rsq ... etc
template <int KIND>
__global__ __launch_bounds__(256) void k_mix(float* out, unsigned long long* clk, int trips, float seed) {
    float a[U], b[U];
    f2 p[U];
    for (int i = 0; i < U; ++i) { a[i] = seed + i + threadIdx.x * 1e-3f; b[i] = 1.0f + i * 1e-6f; p[i] = f2{a[i], b[i]}; }
    float c = seed * 0.5f + 1.0f, r0 = seed + 2.0f, r1 = seed + 3.0f;
    f2 c2 = f2{c, c * 1.0001f};
    f4 acc4 = {0, 0, 0, 0}, acc4b = {0, 0, 0, 0};
    f16v acc16 = {0};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long w0 = __builtin_amdgcn_s_memrealtime();
    for (int t = 0; t < trips; ++t) {
#pragma unroll
        for (int r = 0; r < 32; ++r) {
            if (KIND == 0 || KIND == 3 || KIND == 4 || KIND == 7 || KIND == 9 || KIND == 10) {
#pragma unroll
                for (int i = 0; i < U; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(b[i]));
            }
            if (KIND == 0) asm volatile("v_rsq_f32 %0, %0" : "+v"(r0));
            if (KIND == 1) {
#pragma unroll
                for (int i = 0; i < U; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(b[i]));
#pragma unroll
                for (int i = 0; i < U; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(b[i]));
                asm volatile("v_rsq_f32 %0, %0" : "+v"(r0));
                asm volatile("v_rsq_f32 %0, %0" : "+v"(r1));
            }
            if (KIND == 2) {
#pragma unroll
                for (int i = 0; i < U; ++i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                asm volatile("v_rsq_f32 %0, %0" : "+v"(r0));
            }
            if (KIND == 4) asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(acc4) : "v"(c), "v"(r0));
            if (KIND == 5) {
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(b[i]));
                asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(acc4) : "v"(c), "v"(r0));
            }
            if (KIND == 6) {
                asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(acc4) : "v"(c), "v"(r0));
                asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(acc4b) : "v"(c), "v"(r0));
            }
            if (KIND == 7 && (r & 3) == 3) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc4) : "v"(c), "v"(r0));
            if (KIND == 8) {
#pragma unroll
                for (int i = 0; i < 6; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(c2), "v"(c2));
                asm volatile("v_rsq_f32 %0, %0" : "+v"(r0));
            }
            if (KIND == 9) {  // 12 fma + rsq + mfma4x4x1
                asm volatile("v_rsq_f32 %0, %0" : "+v"(r0));
                asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(acc4) : "v"(c), "v"(r1));
            }
            if (KIND == 10 && (r & 7) == 7) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc16) : "v"(c), "v"(r0));
            if (KIND == 11) {  // 9 fma + rsq + mfma4x4x1 : the MFMA-assisted interaction shape
#pragma unroll
                for (int i = 0; i < 9; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(b[i]));
                asm volatile("v_rsq_f32 %0, %0" : "+v"(r0));
                asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(acc4) : "v"(c), "v"(r1));
            }
            if (KIND == 12) {  // op_sel broadcast form of pk_fma: both halves take src1.lo
#pragma unroll
                for (int i = 0; i < 6; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,0,1]" : "+v"(p[i]) : "v"(c2), "v"(c2));
                asm volatile("v_rsq_f32 %0, %0" : "+v"(r0));
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long w1 = __builtin_amdgcn_s_memrealtime();
    float s = r0 + r1 + acc4[0] + acc4[1] + acc4b[2] + acc16[0] + acc16[7];
    for (int i = 0; i < U; ++i) s += a[i] + p[i].x + p[i].y;
    if (s == 123.456f) out[0] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}

template <int KIND>
int run(const char* name, double valu_per_group, double trans_per_group) {
    float* out;
    unsigned long long* clk;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    CHECK(hipMalloc(&out, 4));
    CHECK(hipMalloc(&clk, 16 * cus * 8));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int wps : {1, 2, 4}) {
        int blocks = cus * wps;
        int trips = 4000 / wps;
        k_mix<KIND><<<blocks, 256>>>(out, clk, 10, 1.0f);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        k_mix<KIND><<<blocks, 256>>>(out, clk, trips, 1.0f);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> h(2 * blocks);
        CHECK(hipMemcpy(h.data(), clk, 16 * blocks, hipMemcpyDeviceToHost));
        std::vector<double> mhz;
        for (int b = 0; b < blocks; ++b) mhz.push_back((double)h[2 * b] / (double)h[2 * b + 1] * 100.0);
        std::sort(mhz.begin(), mhz.end());
        double groups = (double)trips * 32;  // per wave
        double cyc_per_group = (double)h[0] / groups;  // block 0's wave-0 cycles per group (shared SIMD with wps waves)
        double ns_per_group_simd = ms * 1e6 / (groups * wps);  // each SIMD runs wps waves
        printf("%-28s w/SIMD=%d  %.3f ms  clock median %.0f MHz (min %.0f max %.0f)  %.2f ns/group/SIMD = %.1f cyc @clock  [%g valu + %g trans]\n",
               name, wps, ms, mhz[mhz.size() / 2], mhz.front(), mhz.back(), ns_per_group_simd,
               ns_per_group_simd * mhz[mhz.size() / 2] * 1e-3, valu_per_group, trans_per_group);
        (void)cyc_per_group;
    }
    CHECK(hipFree(out));
    CHECK(hipFree(clk));
    return 0;
}

int main() {
    run<3>("12 fma", 12, 0);
    run<0>("12 fma + 1 rsq", 12, 1);
    run<1>("24 fma + 2 rsq (adjacent)", 24, 2);
    run<2>("12 mul + 1 rsq", 12, 1);
    run<8>("6 pk_fma + 1 rsq", 6, 1);
    run<12>("6 pk_fma(op_sel) + 1 rsq", 6, 1);
    run<6>("2 mfma4x4x1", 0, 0);
    run<4>("12 fma + 1 mfma4x4x1", 12, 0);
    run<5>("8 fma + 1 mfma4x4x1", 8, 0);
    run<9>("12 fma + rsq + mfma4x4x1", 12, 1);
    run<11>("9 fma + rsq + mfma4x4x1", 9, 1);
    run<7>("12 fma + 1/4 mfma16x16x4", 12, 0);
    run<10>("12 fma + 1/8 mfma32x32x2", 12, 0);
    return 0;
}
