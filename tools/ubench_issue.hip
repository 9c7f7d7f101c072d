// Micro-benchmark: do vector, scalar and branch instructions of DIFFERENT wavefronts on one SIMD
// issue side by side on gfx950, or one at a time?  Each kernel repeats a fixed instruction mix;
// the printed figure is SIMD cycles per repetition at 1, 2, 4 and 8 wavefronts per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o ubench_issue ubench_issue.hip
#include <hip/hip_runtime.h>
#include <cstdio>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// KIND 0: 16 VALU            1: 16 SALU            2: 16 VALU + 16 SALU interleaved
//      3: 16 VALU + 8 SALU   4: 16 VALU + 4 untaken branches   5: 16 VALU + 4 taken branches
//      6: 16 VALU + 8 SALU + 4 taken branches (about the traversal's mix per visit)
template <int KIND>
__global__ __launch_bounds__(256) void k_mix(float* out, int trips, float seed) {
    float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3;
    const float c = seed * 0.5f + 1.0f;
    unsigned s0 = 1, s1 = 2, s2 = 3, s3 = 4;
    for (int t = 0; t < trips; ++t) {
#define V4 "v_fma_f32 %0, %0, %8, %8\n\tv_fma_f32 %1, %1, %8, %8\n\tv_fma_f32 %2, %2, %8, %8\n\tv_fma_f32 %3, %3, %8, %8\n\t"
#define S4 "s_add_u32 %4, %4, 1\n\ts_add_u32 %5, %5, 1\n\ts_add_u32 %6, %6, 1\n\ts_add_u32 %7, %7, 1\n\t"
#define S2 "s_add_u32 %4, %4, 1\n\ts_add_u32 %5, %5, 1\n\t"
#define BN "s_cmp_eq_u32 %4, 0\n\ts_cbranch_scc1 1f\n\t1:\n\t"   /* s4 is never 0 here: untaken */
#define BT "s_cmp_lg_u32 %4, 0\n\ts_cbranch_scc1 1f\n\ts_nop 0\n\t1:\n\t" /* taken, skips one instruction */
        if (KIND == 0) asm volatile(V4 V4 V4 V4 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "v"(c) : "scc");
        if (KIND == 1) asm volatile(S4 S4 S4 S4 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "v"(c) : "scc");
        if (KIND == 2) asm volatile(V4 S4 V4 S4 V4 S4 V4 S4 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "v"(c) : "scc");
        if (KIND == 3) asm volatile(V4 S2 V4 S2 V4 S2 V4 S2 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "v"(c) : "scc");
        if (KIND == 4) asm volatile(V4 BN V4 BN V4 BN V4 BN : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "v"(c) : "scc");
        if (KIND == 5) asm volatile(V4 BT V4 BT V4 BT V4 BT : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "v"(c) : "scc");
        if (KIND == 6) asm volatile(V4 S2 BT V4 S2 BT V4 S2 BT V4 S2 BT : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "v"(c) : "scc");
    }
    if (a0 + a1 + a2 + a3 + (float)(s0 + s1 + s2 + s3) == 123.456f) out[0] = a0;
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
    const int cus = prop.multiProcessorCount;
    for (int wps : {1, 2, 4, 8}) {
        const int blocks = cus * wps, trips = 200000 / wps;
        k_mix<KIND><<<blocks, 256>>>(out, 100, 1.0f);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        k_mix<KIND><<<blocks, 256>>>(out, trips, 1.0f);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        // repetitions executed per SIMD = wps * trips; cycles at the nominal 2.4 GHz
        const double cyc = ms * 1e-3 * 2.4e9 / ((double)wps * trips);
        printf("%-44s waves/SIMD=%d  %8.3f ms  %7.1f cycles per repetition per SIMD (at 2.4 GHz)\n", name, wps, ms, cyc);
    }
    CHECK(hipFree(out));
    return 0;
}

int main() {
    run<0>("16 VALU");
    run<1>("16 SALU");
    run<2>("16 VALU + 16 SALU");
    run<3>("16 VALU + 8 SALU");
    run<4>("16 VALU + 4 (cmp + untaken branch)");
    run<5>("16 VALU + 4 (cmp + taken branch)");
    run<6>("16 VALU + 8 SALU + 4 (cmp + taken branch)");
    return 0;
}
