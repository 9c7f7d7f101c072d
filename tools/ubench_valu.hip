// Micro-benchmark: sustained chip-wide issue rate of the VALU instructions the all-pairs inner
// loop is made of (v_fma_f32, v_pk_fma_f32, v_rsq_f32, v_pk_mul_f32, v_pk_add_f32) on gfx950.
// Prints instructions/s per kind and the implied TFLOP/s, for several waves-per-SIMD settings.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int UNROLL = 16;   // independent chains
constexpr int INNER = 64;    // repetitions of the unrolled group per loop trip

template <int KIND>
__global__ __launch_bounds__(256) void k_rate(float* out, int trips, float seed) {
    float a[UNROLL], b[UNROLL];
    typedef float float2_ __attribute__((ext_vector_type(2)));
    float2_ p[UNROLL];
    for (int i = 0; i < UNROLL; ++i) { a[i] = seed + i + threadIdx.x * 1e-3f; b[i] = 1.0f + i * 1e-6f; p[i] = float2_{a[i], b[i]}; }
    float c = seed * 0.5f + 1.0f;
    float2_ c2 = float2_{c, c * 1.0001f};
    for (int t = 0; t < trips; ++t) {
#pragma unroll
        for (int r = 0; r < INNER; ++r) {
#pragma unroll
            for (int i = 0; i < UNROLL; ++i) {
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(b[i]));
                if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(c2), "v"(c2));
                if (KIND == 2) asm volatile("v_rsq_f32 %0, %0" : "+v"(a[i]));
                if (KIND == 3) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(c2));
                if (KIND == 4) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(c2));
                if (KIND == 5) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                if (KIND == 6) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                if (KIND == 7) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
                if (KIND == 8) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
                if (KIND == 9) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "s"(c), "v"(b[i]));
                if (KIND == 10) {  // mixed: 12 fma + 1 rsq pattern
                    if ((i % 13) == 12) asm volatile("v_rsq_f32 %0, %0" : "+v"(a[i]));
                    else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(b[i]));
                }
                if (KIND == 11) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(b[i]));
                if (KIND == 12) asm volatile("v_log_f32 %0, %0" : "+v"(a[i]));
                if (KIND == 13) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
            }
        }
    }
    float s = 0;
    for (int i = 0; i < UNROLL; ++i) s += a[i] + p[i].x + p[i].y;
    if (s == 123.456f) out[0] = s;
}

template <int KIND>
int run(const char* name, double flop_per_lane_inst) {
    float* out;
    CHECK(hipMalloc(&out, 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    for (int wps : {1, 2, 4, 8}) {
        // 256-thread blocks = 4 waves = 1 wave per SIMD per block; wps blocks per CU
        int blocks = cus * wps;
        int trips = 2000 / wps;
        k_rate<KIND><<<blocks, 256>>>(out, 10, 1.0f);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        k_rate<KIND><<<blocks, 256>>>(out, trips, 1.0f);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        double winst = (double)blocks * 4 * trips * INNER * UNROLL;  // wave-instructions
        double rate = winst / (ms * 1e-3);
        double per_simd_ghz = rate / (cus * 4.0) / 1e9;  // wave-instr per ns per SIMD
        printf("%-14s waves/SIMD=%d  %.3f ms  %.3e winst/s  %.4f winst/ns/SIMD (=> %.2f cyc/inst @2.4GHz)  %.1f TFLOP/s\n",
               name, wps, ms, rate, per_simd_ghz, 2.4 / per_simd_ghz, rate * 64 * flop_per_lane_inst / 1e12);
    }
    CHECK(hipFree(out));
    return 0;
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    printf("device %s CUs=%d clock=%d kHz\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate);
    run<0>("v_fma_f32", 2);
    run<11>("v_fmac_f32", 2);
    run<9>("v_fma_f32(sgpr)", 2);
    run<1>("v_pk_fma_f32", 4);
    run<3>("v_pk_mul_f32", 2);
    run<4>("v_pk_add_f32", 2);
    run<5>("v_mul_f32", 1);
    run<6>("v_sub_f32", 1);
    run<2>("v_rsq_f32", 1);
    run<7>("v_rcp_f32", 1);
    run<8>("v_sqrt_f32", 1);
    run<12>("v_log_f32", 1);
    run<13>("v_exp_f32", 1);
    run<10>("12fma+1rsq", 24.0 / 13.0);
    return 0;
}
