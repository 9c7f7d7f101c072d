// Micro-benchmark: what one v_rsq_f32 costs inside a stream of v_fma_f32 on gfx950 (cycles per group per SIMD at
// 1, 4 and 8 resident wavefronts): alone it issues in 8 cycles, in a mix it costs 11 to 22.
//   hipcc --offload-arch=gfx950 -O3 -o ubench_trans ubench_trans.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
#define F "v_fma_f32 %0, %0, %4, %5\n\tv_fma_f32 %1, %1, %4, %5\n\tv_fma_f32 %2, %2, %4, %5\n\tv_fma_f32 %3, %3, %4, %5\n\t"
#define F2 "v_fma_f32 %0, %0, %4, %5\n\tv_fma_f32 %1, %1, %4, %5\n\t"
#define FS "v_fma_f32 %0, %0, %6, %5\n\tv_fma_f32 %1, %1, %6, %5\n\tv_fma_f32 %2, %2, %6, %5\n\tv_fma_f32 %3, %3, %6, %5\n\t"
#define R(x) "v_rsq_f32 " x ", " x "\n\t"
#define X8(Q) Q Q Q Q Q Q Q Q
template <int KIND, int THREADS>
__global__ __launch_bounds__(THREADS) void k(float* out, int trips, float seed, float sc) {
    float a0 = seed + threadIdx.x, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, e = seed + 7;
    const float c = 1.0001f, b = 0.5f;
    for (int t = 0; t < trips; ++t) {
#define OPS : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c), "v"(b), "s"(sc), "v"(e)
        if (KIND == 0) asm volatile(X8(F F F F2) OPS);                 // 14 fma
        if (KIND == 1) asm volatile(X8(F F F F2 R("%0")) OPS);         // 14 fma + rsq on a chain value
        if (KIND == 2) asm volatile(X8(FS FS FS F2 R("%0")) OPS);      // sgpr operand
        if (KIND == 3) asm volatile(X8(F F F F2 R("%0") R("%1")) OPS); // two rsq
        if (KIND == 4) asm volatile(X8(F F F F F F F R("%0")) OPS);    // 28 fma + rsq
        if (KIND == 5) asm volatile(X8(F F2 R("%0") F F2 R("%1")) OPS);// 6 fma + rsq, twice
    }
    if (a0 + a1 + a2 + a3 == 123.456f) out[0] = a0;
}
template <int KIND, int THREADS>
int run(const char* name, int per) {
    float* out; CHECK(hipMalloc(&out, 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int simds = prop.multiProcessorCount * 4;
    for (int wps : {1, 4, 8}) {
        const int blocks = simds * wps * 64 / THREADS, trips = 40000 / wps;
        k<KIND, THREADS><<<blocks, THREADS>>>(out, 10, 1.0f, 1.0001f);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        k<KIND, THREADS><<<blocks, THREADS>>>(out, trips, 1.0f, 1.0001f);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-34s threads=%3d waves/SIMD=%d %8.3f ms %6.1f cycles per group of %d (at 2.4 GHz)\n", name, THREADS, wps, ms,
               ms * 1e-3 * 2.4e9 / ((double)wps * trips * 8), per);
    }
    return 0;
}
int main() {
    run<0, 64>("14 fma", 14); run<1, 64>("14 fma + rsq", 15); run<1, 256>("14 fma + rsq", 15); run<2, 64>("14 fma(sgpr) + rsq", 15);
    run<3, 64>("14 fma + 2 rsq", 16); run<4, 64>("28 fma + rsq", 29); run<5, 64>("2 x (6 fma + rsq)", 14);
    return 0;
}
