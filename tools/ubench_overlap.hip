// Micro-benchmark: do quarter-rate transcendentals (v_rsq_f32 ...) overlap with packed FP32 work on gfx950?
// Independent register chains; P packed FMAs and T transcendentals per group, grouped or interleaved;
// cycles per group per SIMD at 1, 4 and 8 resident wavefronts.
//   hipcc --offload-arch=gfx950 -O3 -o ubench_overlap ubench_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));
#define PK(i) "v_pk_fma_f32 %" #i ", %" #i ", %8, %9\n\t"
#define TR(r) "v_rsq_f32 " r ", " r "\n\t"
#define FM(r) "v_fma_f32 " r ", " r ", v40, v41\n\t"
template <int KIND>
__global__ __launch_bounds__(64) void k(float* out, int trips, float seed) {
    f2 a0 = {seed + threadIdx.x, seed}, a1 = a0 + 1.0f, a2 = a0 + 2.0f, a3 = a0 + 3.0f, a4 = a0 + 4.0f, a5 = a0 + 5.0f, a6 = a0 + 6.0f, a7 = a0 + 7.0f;
    const f2 c = {1.0001f, 1.0002f}, b = {0.5f, 0.25f};
    for (int t = 0; t < trips; ++t) {
#define OPS : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(b) : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49"
        if (KIND == 0) asm volatile(PK(0) PK(1) PK(2) PK(3) PK(4) PK(5) PK(6) PK(7) OPS);
        if (KIND == 1) asm volatile(PK(0) PK(1) PK(2) PK(3) PK(4) PK(5) PK(6) PK(7) TR("v42") TR("v43") OPS);
        if (KIND == 2) asm volatile(PK(0) PK(1) PK(2) PK(3) TR("v42") PK(4) PK(5) PK(6) PK(7) TR("v43") OPS);
        if (KIND == 3) asm volatile(TR("v42") TR("v43") TR("v44") TR("v45") OPS);
        if (KIND == 4) asm volatile(PK(0) PK(1) TR("v42") PK(2) PK(3) TR("v43") PK(4) PK(5) TR("v44") PK(6) PK(7) TR("v45") OPS);
        if (KIND == 5) asm volatile(PK(0) TR("v42") PK(1) TR("v43") PK(2) TR("v44") PK(3) TR("v45") PK(4) TR("v46") PK(5) TR("v47") PK(6) TR("v48") PK(7) TR("v49") OPS);
        if (KIND == 6) asm volatile(PK(0) PK(1) PK(2) PK(3) PK(4) PK(5) PK(6) PK(7) TR("v42") TR("v43") TR("v44") TR("v45") TR("v46") TR("v47") TR("v48") TR("v49") OPS);
        if (KIND == 7) asm volatile(FM("v42") FM("v43") FM("v44") FM("v45") FM("v46") FM("v47") FM("v48") FM("v49") OPS);
        if (KIND == 8) asm volatile(FM("v42") TR("v43") FM("v44") TR("v45") FM("v46") TR("v47") FM("v48") TR("v49") OPS);
    }
    f2 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (s.x + s.y == 123.456f) out[0] = s.x;
}
template <int KIND>
int run(const char* name) {
    float* out; CHECK(hipMalloc(&out, 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int simds = prop.multiProcessorCount * 4;
    for (int wps : {1, 4, 8}) {
        const int blocks = simds * wps, trips = 200000 / wps;
        k<KIND><<<blocks, 64>>>(out, 10, 1.0f);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        k<KIND><<<blocks, 64>>>(out, trips, 1.0f);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-46s waves/SIMD=%d %8.3f ms %6.1f cycles per group per SIMD (at 2.4 GHz)\n", name, wps, ms, ms * 1e-3 * 2.4e9 / ((double)wps * trips));
    }
    return 0;
}
int main() {
    run<0>("8 pk_fma");
    run<1>("8 pk_fma, then 2 rsq");
    run<2>("8 pk_fma with 2 rsq interleaved");
    run<3>("4 rsq");
    run<4>("8 pk_fma with 4 rsq interleaved");
    run<5>("8 pk_fma with 8 rsq interleaved");
    run<6>("8 pk_fma, then 8 rsq");
    run<7>("8 fma");
    run<8>("4 fma with 4 rsq interleaved");
    return 0;
}
