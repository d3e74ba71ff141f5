// Does v_mfma_f32_32x32x2_f32 sustain more than one instruction per 64 cycles per SIMD when several
// waves share the SIMD?  Bare register-only loop, W waves per SIMD, wall-clock TFLOP/s via HIP events.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = seed * (float)(r + i);
    float a = seed + threadIdx.x * 1e-3f, b = seed - threadIdx.x * 2e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC>
void run(float* out, int wg_per_cu) {
    const int iters = 4000, grid = 256 * wg_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(256), 0, 0, out, iters, 0.001f);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
    }
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)grid * 4 * iters * 8 * NACC * 4096.0;
    printf("NACC=%d  %d WG/CU (%d waves/SIMD): %.3f ms  %.1f TFLOP/s  (%.1f%% of 157.3)\n", NACC, wg_per_cu, wg_per_cu, ms,
           flops / ms / 1e9, flops / ms / 1e9 / 157.3 * 100);
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 8 * 256 * 4);
    run<1>(out, 1);
    run<2>(out, 1);
    run<2>(out, 2);
    run<2>(out, 3);
    run<2>(out, 4);
    run<1>(out, 4);
    run<4>(out, 2);
    return 0;
}
