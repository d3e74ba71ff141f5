// Do VALU instructions overlap with v_mfma_f32_32x32x2_f32 on gfx950?
//  mode 0: ONE wave per SIMD, loop body = 1 MFMA (dependent chain) + K independent v_fma_f32 -> cycles per iteration
//  mode 1: TWO waves per SIMD (2 workgroups per CU): even workgroups run MFMA only, odd ones run VALU only
//          (K v_fma per iteration) -> cycles per iteration of each when run together vs alone.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_valu_coexec.hip -o tools/ubench/mfma_valu_coexec
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int K>
__device__ __forceinline__ void valu_block(float& x0, float& x1, float& x2, float& x3, float c) {
#pragma unroll
    for (int i = 0; i < K; ++i) {
        // four independent chains so that no v_fma waits for the previous one
        if ((i & 3) == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x0) : "v"(c));
        if ((i & 3) == 1) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x1) : "v"(c));
        if ((i & 3) == 2) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x2) : "v"(c));
        if ((i & 3) == 3) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x3) : "v"(c));
    }
}

template <int K, int NACC>
__global__ __launch_bounds__(256, 2) void k_same_wave(float* out, unsigned long long* cyc, int iters) {
    f32x16 acc0 = {0}, acc1 = {0};
    float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 1e-4f;
    float x0 = a, x1 = b, x2 = a + b, x3 = a - b;
    const float c = 0.999f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc0) : "v"(a), "v"(b));
        if (NACC == 2) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc1) : "v"(a), "v"(b));
        valu_block<K>(x0, x1, x2, x3, c);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = x0 + x1 + x2 + x3;
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// role 0 = MFMA only, 1 = VALU only; which = 0 both kinds resident together, 1 only MFMA workgroups work, 2 only VALU ones
template <int K>
__global__ __launch_bounds__(256, 2) void k_two_waves(float* out, unsigned long long* cyc, int iters, int which, int* cu_count, int* roles) {
    // role by arrival order on the CU, so that each CU gets one workgroup of each kind
    __shared__ int s_role;
    if (threadIdx.x == 0) {
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 15u;
        const unsigned cu = (((xcc * 8 + ((hw >> 13) & 7)) * 2 + ((hw >> 12) & 1)) * 16 + ((hw >> 8) & 15));
        s_role = atomicAdd(&cu_count[cu], 1) & 1;
        roles[blockIdx.x] = s_role;
    }
    __syncthreads();
    const int role = s_role;
    f32x16 acc0 = {0};
    float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 1e-4f;
    float x0 = a, x1 = b, x2 = a + b, x3 = a - b;
    const float c = 0.999f;
    const bool work = (which == 0) || (which == 1 && role == 0) || (which == 2 && role == 1);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (work) {
        if (role == 0) {
            for (int it = 0; it < iters; ++it) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc0) : "v"(a), "v"(b));
        } else {
            for (int it = 0; it < iters; ++it) valu_block<K>(x0, x1, x2, x3, c);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = x0 + x1 + x2 + x3;
    for (int r = 0; r < 16; ++r) s += acc0[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main() {
    const int NB = 512, iters = 20000;
    float* out; unsigned long long* cyc; int *cu_count, *roles;
    CK(hipMalloc(&out, NB * 256 * 4)); CK(hipMalloc(&cyc, NB * 8)); CK(hipMalloc(&cu_count, 4096 * 4)); CK(hipMalloc(&roles, NB * 4));
    std::vector<int> hr(NB);
    std::vector<unsigned long long> h(NB);
    auto mean = [&](int n, int stride, int off) { double s = 0; int c = 0; for (int i = off; i < n; i += stride) { s += h[i]; ++c; } return s / c / iters; };
#define SAME(K, NACC) { hipLaunchKernelGGL((k_same_wave<K, NACC>), dim3(256), dim3(256), 0, 0, out, cyc, iters); CK(hipDeviceSynchronize()); \
        CK(hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost)); \
        printf("same wave, 1 wave/SIMD: %d MFMA + %2d v_fma per iteration: %.1f cycles/iter\n", NACC, K, mean(256, 1, 0)); }
    SAME(0, 1) SAME(4, 1) SAME(8, 1) SAME(12, 1) SAME(16, 1) SAME(24, 1) SAME(0, 2) SAME(8, 2) SAME(16, 2) SAME(24, 2) SAME(32, 2)
    auto rmean = [&](int role) { double s = 0; int c = 0; for (int i = 0; i < 512; ++i) if (hr[i] == role) { s += h[i]; ++c; } return c ? s / c / iters : 0.0; };
#define TWO(K) for (int which = 0; which < 3; ++which) { CK(hipMemset(cu_count, 0, 4096 * 4)); \
        hipLaunchKernelGGL((k_two_waves<K>), dim3(512), dim3(256), 0, 0, out, cyc, iters, which, cu_count, roles); CK(hipDeviceSynchronize()); \
        CK(hipMemcpy(h.data(), cyc, 512 * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(hr.data(), roles, 512 * 4, hipMemcpyDeviceToHost)); \
        printf("two waves/SIMD, K=%2d, %s: MFMA wave %.1f cycles/iter, VALU wave %.1f cycles/iter (%.2f per v_fma)\n", K, \
               which == 0 ? "both working " : which == 1 ? "MFMA alone   " : "VALU alone   ", rmean(0), rmean(1), rmean(1) / K); }
#define SAME2(K, NACC) { hipLaunchKernelGGL((k_same_wave<K, NACC>), dim3(512), dim3(256), 0, 0, out, cyc, iters); CK(hipDeviceSynchronize()); \
        CK(hipMemcpy(h.data(), cyc, 512 * 8, hipMemcpyDeviceToHost)); \
        printf("same program, 2 waves/SIMD: %d MFMA + %2d v_fma per iteration: %.1f cycles/iter per wave\n", NACC, K, mean(512, 1, 0)); }
    SAME2(0, 1) SAME2(4, 1) SAME2(8, 1) SAME2(16, 1) SAME2(0, 2) SAME2(8, 2) SAME2(16, 2)
    TWO(8) TWO(16)
    return 0;
}
