// Wave specialisation test: does moving every VMEM (LDS-DMA) and VALU instruction off the waves that issue the
// f32 MFMAs raise the matrix-pipe rate?  One "step" = 32 MFMA + 12 ds_read_b128 per MFMA wave, then a barrier.
//  SPEC 0: 256-thread workgroup, every wave also issues D LDS-DMAs + V v_fma per step and waits vmcnt(0)
//  SPEC 1: 512-thread workgroup, waves 0-3 issue only ds_read + MFMA + barrier; waves 4-7 issue the D LDS-DMAs
//          + V v_fma per step each, wait vmcnt(0), barrier.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_ws.hip -o tools/ubench/mfma_ws
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define LDS_AS __attribute__((address_space(3)))
__device__ __forceinline__ void glds16(const float* gsrc, unsigned lds) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds) : "memory");
}
template <int V>
__device__ __forceinline__ void valu_block(float& x0, float& x1, float& x2, float& x3, float c) {
#pragma unroll
    for (int i = 0; i < V; ++i) {
        if ((i & 3) == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x0) : "v"(c));
        if ((i & 3) == 1) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x1) : "v"(c));
        if ((i & 3) == 2) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x2) : "v"(c));
        if ((i & 3) == 3) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x3) : "v"(c));
    }
}

template <int SPEC, int D, int V, int OCC>
__global__ __launch_bounds__(SPEC ? 512 : 256, OCC) void k(const float* g, float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 8192; i += (SPEC ? 512 : 256)) ((float*)smem)[i] = g[i];
    __syncthreads();
    const unsigned lds0 = (unsigned)(size_t)((LDS_AS unsigned char*)smem);
    const bool mfma_wave = !SPEC || wave < 4;
    const bool helper = !SPEC || wave >= 4;
    const int hw = SPEC ? (wave & 3) : wave, ht = tid & 255;
    f32x16 acc0 = {0}, acc1 = {0};
    const int ao = (tid & 63) * 128 + ((hw) << 4);
    f32x4 a0 = *(f32x4*)(smem + ao), a1 = *(f32x4*)(smem + 8192 + ao), b = *(f32x4*)(smem + 16384 + ao);
    float x0 = tid * 0.001f, x1 = 1.f + tid * 1e-4f, x2 = x0 + x1, x3 = x0 - x1;
    for (int it = 0; it < iters; ++it) {
        if (helper) {
#pragma unroll
            for (int d = 0; d < D; ++d) glds16(g + ht * 4 + d * 8192 + (it & 7) * 1024, lds0 + 24576 + d * 4096 + hw * 1024);
            valu_block<V>(x0, x1, x2, x3, 0.999f);
        }
        if (mfma_wave) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                f32x4 na0 = *(f32x4*)(smem + (ao ^ (s << 5)));
                f32x4 na1 = *(f32x4*)(smem + 8192 + (ao ^ (s << 5)));
                f32x4 nb = *(f32x4*)(smem + 16384 + (ao ^ (s << 5)));
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b.x, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, b.x, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b.y, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, b.y, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, b.z, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, b.z, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, b.w, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, b.w, acc1, 0, 0, 0);
                a0 = na0; a1 = na1; b = nb;
            }
        }
        if (helper) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    const size_t o = ((size_t)blockIdx.x * 512 + tid) * 33;
    for (int i = 0; i < 16; ++i) out[o + i] = acc0[i], out[o + 16 + i] = acc1[i];
    out[o + 32] = x0 + x1 + x2 + x3;
}
template <int SPEC, int D, int V, int OCC>
void run(const float* g, float* out) {
    const int iters = 3000, grid = 256 * OCC * 4;
    hipFuncSetAttribute((const void*)k<SPEC, D, V, OCC>, hipFuncAttributeMaxDynamicSharedMemorySize, 24576 + D * 4096);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k<SPEC, D, V, OCC>), dim3(grid), dim3(SPEC ? 512 : 256), 24576 + D * 4096, 0, g, out, iters);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
    }
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)grid * 4 * iters * 32 * 4096.0;
    printf("%s  D=%d DMA + V=%2d v_fma per wave-step, %d WG/CU: %8.3f ms  %6.1f TFLOP/s (%.1f%% of 157.3)\n",
           SPEC ? "specialised (4 MFMA + 4 helper waves)" : "uniform     (4 waves do everything)  ", D, V, OCC, ms, flops / ms / 1e9, flops / ms / 1e9 / 157.3 * 100);
}
int main() {
    float *g, *out;
    hipMalloc(&g, 1 << 22);
    hipMalloc(&out, (size_t)256 * 16 * 512 * 33 * 4);
    hipMemset(g, 0, 1 << 22);
    run<0, 2, 0, 2>(g, out); run<1, 2, 0, 2>(g, out); run<0, 2, 0, 3>(g, out); run<1, 2, 0, 3>(g, out);
    run<0, 2, 8, 2>(g, out); run<1, 2, 8, 2>(g, out); run<0, 2, 8, 3>(g, out); run<1, 2, 8, 3>(g, out);
    run<0, 4, 8, 2>(g, out); run<1, 4, 8, 2>(g, out); run<0, 4, 8, 3>(g, out); run<1, 4, 8, 3>(g, out);
    run<0, 4, 32, 2>(g, out); run<1, 4, 32, 2>(g, out); run<0, 4, 32, 3>(g, out); run<1, 4, 32, 3>(g, out);
    run<1, 4, 32, 1>(g, out); run<0, 4, 32, 1>(g, out);
    return 0;
}
