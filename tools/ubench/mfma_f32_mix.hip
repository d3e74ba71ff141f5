// Wall-clock TFLOP/s of the conv kernel's instruction mix (8 MFMA f32 per 3 ds_read_b128, 2 LDS-DMA +
// vmcnt(0) + barrier per 32 MFMA) at 1..4 workgroups per CU.  Separates "instruction mix ceiling" from
// "memory / tile effects" of the real kernel.
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
// MIX: 0 MFMA only | 1 + ds_reads | 2 + barrier | 3 + LDS-DMA + vmcnt
template <int MIX, int OCC>
__global__ __launch_bounds__(256, OCC) void k(const float* g, float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 8192; i += 256) ((float*)smem)[i] = g[i];
    __syncthreads();
    const unsigned lds0 = (unsigned)(size_t)((LDS_AS unsigned char*)smem);
    f32x16 acc0 = {0}, acc1 = {0};
    const int ao = (tid & 63) * 128 + ((tid >> 6) << 4);
    f32x4 a0 = *(f32x4*)(smem + ao), a1 = *(f32x4*)(smem + 8192 + ao), b = *(f32x4*)(smem + 16384 + ao);
    for (int it = 0; it < iters; ++it) {
        f32x4 r0, r1;
        if (MIX == 3 || MIX == 6) {
            glds16(g + tid * 4 + (it & 7) * 1024, lds0 + 24576 + wave * 1024);
            glds16(g + tid * 4 + 8192 + (it & 7) * 1024, lds0 + 24576 + 4096 + wave * 1024);
        }
        if (MIX == 6) {
            glds16(g + tid * 4 + 16384 + (it & 7) * 1024, lds0 + 24576 + 8192 + wave * 1024);
            glds16(g + tid * 4 + 24576 + (it & 7) * 1024, lds0 + 24576 + 12288 + wave * 1024);
        }
        if (MIX == 7) glds16(g + tid * 4 + (it & 7) * 1024, lds0 + 24576 + wave * 1024);
        if (MIX == 5) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + tid * 4 + (it & 7) * 1024), (LDS_AS void*)(smem + 24576 + wave * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + tid * 4 + 8192 + (it & 7) * 1024), (LDS_AS void*)(smem + 24576 + 4096 + wave * 1024), 16, 0, 0);
        }
        if (MIX == 4) {  // register staging: same bytes through VGPRs
            r0 = *(const f32x4*)(g + tid * 4 + (it & 7) * 1024);
            r1 = *(const f32x4*)(g + tid * 4 + 8192 + (it & 7) * 1024);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            f32x4 na0 = a0, na1 = a1, nb = b;
            if (MIX >= 1) {
                na0 = *(f32x4*)(smem + (ao ^ (s << 5)));
                na1 = *(f32x4*)(smem + 8192 + (ao ^ (s << 5)));
                nb = *(f32x4*)(smem + 16384 + (ao ^ (s << 5)));
            }
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
        if (MIX == 4) {
            *(f32x4*)(smem + 24576 + tid * 16) = r0;
            *(f32x4*)(smem + 24576 + 4096 + tid * 16) = r1;
        }
        if (MIX == 3 || MIX == 6 || MIX == 7) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (MIX >= 2) __syncthreads();
    }
    for (int i = 0; i < 16; ++i) out[((size_t)blockIdx.x * 256 + tid) * 32 + i] = acc0[i], out[((size_t)blockIdx.x * 256 + tid) * 32 + 16 + i] = acc1[i];
}
template <int MIX, int OCC>
void run(const float* g, float* out) {
    const int iters = 3000, grid = 256 * OCC * 4;   // 4 rounds of resident workgroups
    hipFuncSetAttribute((const void*)k<MIX, OCC>, hipFuncAttributeMaxDynamicSharedMemorySize, 40960);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k<MIX, OCC>), dim3(grid), dim3(256), 40960, 0, g, out, iters);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
    }
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)grid * 4 * iters * 32 * 4096.0;
    printf("mix=%d  %d WG/CU: %8.3f ms  %6.1f TFLOP/s (%.1f%% of 157.3)\n", MIX, OCC, ms, flops / ms / 1e9, flops / ms / 1e9 / 157.3 * 100);
}
int main() {
    float *g, *out;
    hipMalloc(&g, 1 << 22);
    hipMalloc(&out, (size_t)256 * 16 * 256 * 32 * 4);
    hipMemset(g, 0, 1 << 22);
    run<0, 1>(g, out); run<0, 3>(g, out);
    run<1, 1>(g, out); run<1, 2>(g, out); run<1, 3>(g, out);
    run<2, 1>(g, out); run<2, 2>(g, out); run<2, 3>(g, out);
    run<3, 1>(g, out); run<3, 2>(g, out); run<3, 3>(g, out);
    printf("-- 7: one LDS-DMA per 32 MFMA; 6: four; 5: two via the builtin; 4: two KiB via VGPRs + ds_write_b128\n");
    run<7, 2>(g, out); run<7, 3>(g, out);
    run<6, 2>(g, out); run<6, 3>(g, out);
    run<5, 2>(g, out); run<5, 3>(g, out);
    run<4, 1>(g, out); run<4, 2>(g, out); run<4, 3>(g, out);
    return 0;
}
