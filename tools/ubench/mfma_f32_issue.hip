// Microbenchmark: cycles per v_mfma_f32_32x32x2_f32 for one wave per SIMD (256-thread WG, 1 WG/CU)
// under the ingredients of k_conv_mfma_p: accumulator count, ds_read_b128 fragments, LDS-DMA.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define LDS_AS __attribute__((address_space(3)))

__device__ __forceinline__ void glds16(const float* gsrc, unsigned lds) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds) : "memory");
}

// VAR: 0 bare 1 acc | 1 bare 2 acc | 2: 2 acc + ds_read frags (3 per 8 mfma) | 3: +2 glds per 32 mfma
// 4: = 3 + barrier per 32 | 5: 1 acc + ds_read (2 per 4 mfma)
template <int VAR>
__global__ __launch_bounds__(256, 2) void k(const float* g, float* out, unsigned long long* cyc, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 16384; i += 256) ((float*)smem)[i] = g[i];
    __syncthreads();
    const unsigned lds0 = (unsigned)(size_t)((LDS_AS unsigned char*)smem);
    f32x16 acc0 = {0}, acc1 = {0};
    f32x4 a0 = *(f32x4*)(smem + tid * 16), a1 = *(f32x4*)(smem + 4096 + tid * 16), b = *(f32x4*)(smem + 8192 + tid * 16);
    const int ao = (tid & 63) * 128 + ((tid >> 6) << 4);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (VAR == 3 || VAR == 4 || VAR == 7 || VAR == 8) {
            glds16(g + tid * 4 + (it & 7) * 1024, lds0 + 32768 + wave * 1024);
            glds16(g + tid * 4 + 8192 + (it & 7) * 1024, lds0 + 32768 + 4096 + wave * 1024);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            f32x4 na0 = a0, na1 = a1, nb = b;
            if (VAR >= 2) {
                na0 = *(f32x4*)(smem + (ao ^ (s << 5)));
                if (VAR != 5) na1 = *(f32x4*)(smem + 8192 + (ao ^ (s << 5)));
                nb = *(f32x4*)(smem + 16384 + (ao ^ (s << 5)));
                if ((VAR == 4) && s == 3) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }
                if ((VAR == 6) && s == 3) { __syncthreads(); }                                   // barrier (+lgkmcnt(0)) only
                if ((VAR == 7) && s == 3) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }   // DMA wait only
                if ((VAR == 8) && s == 3) { asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory"); }  // raw barrier, no lgkmcnt drain
                if ((VAR == 9) && s == 3) { asm volatile("s_barrier" ::: "memory"); }
            }
            if (VAR == 10 || VAR == 11) {
                if (VAR == 11 && s < 2) glds16(g + tid * 4 + s * 8192 + (it & 7) * 1024, lds0 + 32768 + s * 4096 + wave * 1024);
                na0 = *(f32x4*)(smem + (ao ^ (s << 5)));
                na1 = *(f32x4*)(smem + 8192 + (ao ^ (s << 5)));
                nb = *(f32x4*)(smem + 16384 + (ao ^ (s << 5)));
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b.x, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, b.x, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b.y, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, b.y, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, b.z, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, b.z, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, b.w, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, b.w, acc1, 0, 0, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 5, 0);
                a0 = na0; a1 = na1; b = nb;
                continue;
            }
            __builtin_amdgcn_sched_barrier(0);
            if (VAR == 0 || VAR == 5) {
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b.x, acc0, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b.y, acc0, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, b.z, acc0, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, b.w, acc0, 0, 0, 0);
                }
            } else {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b.x, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, b.x, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b.y, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, b.y, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, b.z, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, b.z, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, b.w, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, b.w, acc1, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            a0 = na0; a1 = na1; b = nb;
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
    for (int i = 0; i < 16; ++i) out[(blockIdx.x * 256 + tid) * 32 + i] = acc0[i], out[(blockIdx.x * 256 + tid) * 32 + 16 + i] = acc1[i];
}

template <int VAR>
void run(const float* g, float* out, unsigned long long* cyc, const char* name, int grid = 256) {
    const int iters = 2000;
    hipFuncSetAttribute((const void*)k<VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k<VAR>, dim3(grid), dim3(256), 65536, 0, g, out, cyc, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(grid);
    name = (grid == 512) ? (std::string(name) + "  [2 WG/CU]").c_str() : name;
    hipMemcpy(h.data(), cyc, grid * 8, hipMemcpyDeviceToHost);
    double s = 0;
    for (auto v : h) s += (double)v;
    printf("%-52s %7.2f cycles per MFMA per wave (ideal %d)\n", name, s / grid / (iters * 32.0), grid == 512 ? 128 : 64);
}

int main() {
    float *g, *out;
    unsigned long long* cyc;
    hipMalloc(&g, 1 << 22);
    hipMalloc(&out, 512 * 256 * 32 * 4);
    hipMalloc(&cyc, 512 * 8);
    std::vector<float> h(1 << 20);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) / 1000.f - 0.5f;
    hipMemcpy(g, h.data(), 1 << 22, hipMemcpyHostToDevice);
    run<0>(g, out, cyc, "bare, 1 accumulator (dependent chain)");
    run<1>(g, out, cyc, "bare, 2 accumulators alternating");
    run<2>(g, out, cyc, "2 acc + 3 ds_read_b128 per 8 MFMA");
    run<3>(g, out, cyc, "  + 2 LDS-DMA per 32 MFMA");
    run<4>(g, out, cyc, "  + vmcnt(0)+barrier per 32 MFMA");
    run<5>(g, out, cyc, "1 acc + 2 ds_read_b128 per 8 MFMA");
    run<10>(g, out, cyc, "2 acc + 3 ds_read INTERLEAVED 1 per MFMA gap");
    run<11>(g, out, cyc, "  + 2 LDS-DMA per 32, at substep starts");
    run<6>(g, out, cyc, "VAR2 + __syncthreads per 32 MFMA");
    run<9>(g, out, cyc, "VAR2 + raw s_barrier per 32 MFMA");
    run<7>(g, out, cyc, "VAR3 + vmcnt(0) only per 32 MFMA");
    run<8>(g, out, cyc, "VAR3 + vmcnt(0)+raw s_barrier per 32");
    return 0;
}
