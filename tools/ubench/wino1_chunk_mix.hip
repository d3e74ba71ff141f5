// Ceiling of k_conv_wino<1>'s 8-CHANNEL-CHUNK instruction mix on one wave per SIMD.
// k_conv_wino<1> (the 32-column layers d0b / u7a / u7b: 25 % of the 64-frame chain) executes 0.47 of the f32 MFMA peak; its main loop
// 0.67-0.79.  This loop issues, per wave and chunk, exactly what that kernel's main loop issues (og_kernels.hpp: k_conv_wino, NT = 1) --
//   64 x v_mfma_f32_32x32x2_f32 on 16 accumulators (256 registers: ONE workgroup per CU, one wave per SIMD),
//   32 fragment ds_read_b128, 16 raw-halo ds_read_b128, 16 V ds_write_b128, 128 vector adds (B^T d B on four channels),
//   9 LDS-DMA pieces (4 of transformed weights: contiguous KB; 5 of raw halo: 64 lanes x 16 B gathered from 32 cache lines),
//   2 x (s_waitcnt vmcnt(0) + s_barrier)
// -- in the same slots between the MFMAs, on synthetic in-range addresses, with no tile prologue / epilogue around it, and adds the
// ingredients one at a time.  Prints cycles per chunk (4 096 = the MFMAs alone) and the MFMA-busy fraction.
// build: hipcc --offload-arch=gfx950 -O3 -o wino1_chunk_mix wino1_chunk_mix.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
#define LDS_AS __attribute__((address_space(3)))
__device__ __forceinline__ void glds16b(unsigned voff, i32x4 rsrc, unsigned soff, unsigned lds) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(soff), "s"(lds) : "memory");
}
// a - b as two v_pk_add_f32 with the negate modifier (hipcc lowers a vector subtraction to four v_sub_f32): what the kernels do since round 4
__device__ __forceinline__ f32x4 sub4(f32x4 a, f32x4 b) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 lo, hi;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(lo) : "v"(f2{a.x, a.y}), "v"(f2{b.x, b.y}));
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(hi) : "v"(f2{a.z, a.w}), "v"(f2{b.z, b.w}));
    return f32x4{lo.x, lo.y, hi.x, hi.y};
}
__device__ __forceinline__ f32x4 ldsr(unsigned a) { return *(const LDS_AS f32x4*)(unsigned long long)a; }
__device__ __forceinline__ void ldsw(unsigned a, f32x4 v) { *(LDS_AS f32x4*)(unsigned long long)a = v; }
constexpr int RAW = 19584, VB = 65536, UG = 8192, LDS_BYTES = RAW + VB + 2 * UG;
// MIX 0 MFMA only | 1 + fragment reads | 2 + transform (raw reads, adds, V writes) | 3 + LDS-DMA (source resident in L2) + vmcnt(0)
//     4 + barriers = the full mix | 5 = 4 with the raw halo streamed from a 512 MB tensor (as the first 32-column layer's input is)
template <int MIX>
__global__ __launch_bounds__(256, 1) void k(const float* g, size_t g_bytes, float* out, int chunks, unsigned long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    for (int i = tid; i < LDS_BYTES / 4; i += 256) ((float*)smem)[i] = 1e-3f * (float)((i * 2654435761u) >> 20);
    __syncthreads();
    const unsigned lds0 = (unsigned)(size_t)((LDS_AS unsigned char*)smem);
    i32x4 rs;
    {
        const unsigned long long b = (unsigned long long)g;
        rs.x = __builtin_amdgcn_readfirstlane((int)(unsigned)b);
        rs.y = __builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32) & 0xFFFF);
        rs.z = __builtin_amdgcn_readfirstlane((int)(g_bytes > 0x7fffffffull ? 0x7fffffff : (unsigned)g_bytes));
        rs.w = 0x00020000;
    }
    // fragment / transform addresses as in k_conv_wino<1> (conflict-free layouts)
    const int wxa = 2 * (li >> 3) + ((li >> 2) & 1);
    unsigned abase = lds0 + RAW + (8 * (4 * wave + (wxa >> 1)) + (li & 3) + 4 * (wxa & 1)) * 32 + ((lh ^ ((wxa >> 1) & 1)) << 4);
    unsigned bbase = lds0 + RAW + VB + li * 32 + ((lh ^ ((li >> 3) & 1)) << 4);
    const int qc = tid & 1, wr = 4 * (tid >> 6) + ((tid >> 1) & 3), wc = ((tid >> 3) & 1) + 2 * ((tid >> 4) & 3);
    unsigned rbase = lds0 + (unsigned)((2 * wr * 18 + wc) * 32 + qc * 16);
    unsigned vwbase = lds0 + RAW + (unsigned)((8 * (4 * (wr >> 2) + (wc >> 1)) + (wr & 3) + 4 * (wc & 1)) * 32 + ((qc ^ ((wc >> 1) & 1)) << 4));
    asm volatile("" : "+v"(abase), "+v"(bbase), "+v"(rbase), "+v"(vwbase));
    // raw-halo gather: piece = it * 256 + tid -> pixel (34 x 18 tile at this workgroup's place in a 256 x 256 x 32-channel frame), 32-byte pieces
    unsigned hoff[5];
    const int tile = blockIdx.x & 127, frame = blockIdx.x >> 7;
    for (int it = 0; it < 5; ++it) {
        const int p = (it * 256 + tid) >> 1, hy = p / 18, hx = p - hy * 18;
        const int gy = (tile >> 4) * 32 + hy, gx = (tile & 15) * 16 + hx;
        hoff[it] = (it * 256 + tid < 1224 && gy < 256 && gx < 256) ? (unsigned)((gy * 256 + gx) * 128 + (tid & 1) * 16) : 0x80000000u;
    }
    const unsigned frame_bytes = 256u * 256u * 128u;
    f32x16 acc[16];
    for (int q = 0; q < 16; ++q) {
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
        asm volatile("" : "+a"(acc[q]));
    }
    f32x4 d[16], t[16], tv[16];
    for (int n = 0; n < 16; ++n) d[n] = t[n] = tv[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto xf_op = [&](int n) {
        if (n < 16) d[n] = ldsr(rbase + (unsigned)(((n >> 2) * 18 + (n & 1) * 9 + ((n & 3) >> 1)) * 32));
        else if (n < 32) { const int r = (n - 16) >> 2, j = n & 3; t[n - 16] = (r == 0) ? sub4(d[j], d[8 + j]) : (r == 1) ? d[4 + j] + d[8 + j] : (r == 2) ? sub4(d[8 + j], d[4 + j]) : sub4(d[4 + j], d[12 + j]); }
        else if (n < 48) { const int i = (n - 32) >> 2, cc = n & 3; tv[n - 32] = (cc == 0) ? sub4(t[4 * i], t[4 * i + 2]) : (cc == 1) ? t[4 * i + 1] + t[4 * i + 2] : (cc == 2) ? sub4(t[4 * i + 2], t[4 * i + 1]) : sub4(t[4 * i + 1], t[4 * i + 3]); }
        else if (n < 56) ldsw(vwbase + (unsigned)((n - 48) * 4096), tv[n - 48]);
    };
    auto xf4 = [&](int n) { xf_op(n); xf_op(n + 1); xf_op(n + 2); xf_op(n + 3); };
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int c = 0; c < chunks; ++c) {
        const unsigned fsel = (MIX == 5) ? (unsigned)((frame * 8 + (c >> 2)) % 60) * frame_bytes : 0u;   // MIX 5: another frame of a 480 MB tensor every four chunks
#pragma unroll
        for (int gq = 0; gq < 2; ++gq) {
            f32x4 fa[2], fb[2];
            fa[0] = ldsr(abase + (unsigned)(8 * gq * 4096));
            fb[0] = ldsr(bbase + (unsigned)(gq * UG));
#pragma unroll
            for (int st = 0; st < 8; ++st) {
                const int kq = 8 * gq + st;
                const f32x4 av = fa[st & 1], bv = fb[st & 1];
                acc[kq] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc[kq], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (MIX >= 1 && st + 1 < 8) {
                    fa[(st + 1) & 1] = ldsr(abase + (unsigned)((8 * gq + st + 1) * 4096));
                    fb[(st + 1) & 1] = ldsr(bbase + (unsigned)(gq * UG + (st + 1) * 1024));
                }
                __builtin_amdgcn_sched_barrier(0);
                acc[kq] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc[kq], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (MIX >= 2) {
                    if (gq == 0) ldsw(vwbase + (unsigned)((8 + st) * 4096), tv[8 + st]);
                    else if (st < 2) xf4(8 * st);
                    else if (st < 4) xf4(16 + 8 * (st - 2));
                    else if (st < 6) xf4(32 + 8 * (st - 4));
                    else { xf_op(48 + 4 * (st - 6)); xf_op(48 + 4 * (st - 6) + 1); }
                }
                __builtin_amdgcn_sched_barrier(0);
                acc[kq] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc[kq], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (MIX >= 3 && st < 2) glds16b((unsigned)tid * 16u, rs, (unsigned)(((2 * c + gq) & 63) * UG + st * 4096), lds0 + RAW + VB + (gq ^ 1) * UG + st * 4096 + wave * 1024);
                if (MIX >= 2 && gq == 1) {
                    if (st < 2) xf4(8 * st + 4);
                    else if (st < 4) xf4(16 + 8 * (st - 2) + 4);
                    else if (st < 6) xf4(32 + 8 * (st - 4) + 4);
                    else { xf_op(48 + 4 * (st - 6) + 2); xf_op(48 + 4 * (st - 6) + 3); }
                }
                __builtin_amdgcn_sched_barrier(0);
                acc[kq] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc[kq], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (MIX >= 3 && gq == 0 && st < 5) glds16b(hoff[st], rs, fsel + (unsigned)((c & 3) * 32), lds0 + st * 4096 + wave * 1024);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (MIX >= 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (MIX >= 4) __syncthreads();
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
    float s = 0.f;
    for (int q = 0; q < 16; ++q) for (int r = 0; r < 16; ++r) s += acc[q][r];
    out[(size_t)blockIdx.x * 256 + tid] = s + tv[3].x;
}
template <int MIX>
void run(const float* g, size_t gb, float* out, unsigned long long* cyc, const char* what) {
    const int chunks = 400, grid = 256;
    hipFuncSetAttribute((const void*)k<MIX>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k<MIX>, dim3(grid), dim3(256), LDS_BYTES, 0, g, gb, out, chunks, cyc);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    std::vector<unsigned long long> h(grid);
    hipMemcpy(h.data(), cyc, grid * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double cpc = (double)h[grid / 2] / chunks;
    printf("MIX %d  %-78s %7.0f cycles per chunk (median workgroup) -> MFMA busy %.3f | wall %.3f ms -> %.1f TFLOP/s executed\n", MIX, what, cpc,
           4096.0 / cpc, ms, 2.0 * 64 * 2048 * 4.0 * grid * chunks / (ms * 1e-3) / 1e12);
}
#include <algorithm>
int main() {
    const size_t gb = 480ull << 20;
    float *g, *out;
    unsigned long long* cyc;
    hipMalloc(&g, gb); hipMemset(g, 0, gb);
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
    run<0>(g, gb, out, cyc, "64 MFMA on 16 accumulators");
    run<1>(g, gb, out, cyc, "+ 32 fragment ds_read_b128");
    run<2>(g, gb, out, cyc, "+ transform: 16 raw ds_read_b128, 128 adds, 16 V ds_write_b128");
    run<3>(g, gb, out, cyc, "+ 9 LDS-DMA pieces (4 weights, 5 gathered halo, L2-resident source) + 2 vmcnt(0)");
    run<4>(g, gb, out, cyc, "+ 2 barriers = k_conv_wino<1>'s chunk");
    run<5>(g, gb, out, cyc, "the same, halo streamed from a 480 MB tensor (cold lines)");
    return 0;
}
