// Would a SECOND wave per SIMD hide k_conv_wino<1>'s non-MFMA work?  (round 4, follow-up to wino1_chunk_mix.hip)
// k_conv_wino runs one wave per SIMD because a wave owns all 16 Winograd positions of its 32 windows (256 accumulator registers).
// The alternative costed in DESIGN: 8 waves per workgroup, wave (wm, ph) owning HALF of the positions of the same 32 windows
// (128 accumulators -> two waves fit a SIMD), V and U shared through LDS exactly as today (no extra LDS bytes), the transform item
// of a thread halved by channels (two threads per (window, 4-channel) item: 8-byte LDS accesses, half the adds each).
// This loop issues k_conv_wino<1>'s 8-channel chunk in both shapes, on synthetic in-range addresses, with the same two barriers:
//   A  4 waves x 64 MFMA (16 accumulators)   -- the kernel's chunk, as wino1_chunk_mix MIX 4 / 5
//   B  8 waves x 32 MFMA ( 8 accumulators)   -- two waves per SIMD, half the side work each
// and prints cycles per chunk (4 096 = the SIMD's MFMAs alone).  No prologue / epilogue: this is the LOOP's ceiling only.
// build: hipcc --offload-arch=gfx950 -O3 -o wino1_two_waves_per_simd wino1_two_waves_per_simd.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
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
__device__ __forceinline__ f32x2 sub2(f32x2 a, f32x2 b) {
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ f32x4 ldsr(unsigned a) { return *(const LDS_AS f32x4*)(unsigned long long)a; }
__device__ __forceinline__ f32x2 ldsr2(unsigned a) { return *(const LDS_AS f32x2*)(unsigned long long)a; }
__device__ __forceinline__ void ldsw2(unsigned a, f32x2 v) { *(LDS_AS f32x2*)(unsigned long long)a = v; }
__device__ __forceinline__ void ldsw(unsigned a, f32x4 v) { *(LDS_AS f32x4*)(unsigned long long)a = v; }
constexpr int RAW = 19584, VB = 65536, UG = 8192, LDS_BYTES = RAW + VB + 2 * UG;

// WPS = waves per SIMD (1: the kernel's shape, 256 threads; 2: 512 threads).  COLD: halo streamed from a 480 MB tensor.
template <int WPS, bool COLD>
__global__ __launch_bounds__(256 * WPS, 1) void k(const float* g, size_t g_bytes, float* out, int chunks, unsigned long long* cyc) {
    constexpr int NACC = 16 / WPS;        // accumulators per wave
    constexpr int NST = 8 / WPS;          // MFMA steps (of 4) per group and wave
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 3, ph = wave >> 2;
    const int li = lane & 31, lh = lane >> 5;
    for (int i = tid; i < LDS_BYTES / 4; i += 256 * WPS) ((float*)smem)[i] = 1e-3f * (float)((i * 2654435761u) >> 20);
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
    const int wxa = 2 * (li >> 3) + ((li >> 2) & 1);
    unsigned abase = lds0 + RAW + (8 * (4 * wm + (wxa >> 1)) + (li & 3) + 4 * (wxa & 1)) * 32 + ((lh ^ ((wxa >> 1) & 1)) << 4) + ph * 4 * 4096;
    unsigned bbase = lds0 + RAW + VB + li * 32 + ((lh ^ ((li >> 3) & 1)) << 4) + ph * 4 * 1024;
    // transform item: WPS 1: (window, 4 channels) per thread, 16-byte accesses; WPS 2: (window, 2 channels), 8-byte accesses
    const int item = (WPS == 1) ? tid : (tid >> 1);
    const int qc = item & 1, wr = 4 * ((item >> 6) & 3) + ((item >> 1) & 3), wc = ((item >> 3) & 1) + 2 * ((item >> 4) & 3);
    const unsigned sub = (WPS == 1) ? 0u : (unsigned)((tid & 1) * 8);
    unsigned rbase = lds0 + (unsigned)((2 * wr * 18 + wc) * 32 + qc * 16) + sub;
    unsigned vwbase = lds0 + RAW + (unsigned)((8 * (4 * (wr >> 2) + (wc >> 1)) + (wr & 3) + 4 * (wc & 1)) * 32 + ((qc ^ ((wc >> 1) & 1)) << 4)) + sub;
    asm volatile("" : "+v"(abase), "+v"(bbase), "+v"(rbase), "+v"(vwbase));
    unsigned hoff[5];
    const int tile = blockIdx.x & 127, frame = blockIdx.x >> 7;
    for (int it = 0; it < 5; ++it) {
        const int t256 = tid & 255;
        const int p = (it * 256 + t256) >> 1, hy = p / 18, hx = p - hy * 18;
        const int gy = (tile >> 4) * 32 + hy, gx = (tile & 15) * 16 + hx;
        hoff[it] = (it * 256 + t256 < 1224 && gy < 256 && gx < 256) ? (unsigned)((gy * 256 + gx) * 128 + (t256 & 1) * 16) : 0x80000000u;
    }
    const unsigned frame_bytes = 256u * 256u * 128u;
    f32x16 acc[NACC];
    for (int q = 0; q < NACC; ++q) {
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
        asm volatile("" : "+a"(acc[q]));
    }
    // transform registers: f32x4 items (WPS 1) or f32x2 half items (WPS 2)
    f32x4 d4[16], t4[16], v4[16];
    f32x2 d2[16], t2[16], v2[16];
    for (int n = 0; n < 16; ++n) { d4[n] = t4[n] = v4[n] = f32x4{0.f, 0.f, 0.f, 0.f}; d2[n] = t2[n] = v2[n] = f32x2{0.f, 0.f}; }
    auto xf_op = [&](int n) {   // 0-15 raw reads, 16-31 row ops, 32-47 column ops, 48-55 lo V writes (hi writes 56-63 issued in the other group)
        if (WPS == 1) {
            if (n < 16) d4[n] = ldsr(rbase + (unsigned)(((n >> 2) * 18 + (n & 1) * 9 + ((n & 3) >> 1)) * 32));
            else if (n < 32) { const int r = (n - 16) >> 2, j = n & 3; t4[n - 16] = (r == 0) ? sub4(d4[j], d4[8 + j]) : (r == 1) ? d4[4 + j] + d4[8 + j] : (r == 2) ? sub4(d4[8 + j], d4[4 + j]) : sub4(d4[4 + j], d4[12 + j]); }
            else if (n < 48) { const int i = (n - 32) >> 2, cc = n & 3; v4[n - 32] = (cc == 0) ? sub4(t4[4 * i], t4[4 * i + 2]) : (cc == 1) ? t4[4 * i + 1] + t4[4 * i + 2] : (cc == 2) ? sub4(t4[4 * i + 2], t4[4 * i + 1]) : sub4(t4[4 * i + 1], t4[4 * i + 3]); }
            else if (n < 64) ldsw(vwbase + (unsigned)((n - 48) * 4096), v4[n - 48]);
        } else {
            if (n < 16) d2[n] = ldsr2(rbase + (unsigned)(((n >> 2) * 18 + (n & 1) * 9 + ((n & 3) >> 1)) * 32));
            else if (n < 32) { const int r = (n - 16) >> 2, j = n & 3; t2[n - 16] = (r == 0) ? sub2(d2[j], d2[8 + j]) : (r == 1) ? d2[4 + j] + d2[8 + j] : (r == 2) ? sub2(d2[8 + j], d2[4 + j]) : sub2(d2[4 + j], d2[12 + j]); }
            else if (n < 48) { const int i = (n - 32) >> 2, cc = n & 3; v2[n - 32] = (cc == 0) ? sub2(t2[4 * i], t2[4 * i + 2]) : (cc == 1) ? t2[4 * i + 1] + t2[4 * i + 2] : (cc == 2) ? sub2(t2[4 * i + 2], t2[4 * i + 1]) : sub2(t2[4 * i + 1], t2[4 * i + 3]); }
            else if (n < 64) ldsw2(vwbase + (unsigned)((n - 48) * 4096), v2[n - 48]);
        }
    };
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int c = 0; c < chunks; ++c) {
        const unsigned fsel = COLD ? (unsigned)((frame * 8 + (c >> 2)) % 60) * frame_bytes : 0u;
#pragma unroll
        for (int gq = 0; gq < 2; ++gq) {
            f32x4 fa[2], fb[2];
            fa[0] = ldsr(abase + (unsigned)(8 * gq * 4096));
            fb[0] = ldsr(bbase + (unsigned)(gq * UG));
#pragma unroll
            for (int st = 0; st < NST; ++st) {
                const int kq = NST * gq + st;
                const f32x4 av = fa[st & 1], bv = fb[st & 1];
                acc[kq] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc[kq], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (st + 1 < NST) {
                    fa[(st + 1) & 1] = ldsr(abase + (unsigned)((8 * gq + st + 1) * 4096));
                    fb[(st + 1) & 1] = ldsr(bbase + (unsigned)(gq * UG + (st + 1) * 1024));
                }
                __builtin_amdgcn_sched_barrier(0);
                acc[kq] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc[kq], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                // side work of this step: 64 transform micro-ops per chunk and thread over the 2 x NST steps, half per slot
                {
                    constexpr int per = 64 / (2 * NST) / 2;   // 2 (WPS 1) | 4 (WPS 2) micro-ops per slot
                    const int base = (gq * NST + st) * 2 * per;
#pragma unroll
                    for (int u = 0; u < per; ++u) xf_op(base + u);
                }
                __builtin_amdgcn_sched_barrier(0);
                acc[kq] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc[kq], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (st < 2 && (WPS == 1 || ph == (st & 1)))   // next U group: 2 pieces per group and 256 threads
                    glds16b((unsigned)(tid & 255) * 16u, rs, (unsigned)(((2 * c + gq) & 63) * UG + st * 4096), lds0 + RAW + VB + (gq ^ 1) * UG + st * 4096 + wm * 1024);
                {
                    constexpr int per = 64 / (2 * NST) / 2;
                    const int base = (gq * NST + st) * 2 * per + per;
#pragma unroll
                    for (int u = 0; u < per; ++u) xf_op(base + u);
                }
                __builtin_amdgcn_sched_barrier(0);
                acc[kq] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc[kq], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (gq == 0) {   // raw halo of the next chunk: 5 pieces per 256 threads
                    if (WPS == 1) { if (st < 5) glds16b(hoff[st], rs, fsel + (unsigned)((c & 3) * 32), lds0 + st * 4096 + wm * 1024); }
                    else {
                        const int pc = 2 * st + ph;   // the two waves of a SIMD alternate
                        const unsigned ho = (ph == 0) ? hoff[2 * st < 5 ? 2 * st : 0] : hoff[2 * st + 1 < 5 ? 2 * st + 1 : 0];
                        if (pc < 5) glds16b(ho, rs, fsel + (unsigned)((c & 3) * 32), lds0 + pc * 4096 + wm * 1024);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
    float s = 0.f;
    for (int q = 0; q < NACC; ++q) for (int r = 0; r < 16; ++r) s += acc[q][r];
    out[(size_t)blockIdx.x * 256 * WPS + tid] = s + v4[3].x + v2[3].x;
}
template <int WPS, bool COLD>
void run(const float* g, size_t gb, float* out, unsigned long long* cyc, const char* what) {
    const int chunks = 400, grid = 256;
    hipFuncSetAttribute((const void*)k<WPS, COLD>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k<WPS, COLD>), dim3(grid), dim3(256 * WPS), LDS_BYTES, 0, g, gb, out, chunks, cyc);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    if (hipGetLastError() != hipSuccess) { printf("launch failed\n"); return; }
    std::vector<unsigned long long> h(grid);
    hipMemcpy(h.data(), cyc, grid * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double cpc = (double)h[grid / 2] / chunks;
    printf("%-86s %7.0f cycles per chunk (median workgroup) -> MFMA busy %.3f | wall %.3f ms -> %.1f TFLOP/s executed\n", what, cpc, 4096.0 / cpc, ms,
           2.0 * 64 * 2048 * 4.0 * grid * chunks / (ms * 1e-3) / 1e12);
}
int main() {
    const size_t gb = 480ull << 20;
    float *g, *out;
    unsigned long long* cyc;
    hipMalloc(&g, gb); hipMemset(g, 0, gb);
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8);
    run<1, false>(g, gb, out, cyc, "A  one wave per SIMD: 4 waves x 64 MFMA, k_conv_wino<1>'s chunk (L2-resident halo)");
    run<2, false>(g, gb, out, cyc, "B  two waves per SIMD: 8 waves x 32 MFMA, half the side work each (L2-resident halo)");
    run<1, true>(g, gb, out, cyc, "A  one wave per SIMD, halo streamed cold from a 480 MB tensor");
    run<2, true>(g, gb, out, cyc, "B  two waves per SIMD, halo streamed cold from a 480 MB tensor");
    return 0;
}
