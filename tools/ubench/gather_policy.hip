// How fast can ONE workgroup per CU pull a Winograd tile's cold input halo into LDS, and does the cache-policy of the LDS-DMA matter?
// (round 4; follow-up to the timeline of k_conv_wino<1>: 9.8 k cycles of waiting for the first 8-channel chunk of a 34 x 18 halo)
// Every workgroup (256 threads, one per CU) gathers, tile after tile, the halo of a 32 x 16-pixel tile of a 256 x 256 x 32-channel f32
// frame (128 bytes per pixel) out of a 1.9 GB tensor, every tile read once (nothing comes from a cache), by `buffer_load_dwordx4 ... lds`, waits vmcnt(0), next tile:
//   PIECE 32   the first 8-channel chunk only: 612 pixels x 32 B (2 lanes per pixel, 5 instructions per thread)   -- what k_conv_wino<1> does
//   PIECE 128  all 32 channels: 612 pixels x 128 B (8 lanes per pixel, 20 instructions per thread)                  -- full lines
// with the cache-policy bits none / sc0 / sc1 / nt / sc0 sc1 / sc1 nt.  Prints cycles per tile and bytes per cycle and CU.
// build: hipcc --offload-arch=gfx950 -O3 -o gather_policy gather_policy.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef int i32x4 __attribute__((ext_vector_type(4)));
#define LDS_AS __attribute__((address_space(3)))
#define GLDS(NAME, MODS)                                                                                                       \
    __device__ __forceinline__ void NAME(unsigned voff, i32x4 rsrc, unsigned soff, unsigned lds) {                              \
        unsigned keep;                                                                                                          \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen" MODS " lds\n\ts_mov_b32 m0, %0" \
                     : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(soff), "s"(lds) : "memory");                                   \
    }
GLDS(g_none, "")
GLDS(g_sc0, " sc0")
GLDS(g_sc1, " sc1")
GLDS(g_nt, " nt")
GLDS(g_sc0sc1, " sc0 sc1")
GLDS(g_sc1nt, " sc1 nt")
template <int POL>
__device__ __forceinline__ void glds(unsigned voff, i32x4 rsrc, unsigned soff, unsigned lds) {
    if (POL == 0) g_none(voff, rsrc, soff, lds);
    else if (POL == 1) g_sc0(voff, rsrc, soff, lds);
    else if (POL == 2) g_sc1(voff, rsrc, soff, lds);
    else if (POL == 3) g_nt(voff, rsrc, soff, lds);
    else if (POL == 4) g_sc0sc1(voff, rsrc, soff, lds);
    else g_sc1nt(voff, rsrc, soff, lds);
}

template <int PIECE, int POL>
__global__ __launch_bounds__(256, 1) void k(const float* g, size_t g_bytes, int tiles, unsigned long long* cyc, float* out) {
    constexpr int LPP = PIECE / 16;                       // lanes per pixel: 2 | 8
    constexpr int PIECES = 612 * LPP;                     // 1224 | 4896
    constexpr int IT = (PIECES + 255) / 256;              // 5 | 20
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds0 = (unsigned)(size_t)((LDS_AS unsigned char*)smem);
    i32x4 rs;
    {
        const unsigned long long b = (unsigned long long)g;
        rs.x = __builtin_amdgcn_readfirstlane((int)(unsigned)b);
        rs.y = __builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32) & 0xFFFF);
        rs.z = __builtin_amdgcn_readfirstlane((int)(g_bytes > 0x7fffffffull ? 0x7fffffff : (unsigned)g_bytes));
        rs.w = 0x00020000;
    }
    unsigned hoff[IT];
    for (int it = 0; it < IT; ++it) {
        const int q = it * 256 + tid, p = q / LPP, hy = p / 18, hx = p - hy * 18;
        hoff[it] = (q < PIECES) ? (unsigned)((hy * 256 + hx) * 128 + (q % LPP) * 16) : 0x80000000u;
    }
    const unsigned frame_bytes = 256u * 256u * 128u;     // 8 MB
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int t = 0; t < tiles; ++t) {
                const unsigned idx = (unsigned)(t * 256 + blockIdx.x);           // 128 tile places per frame, frame after frame: nothing is read twice
        const unsigned frame = idx >> 7, ty = (idx >> 4) & 7u, tx = idx & 15u;
        const unsigned soff = __builtin_amdgcn_readfirstlane(frame * frame_bytes + (ty * 32u * 256u + tx * 16u) * 128u);
#pragma unroll
        for (int it = 0; it < IT; ++it) glds<POL>(hoff[it], rs, soff, lds0 + (unsigned)(it * 4096) + wave * 1024);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
    out[blockIdx.x * 256 + tid] = ((float*)smem)[tid];
}

// The same gather inside a tile's life as k_conv_wino<1> lives it: gather (timed) -> PAUSE cycles of not touching memory (the MFMA loop)
// -> STORE_KB of 16-byte stores (the epilogue: 64 KB of activations + 16 KB pooled) -> next tile.  Only the gather is timed.
// One workgroup PER TILE (as the product launches them), 100 KB of LDS so that one fits a CU; the gather in k_conv_wino<1>'s own piece order
// (even image columns of a halo row first, then the odd ones), then PAUSE, then the stores.  cyc[wg] = entry -> gather landed.
template <int PAUSE, int STORE_KB, bool EO>
__global__ __launch_bounds__(256, 1) void kf(const float* g, size_t g_bytes, float* sink, unsigned long long* cyc, float* out) {
    constexpr int IT = 5;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds0 = (unsigned)(size_t)((LDS_AS unsigned char*)smem);
    i32x4 rs;
    {
        const unsigned long long b = (unsigned long long)g;
        rs.x = __builtin_amdgcn_readfirstlane((int)(unsigned)b);
        rs.y = __builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32) & 0xFFFF);
        rs.z = __builtin_amdgcn_readfirstlane((int)(g_bytes > 0x7fffffffull ? 0x7fffffff : (unsigned)g_bytes));
        rs.w = 0x00020000;
    }
    unsigned hoff[IT];
    for (int it = 0; it < IT; ++it) {
        const int q = it * 256 + tid, p = q / 2, hy = p / 18, r = p - hy * 18;
        const int hx = EO ? ((r >= 9) ? 2 * (r - 9) + 1 : 2 * r) : r;
        hoff[it] = (q < 1224) ? (unsigned)((hy * 256 + hx) * 128 + (q % 2) * 16) : 0x80000000u;
    }
    const unsigned idx = blockIdx.x;
    const unsigned frame = idx >> 7, ty = (idx >> 4) & 7u, tx = idx & 15u;
    const unsigned soff = __builtin_amdgcn_readfirstlane(frame * (256u * 256u * 128u) + (ty * 32u * 256u + tx * 16u) * 128u);
#pragma unroll
    for (int it = 0; it < IT; ++it) glds<0>(hoff[it], rs, soff, lds0 + (unsigned)(it * 4096) + wave * 1024);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    __syncthreads();
    for (int i = 0; i < PAUSE / 8192; ++i) __builtin_amdgcn_s_sleep(127);
    typedef float f4 __attribute__((ext_vector_type(4)));
    if (STORE_KB > 0) {
        f4* dst = (f4*)(sink + ((size_t)idx * (STORE_KB * 256)));
        const f4 v = {(float)idx, 1.f, 2.f, 3.f};
#pragma unroll 4
        for (int i = 0; i < STORE_KB * 1024 / 16 / 256; ++i) dst[i * 256 + tid] = v;
    }
    if (tid == 0) cyc[idx] = t1 - t0;
    if (idx == 0) out[tid] = ((float*)smem)[tid];
}

template <int PAUSE, int STORE_KB, bool EO>
void runf(const float* g, size_t gb, float* sink, unsigned long long* cyc, float* out, const char* what) {
    const int grid = 25600, lds = 100 * 1024;
    (void)hipFuncSetAttribute((const void*)kf<PAUSE, STORE_KB, EO>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL((kf<PAUSE, STORE_KB, EO>), dim3(grid), dim3(256), lds, 0, g, gb, sink, cyc, out);
    (void)hipDeviceSynchronize();
    if (hipGetLastError() != hipSuccess) { printf("%s: launch failed\n", what); return; }
    std::vector<unsigned long long> h(grid);
    (void)hipMemcpy(h.data(), cyc, grid * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("%-100s entry -> landed %7.0f cycles (median of %d workgroups)\n", what, (double)h[grid / 2], grid);
}

template <int PAUSE, int STORE_KB>
__global__ __launch_bounds__(256, 1) void kp(const float* g, size_t g_bytes, float* sink, int tiles, unsigned long long* cyc, float* out) {
    constexpr int IT = 5;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds0 = (unsigned)(size_t)((LDS_AS unsigned char*)smem);
    i32x4 rs;
    {
        const unsigned long long b = (unsigned long long)g;
        rs.x = __builtin_amdgcn_readfirstlane((int)(unsigned)b);
        rs.y = __builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32) & 0xFFFF);
        rs.z = __builtin_amdgcn_readfirstlane((int)(g_bytes > 0x7fffffffull ? 0x7fffffff : (unsigned)g_bytes));
        rs.w = 0x00020000;
    }
    unsigned hoff[IT];
    for (int it = 0; it < IT; ++it) {
        const int q = it * 256 + tid, p = q / 2, hy = p / 18, hx = p - hy * 18;
        hoff[it] = (q < 1224) ? (unsigned)((hy * 256 + hx) * 128 + (q % 2) * 16) : 0x80000000u;
    }
    const unsigned frame_bytes = 256u * 256u * 128u;
    unsigned long long acc = 0;
    typedef float f4 __attribute__((ext_vector_type(4)));
    for (int t = 0; t < tiles; ++t) {
        const unsigned idx = (unsigned)(t * 256 + blockIdx.x);
        const unsigned frame = idx >> 7, ty = (idx >> 4) & 7u, tx = idx & 15u;
        const unsigned soff = __builtin_amdgcn_readfirstlane(frame * frame_bytes + (ty * 32u * 256u + tx * 16u) * 128u);
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
        for (int it = 0; it < IT; ++it) glds<0>(hoff[it], rs, soff, lds0 + (unsigned)(it * 4096) + wave * 1024);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        acc += __builtin_amdgcn_s_memtime() - t0;
        __syncthreads();
        for (int i = 0; i < PAUSE / 8192; ++i) __builtin_amdgcn_s_sleep(127);   // 127 x 64 cycles
        if (STORE_KB > 0) {
            f4* dst = (f4*)(sink + ((size_t)idx * (STORE_KB * 256)));   // STORE_KB KB per tile, tile after tile
            const f4 v = {(float)t, 1.f, 2.f, 3.f};
#pragma unroll 4
            for (int i = 0; i < STORE_KB * 1024 / 16 / 256; ++i) dst[i * 256 + tid] = v;
        }
    }
    if (tid == 0) cyc[blockIdx.x] = acc;
    out[blockIdx.x * 256 + tid] = ((float*)smem)[tid];
}

template <int PAUSE, int STORE_KB>
void runp(const float* g, size_t gb, float* sink, unsigned long long* cyc, float* out, const char* what) {
    const int tiles = 100, grid = 256, lds = 6 * 4096;
    hipLaunchKernelGGL((kp<PAUSE, STORE_KB>), dim3(grid), dim3(256), lds, 0, g, gb, sink, tiles, cyc, out);
    (void)hipDeviceSynchronize();
    if (hipGetLastError() != hipSuccess) { printf("%s: launch failed\n", what); return; }
    std::vector<unsigned long long> h(grid);
    (void)hipMemcpy(h.data(), cyc, grid * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("%-100s gather %7.0f cycles per tile (median workgroup)\n", what, (double)h[grid / 2] / tiles);
}

template <int PIECE, int POL>
void run(const float* g, size_t gb, unsigned long long* cyc, float* out, const char* what) {
    const int tiles = 100, grid = 256, lds = 20 * 4096 + 4096;   // 25 600 distinct tiles of the 30 208 the tensor holds
    (void)hipFuncSetAttribute((const void*)k<PIECE, POL>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    for (int rep = 0; rep < 1; ++rep) {   // (one pass: a second one would find the tensor's tail in the 256 MB MALL)
        hipLaunchKernelGGL((k<PIECE, POL>), dim3(grid), dim3(256), lds, 0, g, gb, tiles, cyc, out);
        (void)hipDeviceSynchronize();
    }
    if (hipGetLastError() != hipSuccess) { printf("%s: launch failed\n", what); return; }
    std::vector<unsigned long long> h(grid);
    (void)hipMemcpy(h.data(), cyc, grid * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double cpt = (double)h[grid / 2] / tiles;
    printf("%3d-byte pieces, %-8s %7.0f cycles per tile (median workgroup of 256, all CUs gathering) = %5.1f cycles per 128-byte line, %5.2f useful bytes per cycle and CU\n",
           PIECE, what, cpt, cpt / 612.0, 612.0 * PIECE / cpt);
}
int main() {
    const size_t gb = 1888ull << 20;   // 236 frames of 8 MB: 7 x the MALL, 59 x the L2s
    float *g, *out;
    unsigned long long* cyc;
    (void)hipMalloc(&g, gb); (void)hipMemset(g, 0, gb);
    (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&cyc, 25600 * 8);
    run<32, 0>(g, gb, cyc, out, "none");
    run<32, 1>(g, gb, cyc, out, "sc0");
    run<32, 2>(g, gb, cyc, out, "sc1");
    run<32, 3>(g, gb, cyc, out, "nt");
    run<32, 4>(g, gb, cyc, out, "sc0 sc1");
    run<32, 5>(g, gb, cyc, out, "sc1 nt");
    run<128, 0>(g, gb, cyc, out, "none");
    run<128, 1>(g, gb, cyc, out, "sc0");
    run<128, 2>(g, gb, cyc, out, "sc1");
    run<128, 3>(g, gb, cyc, out, "nt");
    run<128, 4>(g, gb, cyc, out, "sc0 sc1");
    run<128, 5>(g, gb, cyc, out, "sc1 nt");
    float* sink;
    (void)hipMalloc(&sink, (size_t)25600 * 80 * 1024);   // 2 GB of store targets
    runp<0, 0>(g, gb, sink, cyc, out, "32-byte pieces, gather only, timed inside the loop");
    runp<24576, 0>(g, gb, sink, cyc, out, "+ 24 k cycles without memory traffic between two gathers (the MFMA loop)");
    runp<0, 80>(g, gb, sink, cyc, out, "+ 80 KB of stores after every gather (the epilogue), no pause");
    runp<24576, 80>(g, gb, sink, cyc, out, "+ both: gather -> 24 k cycles -> 80 KB of stores -> next tile (a tile's life in k_conv_wino<1>)");
    runf<24576, 80, false>(g, gb, sink, cyc, out, "one workgroup per tile (100 KB LDS): gather -> 24 k cycles -> 80 KB of stores");
    runf<24576, 80, true>(g, gb, sink, cyc, out, "the same, pieces in k_conv_wino's order (even columns of a row, then odd)");
    runf<24576, 0, true>(g, gb, sink, cyc, out, "the same without the stores");
    runf<0, 0, true>(g, gb, sink, cyc, out, "the same without stores and pause (gather only, one workgroup per tile)");
    return 0;
}
