// Hand-written gfx950 (CDNA4, wave64) kernels for the U-Net forward path.
//
// Data layout in HBM: activations are NHWC float32 with the channel count
// padded to a multiple of 32 ("Cp"); a decoder level's skip and up-sampled
// tensors live in ONE buffer of 2*Cp channels per pixel (skip in [0,Cp), up in
// [Cp,2Cp)), so torch.cat([skip, x]) (unet.py:86) never materialises: the
// encoder conv and the transposed conv write straight into their halves.
//
// The 3x3 convolutions (95 % of the FLOPs) and the 2x2/s2
// transposed convolutions run as im2col-free implicit GEMMs on the exact-f32
// matrix pipe: v_mfma_f32_32x32x2_f32 (64 FLOP/clk/SIMD, bit-identical to an
// fmaf chain, MI355X_MICROARCH.md "Matrix cores").  A = pixels (rows) x input
// channels, B = input channels x output channels, staged through LDS by LDS-DMA
// (buffer_load_dwordx4 ... lds in the default kernel, global_load_lds_dwordx4 in
// the older variants) with an XOR-swizzled 128-B-row image so the ds_read_b128
// fragment reads are bank-conflict-free.
//
// Kernel generations, all bit-identical per output element (tests pin that):
//   k_conv_mfma    gen 1, one tile per workgroup, double-buffered halo (reference variant, conv_impl 0)
//   k_conv_mfma_p  persistent workgroups, cross-tile prefetch, split-K (launches below one workgroup per CU)
//   k_conv_mfma_o  occupancy variant = the default: no vector-ALU instruction in the MFMA loop, lean epilogue
//                  (conv_epilogue_b), MODE 0 3x3 / 1 transposed / 2 1x1 / 3 3x3 stride 2, fused first layer and head,
//                  split-K with fused reduce.  DESIGN.md section 4 has the measurements behind each choice.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define OG_GLOBAL_AS __attribute__((address_space(1)))
#define OG_LDS_AS __attribute__((address_space(3)))

// 16-byte LDS-DMA (global_load_lds_dwordx4): per-lane global source, wave-uniform LDS byte
// address in M0 (+ lane*16 by hardware).  Written as inline asm ON PURPOSE: when hipcc sees the
// LDS-DMA builtin inside a loop it degrades every later `s_waitcnt lgkmcnt(N)` of that loop to
// lgkmcnt(0), which exposes the ds_read latency of the register-prefetched MFMA fragments
// (measured: 86 % -> see DESIGN.md).  The price: hipcc no longer tracks these transfers, so
// every consumer barrier is preceded by an explicit og_wait_dma() (cdna guide §5.7 item 1).
__device__ __forceinline__ void glds16(const float* gsrc, unsigned lds_wave_base) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc), "s"(lds_wave_base)
        : "memory");
}
// The same transfer addressed through a buffer resource: 32-bit per-lane byte offset + wave-uniform byte offset
// in an SGPR.  Two things the flat form cannot do: (1) the per-chunk / per-step advance lives in the SGPR, so a
// stage costs no vector ALU instruction at all (each one takes matrix-pipe time: tools/ubench/mfma_valu_coexec);
// (2) a lane whose offset is >= num_records reads zeros, which is the conv's zero padding without a zero page
// and without a per-lane select.
typedef float f32x2 __attribute__((ext_vector_type(2)));
// two fused multiply-adds in one vector-ALU instruction (v_pk_fma_f32); each component is an ordinary fmaf
__device__ __forceinline__ f32x2 og_fma2s(f32x2 a, f32x2 b, f32x2 c) { return f32x2{fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y)}; }
__device__ __forceinline__ f32x2 og_fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
#define og_fma2_first og_fma2
typedef int og_i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ og_i32x4 og_make_rsrc(const void* base, unsigned bytes) {
    const unsigned long long b = (unsigned long long)base;
    og_i32x4 r;
    r.x = (int)(unsigned)b;
    r.y = (int)(unsigned)(b >> 32) & 0xFFFF;  // stride 0: raw buffer
    r.z = (int)bytes;                          // num_records (bytes)
    r.w = 0x00020000;                          // DATA_FORMAT 32, untyped access
    r.x = __builtin_amdgcn_readfirstlane(r.x);
    r.y = __builtin_amdgcn_readfirstlane(r.y);
    r.z = __builtin_amdgcn_readfirstlane(r.z);
    return r;
}
constexpr unsigned OG_OOB = 0x80000000u;  // a lane offset no buffer of < 2 GiB contains
__device__ __forceinline__ void glds16b(unsigned voff, og_i32x4 rsrc, unsigned soff, unsigned lds_wave_base) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %4\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %1, %2, %3 offen lds\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(rsrc), "s"(soff), "s"(lds_wave_base)
        : "memory");
}
__device__ __forceinline__ float og_act(float v, int act) {
    if (act == 1) return fmaxf(v, 0.f);
    if (act == 2) return v / (1.0f + expf(-v));  // SiLU = x * sigmoid(x)
    return v;
}
__device__ __forceinline__ void og_wait_dma() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// 16-byte LDS read from an absolute LDS byte address (no symbol-relative add: hipcc materialises `smem + x` as
// "0 + x" per access once the control flow keeps it from hoisting that add)
// a - b on four floats as TWO packed instructions: hipcc lowers a vector subtraction to four v_sub_f32 (an addition to two
// v_pk_add_f32); v_pk_add_f32 with the negate modifier on the second source is the same IEEE subtraction, bit for bit
__device__ __forceinline__ f32x4 og_sub4(f32x4 a, f32x4 b) {
    f32x2 lo, hi;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(lo) : "v"(f32x2{a.x, a.y}), "v"(f32x2{b.x, b.y}));
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(hi) : "v"(f32x2{a.z, a.w}), "v"(f32x2{b.z, b.w}));
    return f32x4{lo.x, lo.y, hi.x, hi.y};
}
__device__ __forceinline__ f32x4 og_add4(f32x4 a, f32x4 b) {   // (hipcc packs most vector additions itself, not all)
    f32x2 lo, hi;
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(lo) : "v"(f32x2{a.x, a.y}), "v"(f32x2{b.x, b.y}));
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(hi) : "v"(f32x2{a.z, a.w}), "v"(f32x2{b.z, b.w}));
    return f32x4{lo.x, lo.y, hi.x, hi.y};
}
__device__ __forceinline__ f32x4 og_lds_read16(unsigned addr) { return *(const OG_LDS_AS f32x4*)(unsigned long long)addr; }
__device__ __forceinline__ unsigned og_lds_addr(const void* p) {
    return (unsigned)(size_t)((OG_LDS_AS const unsigned char*)p);
}

// ---- split precision (opt-in "precision" 1, never the default): an f32 value v travels as TWO f16 numbers,
//   hi = f16(v),  lo = f16((v - hi) * 2^11)      =>   v = hi + lo * 2^-11  to ~2^-22 relative,
// and a product a*b is taken as  a_hi*b_hi + 2^-11 * (a_hi*b_lo + a_lo*b_hi)  on v_mfma_f32_32x32x16_f16 with f32
// accumulation (two accumulators, combined once in the epilogue): 3 f16 MFMAs of 32 cycles for 16 k, against 8 f32
// MFMAs of 64 cycles.  The 2^11 scale keeps `lo` out of the f16 subnormals (the MFMA does keep subnormal inputs --
// tools/ubench/mfma_f16_split -- but they carry fewer bits).  Emulated on the CPU against the reference fixture: max
// |dlogit| 1.4e-5, the same as plain f32 re-association noise (1.3e-5); tolerance 5e-5.  |v| must stay below 65504.
// "H layout" of a 32-channel chunk of one pixel (128 bytes, as the f32 layout): eight 16-byte slots, slot s < 4 = hi of
// channels 8s..8s+7, slot 4+s = lo of the same channels -- so the LDS images, the DMA staging, the swizzles and every
// store address of the f32 kernels carry over unchanged; only fragment contents and the MFMA differ.
typedef _Float16 og_h8 __attribute__((ext_vector_type(8)));
constexpr float OG_LO_SCALE = 2048.0f, OG_LO_INV = 1.0f / 2048.0f;
constexpr float OG_H_MAX = 60000.0f;   // f16 tops out at 65504: an activation beyond this makes the split-precision result worthless
// one word in host-mapped memory per handle: the epilogue raises it (rare path: one atomic per offending wave), the host
// checks it at the next synchronisation and fails the call loudly instead of returning saturated / inf garbage
__device__ __forceinline__ void og_flag_range(float vmax_abs, int* flag) {
    if (flag != nullptr && __ballot(vmax_abs > OG_H_MAX) != 0ull) {
        if ((threadIdx.x & 63) == 0) atomicOr(flag, 1);
    }
}
__device__ __forceinline__ void og_split(float v, _Float16& hi, _Float16& lo) {
    hi = (_Float16)v;
    lo = (_Float16)((v - (float)hi) * OG_LO_SCALE);
}
__device__ __forceinline__ f32x4 og_pack8(const _Float16* h) {
    og_h8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = h[e];
    return __builtin_bit_cast(f32x4, v);
}

struct ConvArgs {
    const float* in;        // NHWC, frame-major
    long long in_frame_stride;   // floats per frame
    int in_pix_stride;      // floats per pixel (total channels of the buffer)
    int in_ch_off;          // first channel this layer reads
    int n_chunks;           // padded Cin / 32
    int H, W;               // input spatial size (== output size for the 3x3 conv)
    int tiles_x, tiles_y;   // 16-wide x TH-high tiles per frame
    int n_spatial;          // B * tiles_x * tiles_y
    const float* wpk;       // packed weights, see pack_conv_weights() in og_api.hip
    const float* scale;     // [aff_mod] folded BN scale (1 for convT)
    const float* shift;     // [aff_mod] folded BN shift (bias for convT)
    int aff_mod;            // padded Cout
    float* out;
    long long out_frame_stride;
    int out_pix_stride;
    int out_ch_off;
    float* pool;            // optional fused MaxPool2d(2,2) output (unet.py:59,79), or nullptr
    long long pool_frame_stride;
    int pool_pix_stride;
    int pool_ch_off;
    const float* zero_page; // >= 128 B of zeros: source for the zero padding (padding=1, unet.py:24)
    int act;                // 0 none, 1 ReLU (U-Net), 2 SiLU (YOLOv8 Conv block)
    const float* res;       // optional residual added AFTER the activation (YOLOv8 Bottleneck shortcut), or nullptr
    long long res_frame_stride;
    int res_pix_stride;
    int res_ch_off;
    // Fused first layer (k_conv_mfma_o<..., FIRST=true> only): the 32-channel halo of downs.0's SECOND conv is
    // computed in-kernel from the u8 frame (/255, Conv2d(1,32,3), BN, ReLU: k_conv_first's arithmetic in the same
    // order) instead of being staged from HBM.
    const uint8_t* first_u8;
    const float* first_w9;      // [9][32]
    const float* first_scale;   // [32]
    const float* first_shift;   // [32]
    // Fused head (last U-Net conv only, Cout_p == 32): Conv2d(f0,1,1)+bias -> sigmoid -> > thr -> mask / area,
    // i.e. k_head's arithmetic in the same order, applied to the tile while it sits in the LDS scratch.
    const float* head_w;    // nullptr = no fusion
    float head_bias, head_thr;
    const int32_t* head_boxes;
    float* head_logits;
    uint8_t* head_mask;
    int32_t* head_area;
    int head_store_act;     // also store the activation tensor (parity/debug taps)
    int* range_flag;        // split precision only: set to 1 when an activation leaves the f16 range (|v| > OG_H_MAX); host-mapped, or nullptr
    int zdiv;               // k_conv_mfma_o's grid.z: zdiv = column tiles * ksplit (see the decode in the kernel)
    float zrcp;             // 1 / (zdiv << zgroup_shift)
    int zgroup_shift;       // log2 of the frames per z group
    int frames;             // B
    int vsplit;             // k_conv_mfma_o<..., VS = true> only: > 1 = sum the K parts of a `vsplit`-way split IN REGISTERS, in split order
                            // (bit-identical to ksplit = vsplit with the fused reduce): the detector's batched path (og_yolo.inc)
    int ksplit;             // >1: split-K.  Item = (tile, K-range); raw accumulators go to `partial`, and
    float* partial;         // k_splitk_epilogue sums them in split order and runs the epilogue (small-batch latency mode)
    int* tile_counter;      // k_conv_mfma_o split-K, fused reduce: arrivals per tile (zero between launches); nullptr = separate epilogue kernel
    int prio_mode;          // 0 off; 1/2: k_conv_mfma_p alternates s_setprio per unit, role = upper half of the grid / odd block;
                            // 3: as 2, and k_conv_mfma_o runs its set-up and epilogue at raised priority
    unsigned long long* stamps;  // diagnostic only (nullptr in production): per workgroup
                                 // k_conv_mfma_p: 4 x u64 {s_memtime, s_memrealtime} at entry and exit -> in-kernel clock
                                 // k_conv_mfma_o: 8 x u64 {entry, prologue done, main loop done, stores done, HW_ID, XCC_ID}
};

// MODE 0: 3x3 conv, pad 1, stride 1  (+ per-channel affine, ReLU, optional 2x2 max-pool)
// MODE 1: 2x2 stride-2 transposed conv as one GEMM with N = 4*Cout (dy,dx,co), scattered store
// MODE 2: 1x1 conv (detector): no halo border, one tap, plain epilogue
// MODE 3 (k_conv_mfma_o only): 3x3 stride-2 pad-1 conv (YOLOv8 down-sampling Conv) as a 2x2 stride-1 conv over the
//         space-to-depth view of the input; a.H, a.W are the OUTPUT size (input = 2H x 2W), a.n_chunks = 4 * Cin_p / 32
// NT    : 32-column output sub-tiles per workgroup (Cout tile = 32*NT); waves are laid out (4/NT) x NT
// TH    : tile height in pixels (tile is TH x 16)
template <int NT, int MODE, int TH>
__global__ __launch_bounds__(256, 2) void k_conv_mfma(ConvArgs a) {
    constexpr int TW = 16;
    constexpr int PAD = (MODE == 0) ? 1 : 0;
    constexpr int HW_ = TW + 2 * PAD;
    constexpr int HH_ = TH + 2 * PAD;
    constexpr int HALO_PIX = HW_ * HH_;
    constexpr int HALO_BYTES = HALO_PIX * 128;
    constexpr int HALO_PIECES = HALO_PIX * 8;
    constexpr int HALO_IT = (HALO_PIECES + 255) / 256;
    constexpr int TAPS = (MODE == 0) ? 9 : 1;
    constexpr int WROWS = 32 * NT;
    constexpr int WBYTES = WROWS * 128;
    constexpr int WM = 4 / NT;
    constexpr int MS = (TH / 2) / WM;  // 32-row M sub-tiles (2 pixel rows x 16) per wave
    static_assert(MS >= 1, "tile too small");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const halo0 = smem;
    unsigned char* const wbuf0 = smem + 2 * HALO_BYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave % NT;
    const int wm = wave / NT;
    const int li = lane & 31;
    const int lh = lane >> 5;

    // ---- tile decode (scalar) ----
    const int n_tile = blockIdx.x / a.n_spatial;
    int sp = blockIdx.x - n_tile * a.n_spatial;
    const int tiles_per_frame = a.tiles_x * a.tiles_y;
    const int b = sp / tiles_per_frame;
    sp -= b * tiles_per_frame;
    const int tyi = sp / a.tiles_x;
    const int ty0 = tyi * TH;
    const int tx0 = (sp - tyi * a.tiles_x) * TW;

    const float* in_frame = a.in + (long long)b * a.in_frame_stride + a.in_ch_off;

    // ---- per-thread halo source pointers (fixed across channel chunks) ----
    const float* hsrc[HALO_IT];
    int hstep[HALO_IT];
#pragma unroll
    for (int it = 0; it < HALO_IT; ++it) {
        const int q = it * 256 + tid;
        const int p = q >> 3;
        const int logical = (q & 7) ^ ((p >> 1) & 7);
        const int hy = p / HW_;
        const int hx = p - hy * HW_;
        const int gy = ty0 + hy - PAD, gx = tx0 + hx - PAD;
        const bool inb = (q < HALO_PIECES) && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        hsrc[it] = inb ? in_frame + ((long long)gy * a.W + gx) * a.in_pix_stride + logical * 4 : a.zero_page + logical * 4;
        hstep[it] = inb ? 32 : 0;
    }
    const bool last_valid = ((HALO_IT - 1) * 256 + tid) < HALO_PIECES;

    const unsigned lds0 = og_lds_addr(smem);
    auto stage_halo = [&](int buf, int c) {
        const unsigned base = lds0 + buf * HALO_BYTES + wave * 1024;
#pragma unroll
        for (int it = 0; it < HALO_IT; ++it) {
            if (it < HALO_IT - 1 || last_valid) glds16(hsrc[it] + c * hstep[it], base + it * 4096);
        }
    };
    const float* wtile = a.wpk + (long long)n_tile * a.n_chunks * TAPS * (WROWS * 32);
    auto stage_w = [&](int stage, int step) {
        const float* blk = wtile + (long long)step * (WROWS * 32) + tid * 4;
        const unsigned base = lds0 + 2 * HALO_BYTES + stage * WBYTES + wave * 1024;
#pragma unroll
        for (int i = 0; i < NT; ++i) glds16(blk + i * 1024, base + i * 4096);
    };

    // ---- fragment addressing ----
    // A rows: i -> 2x2-window-major pixel order, so that the 4 accumulator registers
    // (reg&3) of one lane are exactly one pooling window (see epilogue).
    const int px0 = 2 * (li >> 2) + (li & 1);
    const int pyl = (li >> 1) & 1;
    const int brow = wn * 32 + li;
    const int boff = brow * 128 + ((lh ^ ((brow >> 1) & 7)) << 4);

    f32x16 acc[MS];
#pragma unroll
    for (int m = 0; m < MS; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;

    const int total_steps = a.n_chunks * TAPS;

    stage_halo(0, 0);
    stage_w(0, 0);
    og_wait_dma();
    __syncthreads();

    int step = 0;
    for (int c = 0; c < a.n_chunks; ++c) {
        const unsigned char* hb = halo0 + (c & 1) * HALO_BYTES;
#pragma unroll
        for (int t = 0; t < TAPS; ++t, ++step) {
            if (step + 1 < total_steps) stage_w((step + 1) & 1, step + 1);
            if (t == 0 && c + 1 < a.n_chunks) stage_halo((c + 1) & 1, c + 1);

            const unsigned char* wb = wbuf0 + (step & 1) * WBYTES;
            const int dy = (MODE == 0) ? t / 3 : 0;
            const int dx = (MODE == 0) ? t % 3 : 0;
            int aoff[MS];
#pragma unroll
            for (int m = 0; m < MS; ++m) {
                const int py = 2 * (wm * MS + m) + pyl + dy;
                const int p = py * HW_ + px0 + dx;
                aoff[m] = p * 128 + ((lh ^ ((p >> 1) & 7)) << 4);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                // slot = (2j + lh) ^ swz  ==  ((lh ^ swz) << 4) ^ (j << 5) in bytes
                const f32x4 bv = *(const f32x4*)(wb + (boff ^ (j << 5)));
#pragma unroll
                for (int m = 0; m < MS; ++m) {
                    const f32x4 av = *(const f32x4*)(hb + (aoff[m] ^ (j << 5)));
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc[m], 0, 0, 0);
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc[m], 0, 0, 0);
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc[m], 0, 0, 0);
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc[m], 0, 0, 0);
                }
            }
            og_wait_dma();
            __syncthreads();
        }
    }

    // ---- epilogue: affine (+ReLU), store, optional fused 2x2 max-pool ----
    const int ncol = n_tile * WROWS + wn * 32 + li;  // GEMM column of this lane
    int co = ncol, qd = 0;
    if (MODE == 1) {
        qd = ncol / a.aff_mod;
        co = ncol - qd * a.aff_mod;
    }
    const float sc = a.scale[co];
    const float sh = a.shift[co];
    const int OW = (MODE == 1) ? 2 * a.W : a.W;
    float* out_frame = a.out + (long long)b * a.out_frame_stride + a.out_ch_off + co;
#pragma unroll
    for (int m = 0; m < MS; ++m) {
        const int ms = wm * MS + m;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int wdw = 2 * g + lh;  // 2x2 window index along x within the sub-tile
            float vmax = 0.f;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                float v = fmaf(acc[m][4 * g + rr], sc, sh);
                v = og_act(v, a.act);
                const int y = ty0 + 2 * ms + (rr >> 1);
                const int x = tx0 + 2 * wdw + (rr & 1);
                if (y < a.H && x < a.W) {
                    if (MODE == 0) {
                        out_frame[((long long)y * OW + x) * a.out_pix_stride] = v;
                    } else {
                        out_frame[((long long)(2 * y + (qd >> 1)) * OW + (2 * x + (qd & 1))) * a.out_pix_stride] = v;
                    }
                }
                vmax = (rr == 0) ? v : fmaxf(vmax, v);
            }
            if (MODE == 0 && a.pool != nullptr) {
                const int y = ty0 + 2 * ms, x = tx0 + 2 * wdw;
                if (y < a.H && x < a.W) {
                    float* pf = a.pool + (long long)b * a.pool_frame_stride + a.pool_ch_off + co;
                    pf[((long long)(y >> 1) * (a.W >> 1) + (x >> 1)) * a.pool_pix_stride] = vmax;
                }
            }
        }
    }
}

// Shared epilogue of k_conv_mfma_p / k_conv_mfma_o / k_splitk_epilogue: per-channel affine, activation,
// optional residual, NHWC store (scattered for the transposed conv), optional fused 2x2 max-pool.
//
// The accumulator layout has one output channel per lane, so the natural store is 16 dword stores
// (4 B/lane) per 32x32 tile.  Vector-memory instructions are what this kernel cannot afford (each costs
// the SIMD ~85 cycles of matrix-pipe time, tools/ubench/mfma_f32_mix.hip), so the tile is transposed
// through a private 4 KB LDS scratch (the staging buffers are dead by now) and leaves as 4 dwordx4
// stores: 8 lanes write one pixel's 32 channels (128 B).  `scratch` = this wave's 5 KB (4 KB tile + 1 KB pooled tile).
template <int NT, int MODE, int TH>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& a, const f32x16* acc, int n_tile, int b, int ty0, int tx0, int wm, int wn,
                                              int li, int lh, float sc, float sh, unsigned char* scratch) {
    constexpr int WROWS = 32 * NT;
    constexpr int WM = 4 / NT;
    constexpr int MS = (TH / 2) / WM;
    const int lane = li + 32 * lh;
    const int ncol0 = n_tile * WROWS + wn * 32;   // first GEMM column of this wave's tile
    int cbase = ncol0, qd = 0;                    // first output channel of the tile (+ quadrant for convT)
    if (MODE == 1) {
        qd = ncol0 / a.aff_mod;
        cbase = ncol0 - qd * a.aff_mod;
    }
    const int co = cbase + li;
    const int OW = (MODE == 1) ? 2 * a.W : a.W;
    float* const fs = (float*)scratch;
    const int rrow = lane >> 3;          // read-back: this lane handles pixel rows rrow + 8q, channels 4*(lane&7)..+3
    const int rc4 = (lane & 7) * 4;
    int head_cnt = 0, hbx1 = 0, hby1 = 0, hbx2 = 1 << 30, hby2 = 1 << 30;
    f32x4 wv = {0.f, 0.f, 0.f, 0.f};  // head weights of this lane's 4 channels: loaded ONCE, ahead of every store
    if (MODE == 0 && NT == 1 && a.head_w != nullptr) wv = *(const f32x4*)(a.head_w + rc4);
    if (MODE == 0 && NT == 1 && a.head_w != nullptr && a.head_boxes != nullptr) {
        hbx1 = a.head_boxes[b * 4 + 0];
        hby1 = a.head_boxes[b * 4 + 1];
        hbx2 = a.head_boxes[b * 4 + 2];
        hby2 = a.head_boxes[b * 4 + 3];
        if (hbx1 < 0) { hbx1 = hby1 = hbx2 = hby2 = 0; }  // "no detection" -> area 0 (features.py:241-242)
    }
#pragma unroll
    for (int m = 0; m < MS; ++m) {
        const int ms = wm * MS + m;
        float vmaxs[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int wdw = 2 * g + lh;
            float vmax = 0.f;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                float v = fmaf(acc[m][4 * g + rr], sc, sh);
                v = og_act(v, a.act);
                if (MODE != 1 && a.res != nullptr) {
                    const int y = ty0 + 2 * ms + (rr >> 1), x = tx0 + 2 * wdw + (rr & 1);
                    if (y < a.H && x < a.W)
                        v += a.res[(long long)b * a.res_frame_stride + ((long long)y * OW + x) * a.res_pix_stride + a.res_ch_off + co];
                }
                fs[(4 * wdw + rr) * 32 + li] = v;   // [pixel row i = 4*window + rr][channel]
                vmax = (rr == 0) ? v : fmaxf(vmax, v);
            }
            vmaxs[g] = vmax;
        }
        // full-resolution tile: 4 x (64 lanes x 16 B)
        const bool fuse_head = (MODE == 0 && NT == 1 && a.head_w != nullptr);
        const bool store_act = !fuse_head || a.head_store_act;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = rrow + 8 * q;
            const f32x4 v4 = *(const f32x4*)(fs + i * 32 + rc4);
            const int wdw = i >> 2, rr = i & 3;
            const int y = ty0 + 2 * ms + (rr >> 1), x = tx0 + 2 * wdw + (rr & 1);
            const bool inb = (y < a.H && x < a.W);
            if (inb && store_act) {
                const long long pix = (MODE != 1) ? ((long long)y * OW + x) : ((long long)(2 * y + (qd >> 1)) * OW + (2 * x + (qd & 1)));
                *(f32x4*)(a.out + (long long)b * a.out_frame_stride + pix * a.out_pix_stride + a.out_ch_off + cbase + rc4) = v4;
            }
            if (MODE == 0 && NT == 1 && fuse_head) {
                float sdot = 0.f;
                sdot = fmaf(v4.x, wv.x, sdot);
                sdot = fmaf(v4.y, wv.y, sdot);
                sdot = fmaf(v4.z, wv.z, sdot);
                sdot = fmaf(v4.w, wv.w, sdot);
                sdot += __shfl_xor(sdot, 1);
                sdot += __shfl_xor(sdot, 2);
                sdot += __shfl_xor(sdot, 4);
                if ((lane & 7) == 0 && inb) {
                    const float lg = sdot + a.head_bias;
                    const float prob = 1.0f / (1.0f + expf(-lg));
                    const bool on = prob > a.head_thr;
                    const long long o = (long long)b * a.H * a.W + (long long)y * a.W + x;
                    if (a.head_logits) a.head_logits[o] = lg;
                    if (a.head_mask) a.head_mask[o] = on ? 255 : 0;
                    head_cnt += (on && x >= hbx1 && x < hbx2 && y >= hby1 && y < hby2) ? 1 : 0;
                }
            }
        }
        if (MODE == 0 && a.pool != nullptr) {
            // pooled tile: 8 windows x 32 channels = exactly one 16-B store per lane
#pragma unroll
            for (int g = 0; g < 4; ++g) fs[1024 + (2 * g + lh) * 32 + li] = vmaxs[g];   // scratch bytes [4096, 5120)
            const f32x4 p4 = *(const f32x4*)(fs + 1024 + rrow * 32 + rc4);
            const int y = ty0 + 2 * ms, x = tx0 + 2 * rrow;
            if (y < a.H && x < a.W)
                *(f32x4*)(a.pool + (long long)b * a.pool_frame_stride + ((long long)(y >> 1) * (a.W >> 1) + (x >> 1)) * a.pool_pix_stride +
                          a.pool_ch_off + cbase + rc4) = p4;
        }
    }
    if (MODE == 0 && NT == 1 && a.head_w != nullptr && a.head_area != nullptr) {
        // one plain store per wave into a per-(frame, tile, wave) slot; k_sum_counts adds them up per frame.
        // (131 072 atomics onto 64 addresses per launch were measured to double this kernel's time.)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) head_cnt += __shfl_down(head_cnt, o);
        if (lane == 0) {
            const int tile = (ty0 / TH) * a.tiles_x + (tx0 >> 4);
            a.head_area[((long long)b * a.tiles_x * a.tiles_y + tile) * 4 + (wm * NT + wn)] = head_cnt;
        }
    }
}

// Lean epilogue of k_conv_mfma_o.  Same arithmetic, element order and results as conv_epilogue, but written so
// that almost no vector-ALU instruction is spent on addressing (each one costs the SIMD matrix-pipe time):
//  * every global access goes through a raw buffer resource: the lane-dependent byte offset is computed ONCE per
//    wave, the (sub-tile, row group) advance is a wave-uniform SGPR offset, and lanes outside the image carry the
//    offset OG_OOB, which the bounds check drops -- no per-store address arithmetic, no exec masking;
//  * activation and residual are template parameters (no per-element uniform branches);
//  * the fused head evaluates sigmoid / threshold / box test once per 32-pixel sub-tile (one pixel per lane) and
//    counts with a ballot + scalar popcount instead of four 8-pixel rounds and a shuffle reduction.
typedef unsigned og_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t og_rsrc(const void* base, unsigned bytes) {
    const unsigned long long v = (unsigned long long)base;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, (int)__builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
// 16-byte store through a raw buffer resource with the wave-uniform part of the address in an SGPR (soffset), followed
// by ONE WAIT STATE.  The wait state is required on gfx950: a buffer_store_dwordx4 whose data VGPRs are overwritten by
// the very next instruction (here: the next sub-tile's v_pk_fma_f32) stores garbage in one 16-lane beat of one dword.
// It is the GFX9 "VMEM store of more than 64 bits, then a VALU write of the write-data VGPRs: 1 wait state" hazard;
// hipcc's hazard recognizer waives it for stores that use an SGPR soffset (GCNHazardRecognizer::createsVALUHazard), which
// is exactly this store -- so the nop has to be ours.  Root cause of round 1's "non-repeatable wrong lanes"
// (profiles/r02_epilogue_fence_audit.md); tests/test_isa_audit.py scans the shipped ISA for the pattern.
#ifndef OG_STORE_NOP
#define OG_STORE_NOP 1   // -DOG_STORE_NOP=0: audit build that reproduces the failure
#endif
__device__ __forceinline__ void og_buffer_store16(f32x4 v, __amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
#if OG_STORE_NOP
    asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_nop 0" : : "v"(v), "v"(voff), "s"(rs), "s"(soff) : "memory");
#else
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(og_u32x4, v), rs, voff, soff, 0);
#endif
}
template <int NT, int MODE, int TH, int ACT, bool RES, bool HALF = false, int MSO = 0>
__device__ __forceinline__ void conv_epilogue_b(const ConvArgs& a, const f32x16* acc, int n_tile, int b, int ty0, int tx0, int wm, int wn,
                                                int li, int lh, float sc, float sh, unsigned char* scratch) {
    constexpr int WROWS = 32 * NT;
    constexpr int WM = 4 / NT;
    constexpr int MS = MSO ? MSO : (TH / 2) / WM;   // MSO: sub-tiles per wave when the caller lays its waves out differently (wm then counts in units of MSO)
    const int lane = li + 32 * lh;
    const int ncol0 = n_tile * WROWS + wn * 32;
    int cbase = ncol0, qd = 0;
    if (MODE == 1) {
        qd = ncol0 / a.aff_mod;
        cbase = ncol0 - qd * a.aff_mod;
    }
    const int OW = (MODE == 1) ? 2 * a.W : a.W;
    // read-back role of this lane: pixel row (rrow + 8q) of the 32-row sub-tile, channels rc4..rc4+3
    const int rrow = lane >> 3, rc4 = (lane & 7) * 4;
    const int xl = 2 * (rrow >> 2) + (rrow & 1), yl = (rrow & 3) >> 1;  // pixel inside the (2 rows x 16 columns) sub-tile: x = xl + 4q
    const bool fuse_head = (MODE == 0 && NT == 1 && a.head_w != nullptr);
    const bool store_act = !fuse_head || a.head_store_act;

    const __amdgpu_buffer_rsrc_t out_rs = og_rsrc(a.out + (long long)b * a.out_frame_stride, (unsigned)a.out_frame_stride * 4u);
    const unsigned lp = (MODE == 1) ? (unsigned)(((2 * yl * OW + 2 * xl) * a.out_pix_stride + rc4) * 4)
                                    : (unsigned)(((yl * OW + xl) * a.out_pix_stride + rc4) * 4);
    unsigned vq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) vq[q] = (tx0 + xl + 4 * q < a.W) ? lp : OG_OOB;

    // accumulator-layout role: channel li, pixel rows 8g + 4lh + rr
    float* const fw = (float*)scratch + (128 * lh + li);            // + (8g + rr) * 32
    const float* const fr = (const float*)scratch + rrow * 32 + rc4;  // + q * 256
    __amdgpu_buffer_rsrc_t pool_rs = out_rs;
    unsigned vpool = 0;
    const bool pool = (MODE == 0 && a.pool != nullptr);
    if (pool) {
        pool_rs = og_rsrc(a.pool + (long long)b * a.pool_frame_stride, (unsigned)a.pool_frame_stride * 4u);
        vpool = (tx0 + 2 * rrow < a.W) ? (unsigned)((rrow * a.pool_pix_stride + rc4) * 4) : OG_OOB;
    }
    // fused head: weights of this lane's 4 channels, loaded once ahead of every store; one pixel per lane k < 4
    f32x4 wv = {0.f, 0.f, 0.f, 0.f};
    int hbx1 = 0, hby1 = 0, hbx2 = 1 << 30, hby2 = 1 << 30, head_cnt = 0;
    const int hk = lane & 7;
    unsigned vh = OG_OOB;  // pixel offset (elements) of this lane's head pixel inside the frame, relative to the sub-tile origin
    if (fuse_head) {
        wv = *(const f32x4*)(a.head_w + rc4);
        if (a.head_boxes != nullptr) {
            hbx1 = a.head_boxes[b * 4 + 0];
            hby1 = a.head_boxes[b * 4 + 1];
            hbx2 = a.head_boxes[b * 4 + 2];
            hby2 = a.head_boxes[b * 4 + 3];
            if (hbx1 < 0) { hbx1 = hby1 = hbx2 = hby2 = 0; }  // "no detection" -> area 0 (features.py:241-242)
        }
        if (hk < 4 && tx0 + xl + 4 * hk < a.W) vh = (unsigned)(yl * a.W + xl + 4 * hk);
    }

    float h_absmax = 0.f;   // split precision: largest |activation| this lane splits (range check below)
#pragma unroll
    for (int m = 0; m < MS; ++m) {
        const int ms = wm * MS + m;
        const int y0 = ty0 + 2 * ms;          // first of this sub-tile's two pixel rows (H is even: both inside or both outside)
        float vmaxs[4];
        const f32x2 sc2 = {sc, sc}, sh2 = {sh, sh};
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x2 a01 = og_fma2(f32x2{acc[m][4 * g], acc[m][4 * g + 1]}, sc2, sh2);
            const f32x2 a23 = og_fma2(f32x2{acc[m][4 * g + 2], acc[m][4 * g + 3]}, sc2, sh2);
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                float v = (rr == 0) ? a01.x : (rr == 1) ? a01.y : (rr == 2) ? a23.x : a23.y;
                if (ACT == 1) v = fmaxf(v, 0.f);
                if (ACT == 2) v = v / (1.0f + expf(-v));
                if (RES) {  // detector bottlenecks only: plain per-element loads
                    const int y = y0 + (rr >> 1), x = tx0 + 4 * g + 2 * lh + (rr & 1);
                    if (y < a.H && x < a.W)
                        v += a.res[(long long)b * a.res_frame_stride + ((long long)y * OW + x) * a.res_pix_stride + a.res_ch_off + cbase + li];
                }
                if (HALF && !fuse_head) {   // H layout: row i = 8g + 4lh + rr of the tile image, hi at byte 2*li, lo at 64 + 2*li
                    _Float16 hi, lo;
                    og_split(v, hi, lo);
                    h_absmax = fmaxf(h_absmax, fabsf(v));
                    _Float16* const hw = (_Float16*)scratch + (256 * lh + li) + (8 * g + rr) * 64;
                    hw[0] = hi;
                    hw[32] = lo;
                } else {
                    fw[(8 * g + rr) * 32] = v;
                }
                vmaxs[g] = (rr == 0) ? v : fmaxf(vmaxs[g], v);
            }
        }
        f32x4 v4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v4[q] = *(const f32x4*)(fr + q * 256);
        if (y0 < a.H) {
            const bool half = (y0 + 1 >= a.H);   // odd H (detector maps of 160-pixel inputs): only the first row exists
            if (store_act) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int so = (MODE == 1) ? (((2 * y0 + (qd >> 1)) * OW + 2 * (tx0 + 4 * q) + (qd & 1)) * a.out_pix_stride + a.out_ch_off + cbase) * 4
                                               : ((y0 * OW + tx0 + 4 * q) * a.out_pix_stride + a.out_ch_off + cbase) * 4;
                    og_buffer_store16(v4[q], out_rs, (half && yl) ? OG_OOB : vq[q], (unsigned)so);
                }
            }
            if (MODE == 0 && NT == 1 && fuse_head) {
                float sd[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float sdot = 0.f;
                    sdot = fmaf(v4[q].x, wv.x, sdot);
                    sdot = fmaf(v4[q].y, wv.y, sdot);
                    sdot = fmaf(v4[q].z, wv.z, sdot);
                    sdot = fmaf(v4[q].w, wv.w, sdot);
                    sdot += __shfl_xor(sdot, 1);
                    sdot += __shfl_xor(sdot, 2);
                    sdot += __shfl_xor(sdot, 4);
                    sd[q] = sdot;   // every lane of the 8-lane group holds the pixel's sum
                }
                const float mine = (hk == 0) ? sd[0] : (hk == 1) ? sd[1] : (hk == 2) ? sd[2] : sd[3];  // lane k < 4: pixel row rrow + 8k
                const float lg = mine + a.head_bias;
                const float prob = 1.0f / (1.0f + expf(-lg));
                const bool on = prob > a.head_thr;
                const int x = tx0 + xl + 4 * hk, y = y0 + yl;
                const bool counted = on && vh != OG_OOB && x >= hbx1 && x < hbx2 && y >= hby1 && y < hby2;
                head_cnt += __builtin_popcountll(__ballot(counted));
                const unsigned so = (unsigned)(y0 * a.W + tx0);
                if (a.head_logits) {
                    const __amdgpu_buffer_rsrc_t lrs = og_rsrc(a.head_logits + (long long)b * a.H * a.W, (unsigned)(a.H * a.W) * 4u);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, lg), lrs, (vh == OG_OOB) ? OG_OOB : vh * 4u, so * 4u, 0);
                }
                if (a.head_mask) {
                    const __amdgpu_buffer_rsrc_t mrs = og_rsrc(a.head_mask + (long long)b * a.H * a.W, (unsigned)(a.H * a.W));
                    __builtin_amdgcn_raw_buffer_store_b8((unsigned char)(on ? 255 : 0), mrs, vh, so, 0);
                }
            }
            if (pool) {
                // pooled tile: 8 windows x 32 channels = exactly one 16-B store per lane
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    if (HALF) {   // pooled tile image at scratch + 4096: row 2g + lh, hi at byte 2*li, lo at 64 + 2*li
                        _Float16 hi, lo;
                        og_split(vmaxs[g], hi, lo);
                        _Float16* const hw = (_Float16*)scratch + 2048 + (2 * g + lh) * 64 + li;
                        hw[0] = hi;
                        hw[32] = lo;
                    } else {
                        fw[1024 - 96 * lh + g * 64] = vmaxs[g];  // fs[1024 + (2g + lh) * 32 + li]
                    }
                }
                const f32x4 p4 = *(const f32x4*)(fr + 1024);
                const int so = (((y0 >> 1) * (a.W >> 1) + (tx0 >> 1)) * a.pool_pix_stride + a.pool_ch_off + cbase) * 4;
                og_buffer_store16(p4, pool_rs, vpool, (unsigned)so);
            }
        }
    }
    if (HALF) og_flag_range(h_absmax, a.range_flag);
    if (MODE == 0 && NT == 1 && fuse_head && a.head_area != nullptr) {
        // one plain store per wave into a per-(frame, tile, wave) slot; k_sum_counts adds them up per frame
        if (lane == 0) {
            const int tile = (ty0 / TH) * a.tiles_x + (tx0 >> 4);
            a.head_area[((long long)b * a.tiles_x * a.tiles_y + tile) * 4 + (wm * NT + wn)] = head_cnt;
        }
    }
}

// Per-frame sum of the fused head's per-(tile, wave) pixel counts -> area[b]  (features.py:238 / 244-245).
__global__ __launch_bounds__(256) void k_sum_counts(const int32_t* __restrict__ counts, int per_frame, int32_t* __restrict__ area) {
    __shared__ int s[256];
    const int b = blockIdx.x;
    int acc = 0;
    for (int i = threadIdx.x; i < per_frame; i += 256) acc += counts[(long long)b * per_frame + i];
    s[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) area[b] = s[0];
}

// 16-B slot swizzle of the halo image in k_conv_mfma_o: with the 2x2-window-major lane -> pixel order, each
// 16-lane group of a ds_read_b128 ({0-3,12-15,20-27}, ...) touches 8 pixels of an even and 8 of an odd row; the
// column pair index XOR 4x the row parity gives the 16 lanes 16 distinct (128-B half, slot) places -> conflict-free
// (the linear-pixel swizzle (p >> 1) & 7 used by the other variants is 2-way conflicted on these reads).
__device__ __forceinline__ int og_halo_swz(int py, int px) { return ((px >> 1) ^ ((py & 1) << 2)) & 7; }

// Occupancy variant: ONE halo buffer (reloaded at each chunk boundary, the stall is covered by the
// other workgroups) -> 39 KB of LDS, so 3-4 workgroups fit a CU instead of 2.  Same arithmetic and
// accumulation order as k_conv_mfma.
// VS ("virtual split", the detector's batched launches): the workgroup runs the WHOLE K range but keeps the partial sums of an
// a.vsplit-way split apart -- at every part boundary the running accumulators are folded into `vsum` and restart from zero, and the
// epilogue sees vsum + last part -- i.e. exactly what the fused reduce of a real split (ksplit = vsplit) computes:
// ((0 + p0) + p1) + ...  A frame's detector output then does not depend on whether its launch split K over workgroups (one-frame
// latency path) or not (batched path).
template <int NT, int MODE, int TH, int OCC, bool FIRST = false, bool VS = false>
__global__ __launch_bounds__(256, OCC) void k_conv_mfma_o(ConvArgs a) {
    constexpr int TW = 16;
    constexpr int PAD = (MODE == 0 || MODE == 3) ? 1 : 0;                 // rows/columns of halo above / left of the tile
    constexpr int HW_ = (MODE == 3) ? TW + 2 : TW + 2 * PAD;             // MODE 3 needs 17 columns; an even pitch keeps
    constexpr int HH_ = (MODE == 3) ? TH + 1 : TH + 2 * PAD;             // the slot swizzle conflict-free
    constexpr int HALO_PIX = HW_ * HH_;
    constexpr int HALO_BYTES = HALO_PIX * 128;
    constexpr int HALO_PIECES = HALO_PIX * 8;
    constexpr int HALO_IT = (HALO_PIECES + 255) / 256;
    constexpr int TAPS = (MODE == 0) ? 9 : (MODE == 3) ? 4 : 1;
    constexpr int WROWS = 32 * NT;
    constexpr int WBYTES = WROWS * 128;
    // weight ring: 3 stages for the 3x3 conv, so that a tap's stage (t % 3, nine taps per chunk) is a compile-time
    // constant and the B-fragment reads need no address arithmetic; 2 stages (runtime parity) elsewhere
    constexpr int NSTG = (MODE == 0) ? 3 : 2;
    constexpr int WM = 4 / NT;
    constexpr int MS = (TH / 2) / WM;  // 32-row M sub-tiles (2 pixel rows x 16) per wave
    static_assert(MS >= 1, "tile too small");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const halo0 = smem;
    unsigned char* const wbuf0 = smem + HALO_BYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave % NT;
    const int wm = wave / NT;
    const int li = lane & 31;
    const int lh = lane >> 5;

    // A young wave's VALU/VMEM instructions only get issue slots between the MFMAs of the older waves on its SIMD
    // (measured: 14 k cycles from entry to the first DMA, 12 k for the epilogue, vs 17 k in the main loop of a
    // 32-channel layer - tools/ubench/occ_timeline).  Raise the priority outside the main loop so that the address
    // set-up, the DMA issue and the epilogue are served first; the main loops only need a slot every 64 cycles.
    if (a.prio_mode == 3) __builtin_amdgcn_s_setprio(3);
    // diagnostic timeline (tools/ubench/occ_timeline.hip only; nullptr on every product path)
    unsigned long long* const st = a.stamps ? a.stamps + 8ull * (((unsigned long long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) : nullptr;
    if (st != nullptr && tid == 0) {
        st[0] = __builtin_amdgcn_s_memtime();
        st[4] = __builtin_amdgcn_s_getreg((31 << 11) | 4);  // HW_REG_HW_ID: wave slot, SIMD, CU, SE
        st[5] = __builtin_amdgcn_s_getreg((31 << 11) | 20); // HW_REG_XCC_ID
    }

    // ---- tile decode (scalar).  3-D grid (tile column, tile row, frame x column tile x K part): no integer division
    //      on the way to the first DMA (each one is ~20 vector-ALU instructions, paid at the contended issue rate).
    //      split-K (latency mode, MODE 0/1): K part = a range of (chunk, tap) steps ----
    const int ks_n = a.ksplit;
    // grid.z = frame group q x (column tile, K part) x frame-in-group: the column tiles of one spatial tile then sit
    // G * tiles_x * tiles_y (a multiple of 8) blocks apart = on the same XCD, dispatched together, and share the input
    // tile in that XCD's L2 (G = 2^zgroup_shift frames per group; 1 when tiles_x * tiles_y is already a multiple of 8)
    const int bz = (int)blockIdx.z;
    const int gz = a.zdiv << a.zgroup_shift;
    const int q = (gz == 1) ? bz : (int)(((float)bz + 0.5f) * a.zrcp);   // zrcp = 1 / gz; exact for bz < 2^16
    const int rz = bz - q * gz;
    const int zr = rz >> a.zgroup_shift;                                   // column tile x K parts + K part
    const int b = (q << a.zgroup_shift) + (rz & ((1 << a.zgroup_shift) - 1));
    if (b >= a.frames) return;   // tail of the last frame group (whole workgroup, before any barrier)
    const int n_tile = (ks_n == 1) ? zr : zr / ks_n;
    const int kpart = (ks_n == 1) ? 0 : zr - n_tile * ks_n;
    // K parts: ranges of (chunk, tap) steps for the 3x3 conv, of whole chunks elsewhere.  MODE 3 (virtual chunk =
    // (input-pixel parity, 32-channel chunk), 1 / 2 / 2 / 4 non-zero taps per parity): parts are ranges of virtual chunks
    constexpr int TAPS_ = (MODE == 0) ? 9 : 1;
    const int cpc = (MODE == 3) ? a.n_chunks >> 2 : 1;
    auto steps_before = [&](int c) {   // MODE 3: weight steps that precede virtual chunk c
        const int par = c / cpc;
        const int before = (par == 0) ? 0 : (par == 1) ? 1 : (par == 2) ? 3 : (par == 3) ? 5 : 9;
        return before * cpc + (c - par * cpc) * ((1 + (par >> 1)) * (1 + (par & 1)));
    };
    int s_lo, s_hi, c_lo, c_hi;
    if (MODE == 3) {
        c_lo = (ks_n == 1) ? 0 : (kpart * a.n_chunks) / ks_n;
        c_hi = (ks_n == 1) ? a.n_chunks : ((kpart + 1) * a.n_chunks) / ks_n;
        s_lo = (ks_n == 1) ? 0 : steps_before(c_lo);
        s_hi = (ks_n == 1) ? cpc * 9 : steps_before(c_hi);
    } else {
        s_lo = (ks_n == 1) ? 0 : (kpart * a.n_chunks * TAPS_) / ks_n;
        s_hi = (ks_n == 1) ? a.n_chunks * TAPS_ : ((kpart + 1) * a.n_chunks * TAPS_) / ks_n;
        c_lo = (ks_n == 1) ? 0 : s_lo / TAPS_;
        c_hi = (ks_n == 1) ? a.n_chunks : (s_hi + TAPS_ - 1) / TAPS_;
    }
    const int ty0 = (int)blockIdx.y * TH;
    const int tx0 = (int)blockIdx.x * TW;
    // linear ids (split-K partial buffer, diagnostic stamps): as k_splitk_epilogue decodes them
    const int tile_id = n_tile * a.n_spatial + (b * a.tiles_y + (int)blockIdx.y) * a.tiles_x + (int)blockIdx.x;
    const int item_id = tile_id * ks_n + kpart;

    // this frame's input as a raw buffer: offsets past num_records read as zeros (= the conv's zero padding)
    const og_i32x4 in_rsrc = og_make_rsrc(a.in + (long long)b * a.in_frame_stride + a.in_ch_off,
                                          (unsigned)(a.in_frame_stride - a.in_ch_off) * 4u);

    // ---- per-thread halo source offsets (bytes, fixed across channel chunks); pixel index walks incrementally ----
    unsigned hoff[HALO_IT];
    {
        int hy = ((tid >> 3) >= HW_) ? 1 : 0;
        int hx = (tid >> 3) - hy * HW_;
        const int in_w = (MODE == 3) ? 2 * a.W : a.W;
        const int row_b = in_w * a.in_pix_stride * ((MODE == 3) ? 8 : 4);   // bytes per halo row step
        const int col_b = a.in_pix_stride * ((MODE == 3) ? 8 : 4);          // bytes per halo column step
#pragma unroll
        for (int it = 0; it < HALO_IT; ++it) {
            const int logical = (tid & 7) ^ og_halo_swz(hy, hx);
            const int gy = ty0 + hy - PAD, gx = tx0 + hx - PAD;
            // MODE 3: (gy, gx) is a 2x2 input block = the output pixel grid; the block's (0,0) pixel is the base
            const bool inb = (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
            hoff[it] = inb ? (unsigned)(gy * row_b + gx * col_b + logical * 16) : OG_OOB;
            hx += 32 % HW_;
            hy += 32 / HW_;
            if (hx >= HW_) { hx -= HW_; hy += 1; }
        }
    }
    const bool last_valid = ((HALO_IT - 1) * 256 + tid) < HALO_PIECES;

    const unsigned lds0 = og_lds_addr(smem);
    // MODE 3 (3x3 stride-2 conv as a 2x2 stride-1 conv over the space-to-depth view, never materialised):
    // virtual chunk c = (input-pixel parity par = c / cpc, 32-channel chunk c % cpc)
    auto stage_halo = [&](int buf, int c) {
        const unsigned base = lds0 + wave * 1024;
        (void)buf;
        unsigned soff = (unsigned)c * 128u;  // wave-uniform byte offset of the chunk
        if (MODE == 3) {
            const int par = c / cpc;
            soff = (unsigned)((((par >> 1) * 2 * a.W + (par & 1)) * a.in_pix_stride + (c - par * cpc) * 32) * 4);
        }
#pragma unroll
        for (int it = 0; it < HALO_IT; ++it) {
            if (it < HALO_IT - 1 || last_valid) glds16b(hoff[it], in_rsrc, soff, base + it * 4096);
        }
    };
    const int total_steps = (MODE == 3) ? cpc * 9 : a.n_chunks * TAPS;  // MODE 3: only the 9 non-zero (tap, parity) pairs
    const og_i32x4 w_rsrc = og_make_rsrc(a.wpk + (long long)n_tile * total_steps * (WROWS * 32), (unsigned)total_steps * WBYTES);
    const unsigned woff = (unsigned)tid * 16u;
    auto stage_w = [&](int stage, int step) {
        const unsigned base = lds0 + HALO_BYTES + stage * WBYTES + wave * 1024;
#pragma unroll
        for (int i = 0; i < NT; ++i) glds16b(woff, w_rsrc, (unsigned)step * WBYTES + i * 4096, base + i * 4096);
    };

    // first halo and weights are on their way before the rest of the set-up (which then hides their latency)
    if (!FIRST) {
        if (st != nullptr && tid == 0) st[6] = __builtin_amdgcn_s_memtime();
        stage_halo(0, c_lo);
        stage_w((NSTG == 3) ? s_lo % 3 : (s_lo & 1), s_lo);
    }

    // ---- fragment addressing ----
    // A rows: i -> 2x2-window-major pixel order, so that the 4 accumulator registers
    // (reg&3) of one lane are exactly one pooling window (see epilogue).
    const int px0 = 2 * (li >> 2) + (li & 1);
    const int pyl = (li >> 1) & 1;
    const int brow = wn * 32 + li;
    const int boff = brow * 128 + ((lh ^ ((brow >> 1) & 7)) << 4);
    // Everything lane-dependent of an A-fragment address sits in NDX x 4 registers:
    //   abase[dx][pat] = ((pyl*HW_ + px0+dx) * 128 + ((lh ^ swz(pyl, px0+dx)) << 4) + wave row offset) ^ (pat << 5)
    // and a tap's dy, the k-group j and the M sub-tile m only select pat = j ^ ((dy & 1) << 1) (row parity flips slot
    // bit 2) and add the compile-time constant (dy + 2m) * HW_ * 128, which the ds_read carries as its immediate.
    constexpr int NDX = (MODE == 0) ? 3 : (MODE == 3) ? 2 : 1;
    unsigned abase[NDX][4];
#pragma unroll
    for (int dx = 0; dx < NDX; ++dx) {
        const int px = px0 + dx;
        const unsigned o = (unsigned)((pyl * HW_ + px) * 128 + ((lh ^ og_halo_swz(pyl, px)) << 4) + wm * (MS * 2 * HW_ * 128));
#pragma unroll
        for (int pat = 0; pat < 4; ++pat) {
            abase[dx][pat] = lds0 + (o ^ (unsigned)(pat << 5));   // absolute LDS address
            asm volatile("" : "+v"(abase[dx][pat]));              // opaque: or hipcc re-adds the (link-time 0) smem base per read
        }
    }
    unsigned bbase[4];   // B fragments: stage 0 of the weight ring, k-group j
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        bbase[jj] = lds0 + HALO_BYTES + (unsigned)(boff ^ (jj << 5));
        asm volatile("" : "+v"(bbase[jj]));
    }

    // this lane's output channel: folded BN scale / shift, loaded now so that the epilogue does not wait for them
    const int ecol = n_tile * WROWS + wn * 32 + li;
    const int eco = (MODE == 1) ? ecol % a.aff_mod : ecol;
    const float esc = a.scale[eco], esh = a.shift[eco];

    f32x16 acc[MS];
#pragma unroll
    for (int m = 0; m < MS; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;

    if (FIRST) {
        // halo tile of the first layer's OUTPUT, computed here: 12x20 u8 patch -> /255 -> 3x3 conv -> BN -> ReLU,
        // written in the same swizzled [pixel][8 x 16 B] image the LDS-DMA would have produced
        float* patch = (float*)(smem + HALO_BYTES + NSTG * WBYTES);  // [HH_+2][HW_+2]
        float* fw = patch + (HH_ + 2) * (HW_ + 2);                // w9[9][32] | scale[32] | shift[32]
        const uint8_t* fin = a.first_u8 + (long long)b * a.H * a.W;
        for (int i = tid; i < (HH_ + 2) * (HW_ + 2); i += 256) {
            const int hy = i / (HW_ + 2), hx = i - hy * (HW_ + 2);
            const int gy = ty0 + hy - 2, gx = tx0 + hx - 2;
            patch[i] = (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) ? (float)fin[(long long)gy * a.W + gx] / 255.0f : 0.f;
        }
        for (int i = tid; i < 352; i += 256) fw[i] = (i < 288) ? a.first_w9[i] : (i < 320 ? a.first_scale[i - 288] : a.first_shift[i - 320]);
        stage_w(0, 0);
        __syncthreads();
#pragma unroll
        for (int it = 0; it < HALO_IT; ++it) {
            const int q = it * 256 + tid;
            if (q < HALO_PIECES) {
                const int p = q >> 3;
                const int hy = p / HW_, hx = p - hy * HW_;
                const int c0 = ((q & 7) ^ og_halo_swz(hy, hx)) * 4;
                const int gy = ty0 + hy - 1, gx = tx0 + hx - 1;
                f32x4 o = {0.f, 0.f, 0.f, 0.f};   // outside the image: the second conv's zero padding
                if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) {
                    f32x2 s01 = {0.f, 0.f}, s23 = {0.f, 0.f};   // same fma chain per channel, two channels per instruction
#pragma unroll
                    for (int t = 0; t < 9; ++t) {
                        const float xv = patch[(hy + t / 3) * (HW_ + 2) + hx + t % 3];
                        const f32x4 wv = *(const f32x4*)(fw + t * 32 + c0);
                        const f32x2 x2 = {xv, xv};
                        s01 = og_fma2_first(x2, f32x2{wv.x, wv.y}, s01);
                        s23 = og_fma2_first(x2, f32x2{wv.z, wv.w}, s23);
                    }
                    const f32x4 sc4 = *(const f32x4*)(fw + 288 + c0), sh4 = *(const f32x4*)(fw + 320 + c0);
                    s01 = og_fma2_first(s01, f32x2{sc4.x, sc4.y}, f32x2{sh4.x, sh4.y});
                    s23 = og_fma2_first(s23, f32x2{sc4.z, sc4.w}, f32x2{sh4.z, sh4.w});
                    o.x = fmaxf(s01.x, 0.f);
                    o.y = fmaxf(s01.y, 0.f);
                    o.z = fmaxf(s23.x, 0.f);
                    o.w = fmaxf(s23.y, 0.f);
                }
                *(f32x4*)(halo0 + q * 16) = o;
            }
        }
    }
    og_wait_dma();
    if (st != nullptr && tid == 0) st[7] = __builtin_amdgcn_s_memtime();
    __syncthreads();
    if (st != nullptr && tid == 0) st[1] = __builtin_amdgcn_s_memtime();
    if (a.prio_mode == 3) __builtin_amdgcn_s_setprio(0);

    int step = s_lo;                             // absolute (chunk, tap) index: also the weight block's index
    const int step_end = s_hi;
    // VS: part p of vs_n starts at unit (p * units) / vs_n, units = (chunk, tap) steps (MODE 0), chunks (MODE 1, 2), virtual chunks (MODE 3)
    const int vs_n = VS ? a.vsplit : 1;
    const int vs_units = a.n_chunks * TAPS_;
    // (integer division runs on the vector ALU: readfirstlane brings the wave-uniform quotient back to a scalar, or the loop's
    //  branches and the LDS-DMA's scalar operands would count as divergent)
    int vp = 1, vnext = (VS && vs_n > 1) ? __builtin_amdgcn_readfirstlane(vs_units / vs_n) : -1;
    f32x16 vsum[VS ? MS : 1];
    if (VS) {
#pragma unroll
        for (int m = 0; m < MS; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) vsum[m][r] = 0.f;
    }
    for (int c = c_lo; c < c_hi; ++c) {
#pragma unroll
        for (int t = 0; t < TAPS; ++t) {
            if (VS && ((MODE == 3) ? (t == 0 && c == vnext) : (c * TAPS_ + t == vnext))) {   // a part boundary: fold, restart from zero
#pragma unroll
                for (int m = 0; m < MS; ++m) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        vsum[m][r] += acc[m][r];
                        acc[m][r] = 0.f;
                    }
                }
                ++vp;
                vnext = (vp < vs_n) ? __builtin_amdgcn_readfirstlane((vp * vs_units) / vs_n) : -1;
            }
            if (MODE == 0 && ks_n > 1 && (c * TAPS + t < s_lo || c * TAPS + t >= s_hi)) continue;  // another K part's step
            if (MODE == 3) {  // tap (ty,tx) of the 2x2 kernel meets parity (py,px): zero unless (ty==1 || py==1) and (tx==1 || px==1)
                const int par = c / cpc;
                if (((t >> 1) == 0 && (par & 2) == 0) || ((t & 1) == 0 && (par & 1) == 0)) continue;
            }
            const int stg = (NSTG == 3) ? t % 3 : (step & 1), stg_next = (NSTG == 3) ? (t + 1) % 3 : ((step + 1) & 1);
            if (step + 1 < step_end) stage_w(stg_next, step + 1);

            const unsigned wb = (unsigned)stg * WBYTES;
            const int dy = (MODE == 0) ? t / 3 : (MODE == 3) ? (t >> 1) : 0;
            const int dx = (MODE == 0) ? t % 3 : (MODE == 3) ? (t & 1) : 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                // slot = (2j + lh) ^ swz  ==  ((lh ^ swz) << 4) ^ (j << 5) in bytes
                const f32x4 bv = og_lds_read16(bbase[j] + wb);
#pragma unroll
                for (int m = 0; m < MS; ++m) {
                    const f32x4 av = og_lds_read16(abase[dx][j ^ ((dy & 1) << 1)] + (unsigned)((dy + 2 * m) * (HW_ * 128)));
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc[m], 0, 0, 0);
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc[m], 0, 0, 0);
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc[m], 0, 0, 0);
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc[m], 0, 0, 0);
                }
            }
            og_wait_dma();
            __syncthreads();
            ++step;
        }
        if (c + 1 < c_hi) {  // every read of the halo buffer completed before the barrier above
            stage_halo(0, c + 1);
            og_wait_dma();
            __syncthreads();
        }
    }

    if (st != nullptr && tid == 0) st[2] = __builtin_amdgcn_s_memtime();
    if (a.prio_mode == 3) __builtin_amdgcn_s_setprio(3);
    if (VS && vs_n > 1) {   // the fused reduce's last addition: (sum of the earlier parts) + the last part
#pragma unroll
        for (int m = 0; m < MS; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][r] = vsum[m][r] + acc[m][r];
    }
    // ---- epilogue (all staging buffers are dead behind the last barrier: LDS is scratch now) ----
    if (ks_n > 1) {
        // split-K: raw accumulators, register order, lane-contiguous (256-B stores); k_splitk_epilogue sums the parts
        // in split order and runs the epilogue
        float* pw = a.partial + ((long long)item_id * 4 + wave) * (MS * 16 * 64) + lane;
        if (a.tile_counter == nullptr) {
#pragma unroll
            for (int m = 0; m < MS; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) pw[(m * 16 + r) * 64] = acc[m][r];
        } else {
            // device-scope (write-through) stores: visible to the other XCDs once acknowledged, without flushing an L2
#pragma unroll
            for (int m = 0; m < MS; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) __hip_atomic_store(pw + (m * 16 + r) * 64, acc[m][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (a.tile_counter != nullptr) {
            // Fused reduce: the LAST part of a tile to arrive sums all parts in split order (its own from memory too, so
            // the result does not depend on who is last) and runs the epilogue -- one launch less per split layer.
            // Device-scope release/acquire around the arrival counter: the parts come from other XCDs' L2s.
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's part is written through
            __syncthreads();
            int* const flag = (int*)smem;
            if (tid == 0) *flag = (atomicAdd(a.tile_counter + tile_id, 1) == ks_n - 1) ? 1 : 0;
            __syncthreads();
            if (*(volatile int*)flag) {
#pragma unroll
                for (int m = 0; m < MS; ++m)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
                // parts are fetched PG at a time (all their loads in flight together: one memory round trip per group
                // instead of one per part) and added in split order; PG = 1 on the 16-row tiles keeps that kernel's registers
                constexpr int PG = (MS == 1) ? 4 : (MS == 2) ? 2 : 1;
                for (int sp_ = 0; sp_ < ks_n; sp_ += PG) {
                    float pv[PG][MS * 16];
#pragma unroll
                    for (int g = 0; g < PG; ++g) {
                        const int sg = (sp_ + g < ks_n) ? sp_ + g : sp_;   // tail: re-read a live part, not added below
                        const float* pr = a.partial + (((long long)tile_id * ks_n + sg) * 4 + wave) * (MS * 16 * 64) + lane;
#pragma unroll
                        for (int k = 0; k < MS * 16; ++k) pv[g][k] = __hip_atomic_load(pr + k * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
#pragma unroll
                    for (int g = 0; g < PG; ++g)
                        if (sp_ + g < ks_n) {
#pragma unroll
                            for (int m = 0; m < MS; ++m)
#pragma unroll
                                for (int r = 0; r < 16; ++r) acc[m][r] += pv[g][m * 16 + r];
                        }
                }
                __syncthreads();   // the flag word is part of wave 0's scratch
                unsigned char* const scr = smem + wave * 5120;
                if (a.act == 1) conv_epilogue_b<NT, MODE, TH, 1, false>(a, acc, n_tile, b, ty0, tx0, wm, wn, li, lh, esc, esh, scr);
                else if (a.act == 2 && a.res != nullptr) conv_epilogue_b<NT, MODE, TH, 2, true>(a, acc, n_tile, b, ty0, tx0, wm, wn, li, lh, esc, esh, scr);
                else if (a.act == 2) conv_epilogue_b<NT, MODE, TH, 2, false>(a, acc, n_tile, b, ty0, tx0, wm, wn, li, lh, esc, esh, scr);
                else conv_epilogue_b<NT, MODE, TH, 0, false>(a, acc, n_tile, b, ty0, tx0, wm, wn, li, lh, esc, esh, scr);
                if (tid == 0) a.tile_counter[tile_id] = 0;   // ready for the next launch on this stream
            }
        }
    } else {
        unsigned char* const scr = smem + wave * 5120;
        if (a.act == 1) conv_epilogue_b<NT, MODE, TH, 1, false>(a, acc, n_tile, b, ty0, tx0, wm, wn, li, lh, esc, esh, scr);
        else if (a.act == 2 && a.res != nullptr) conv_epilogue_b<NT, MODE, TH, 2, true>(a, acc, n_tile, b, ty0, tx0, wm, wn, li, lh, esc, esh, scr);
        else if (a.act == 2) conv_epilogue_b<NT, MODE, TH, 2, false>(a, acc, n_tile, b, ty0, tx0, wm, wn, li, lh, esc, esh, scr);
        else conv_epilogue_b<NT, MODE, TH, 0, false>(a, acc, n_tile, b, ty0, tx0, wm, wn, li, lh, esc, esh, scr);
    }
    if (st != nullptr && tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        st[3] = __builtin_amdgcn_s_memtime();
    }
}

// Winograd F(2x2, 3x3) form of the 3x3 conv for the 64-column layers, all arithmetic in f32: per 2x2 output window the
// 4x4 input patch d becomes V = B^T d B (adds only), the 16 entries of V meet the pre-transformed weights U = G g G^T in 16
// independent GEMMs over the input channels (MFMA), and the window's outputs are Y = A^T M A (adds only): 16 multiplies per
// window and channel pair instead of 36 -- 2.25x fewer MFMAs than the direct form for the same result to f32 rounding
// (the transforms re-associate the sums; same fixtures, same tolerance as every other variant).
//   tile     16x16 output pixels = 64 windows (GEMM rows) x 64 output channels; wave (wm, wn) = 32 windows x 32 channels x
//            all 16 positions = 256 accumulator registers -> ONE workgroup per CU, one wave per SIMD
//   LDS      raw 18x18-pixel halo of a 16-channel chunk (20 KB, LDS-DMA) | V [16 positions][64 windows][16 ch] (64 KB) |
//            U ring: 2 groups of 8 positions x [64 cout][16 ch] (2 x 32 KB, LDS-DMA)
//   per chunk  each thread transforms one (window, 4-channel) item: 16 x ds_read_b128 -> 32 vector adds -> 16 x
//            ds_write_b128, its reads taken under the MFMAs of the chunk before; 128 MFMAs per wave
//   epilogue the output transform runs in registers (a lane's 16 positions of one accumulator slot are one window) and
//            hands conv_epilogue_b the direct kernel's accumulator layout: window (m = r & 3, 2 (r >> 2) + lh) of a wave
// NT = 1 (the 32-column layers): the same kernel on a 32x16-pixel tile = 128 windows x 32 channels, wave wm = 32 windows x all
//   16 positions, 8-channel chunks (V stays 64 KB: [16][128 windows][8 ch]); one K step per position instead of two, so the
//   same side work is issued in half the MFMA steps.
template <int NT>
__global__ __launch_bounds__(256, 1) void k_conv_wino(ConvArgs a) {
    constexpr int KC = 8 * NT;                    // input channels per chunk
    constexpr int PB = KC * 4;                    // bytes of one pixel's / window's / output channel's chunk row: 64 | 32
    constexpr int SL = KC / 4;                    // 16-B slots per row: 4 | 2
    constexpr int KS = KC / 8;                    // K steps (of 8 channels = 4 MFMAs) per position: 2 | 1
    constexpr int TH = 32 / NT;                   // tile height in pixels: 16 | 32 (width 16)
    constexpr int RP = 18;                        // pixels per raw row
    constexpr int RAW_PIX = (TH + 2) * RP;
    constexpr int RAW_BYTES = RAW_PIX * PB;       // 20736 | 19584
    constexpr int V_BYTES = 16 * 4096;            // [16 positions][64 | 128 windows][PB]
    constexpr int UP_BYTES = 32 * NT * PB;        // one position's weights: [32 NT cout][PB] = 4096 | 1024
    constexpr int UG_BYTES = 8 * UP_BYTES;        // ring slot: 8 positions
    constexpr int U_IT = UG_BYTES / 4096;         // LDS-DMA instructions per wave and ring slot: 8 | 2
    constexpr int RAW_PIECES = RAW_PIX * SL;
    constexpr int RAW_IT = (RAW_PIECES + 255) / 256;   // 6 | 5
    constexpr int NST = 8 * KS;                   // MFMA steps per group: 16 | 8

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = (NT == 2) ? (wave & 1) : 0, wm = (NT == 2) ? (wave >> 1) : wave;
    const int li = lane & 31, lh = lane >> 5;

    // tile decode: as k_conv_mfma_o (grid.z = frame group x column tile x frame-in-group)
    const int bz = (int)blockIdx.z;
    const int gz = a.zdiv << a.zgroup_shift;
    const int q_ = (gz == 1) ? bz : (int)(((float)bz + 0.5f) * a.zrcp);
    const int rz = bz - q_ * gz;
    const int n_tile = rz >> a.zgroup_shift;
    const int b = (q_ << a.zgroup_shift) + (rz & ((1 << a.zgroup_shift) - 1));
    if (b >= a.frames) return;
    const int ty0 = (int)blockIdx.y * TH, tx0 = (int)blockIdx.x * 16;
    // diagnostic timeline (og_unet_clock_probe only; nullptr on every product path): 8 stamps for the first 511 workgroups
    unsigned long long* st = nullptr;
    if (a.stamps != nullptr) {
        const unsigned wg = ((unsigned)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        st = a.stamps + 8ull * (wg < 511u ? wg : 511u);
        if (tid == 0) st[0] = __builtin_amdgcn_s_memtime();
    }

    const og_i32x4 in_rsrc = og_make_rsrc(a.in + (long long)b * a.in_frame_stride + a.in_ch_off,
                                          (unsigned)(a.in_frame_stride - a.in_ch_off) * 4u);
    // raw halo image: pixel (hy, hx) at index hy * RP + (hx & 1) * 9 + (hx >> 1) (even and odd columns apart: the windows of
    // a row then read consecutive pixels), SL pieces of 16 B per pixel
    unsigned hoff[RAW_IT];
    {   // piece id = it * 256 + tid walks in steps of 256 pieces = 256 / SL pixels = QY rows + QR pixels of the RP-pixel image rows: one
        // division for the first piece, then adds (each constant division is ~5 vector-ALU instructions on the way to the first DMA)
        constexpr int QY = (256 / SL) / RP, QR = (256 / SL) % RP;
        const int pc = tid % SL;
        int hy = (tid / SL) / RP, r = (tid / SL) - hy * RP;
#pragma unroll
        for (int it = 0; it < RAW_IT; ++it) {
            const int hx = (r >= 9) ? 2 * (r - 9) + 1 : 2 * r;
            const int gy = ty0 + hy - 1, gx = tx0 + hx - 1;
            const bool inb = it * 256 + tid < RAW_PIECES && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
            hoff[it] = inb ? (unsigned)((gy * a.W + gx) * a.in_pix_stride * 4 + pc * 16) : OG_OOB;
            hy += QY;
            r += QR;
            if (r >= RP) { r -= RP; hy += 1; }
        }
    }
    const bool last_valid = ((RAW_IT - 1) * 256 + tid) < RAW_PIECES;
    const unsigned lds0 = og_lds_addr(smem);
    const int n_ck = a.n_chunks * (32 / KC);
    const og_i32x4 w_rsrc = og_make_rsrc(a.wpk + (long long)n_tile * n_ck * (16 * UP_BYTES / 4), (unsigned)n_ck * (16u * UP_BYTES));
    auto u_piece = [&](int stage, int grp, int i) {
        const unsigned base = __builtin_amdgcn_readfirstlane(lds0 + RAW_BYTES + V_BYTES + stage * UG_BYTES + wave * 1024);
        glds16b((unsigned)tid * 16u, w_rsrc, (unsigned)grp * UG_BYTES + i * 4096, base + i * 4096);
    };
    auto raw_piece = [&](int ck, int it) {
        const unsigned base = __builtin_amdgcn_readfirstlane(lds0 + wave * 1024);
        if (it < RAW_IT - 1 || last_valid) glds16b(hoff[it], in_rsrc, (unsigned)ck * PB, base + it * 4096);
    };
#pragma unroll
    for (int it = 0; it < RAW_IT; ++it) raw_piece(0, it);
#pragma unroll
    for (int i = 0; i < U_IT; ++i) u_piece(0, 0, i);
    if (st != nullptr && tid == 0) st[1] = __builtin_amdgcn_s_memtime();   // first DMAs issued

    // Fragment addressing.  MFMA row i of a wave = window (row 4 wm + (i & 3), column wx = 2 (i >> 3) + ((i >> 2) & 1)).
    // NT 2: window w's 64-B row at w * 64, slot s at s ^ (w >> 3 & 3): the 16 lanes of a read hit 16 distinct 16-B bank groups.
    // NT 1: 32-B rows, so the window ORDER carries half of the spreading: row index 8 (4 wm + (wx >> 1)) + (i & 3) + 4 (wx & 1),
    //       slot s at s ^ (wx >> 1 & 1).  Weights: output channel n's row, slot s at s ^ (n >> 2 & 3) | s ^ (n >> 3 & 1).
    const int wxa = 2 * (li >> 3) + ((li >> 2) & 1);
    const int nb = wn * 32 + li;
    unsigned abase[KS], bbase[KS];
#pragma unroll
    for (int j = 0; j < KS; ++j) {
        if (NT == 2) {
            abase[j] = lds0 + RAW_BYTES + ((4 * wm + (li & 3)) * 8 + wxa) * 64 + (((2 * j + lh) ^ (li & 3)) << 4);
            bbase[j] = lds0 + RAW_BYTES + V_BYTES + nb * 64 + (((2 * j + lh) ^ ((nb >> 2) & 3)) << 4);
        } else {
            abase[j] = lds0 + RAW_BYTES + (8 * (4 * wm + (wxa >> 1)) + (li & 3) + 4 * (wxa & 1)) * 32 + ((lh ^ ((wxa >> 1) & 1)) << 4);
            bbase[j] = lds0 + RAW_BYTES + V_BYTES + nb * 32 + ((lh ^ ((nb >> 3) & 1)) << 4);
        }
        asm volatile("" : "+v"(abase[j]));
        asm volatile("" : "+v"(bbase[j]));
    }
    // transform role: window (row wr, column wc), channels 4 qc .. 4 qc + 3 of the chunk
    // (NT 1: lane bits = quad, row bits 1:0, column bit 0, column bits 2:1: the 8 lanes of a ds_write_b128 service group then write
    //  128 contiguous bytes of V (round 2's order -- column bit 0 below the row bits -- put lanes 0-3 and 4-7 on the same 32 banks:
    //  2.2e7 SQ_LDS_BANK_CONFLICT cycles per launch, all from these writes), and the 16-lane groups of the raw reads stay conflict-free)
    const int qc = tid % SL;
    const int wr = (NT == 2) ? (tid >> 5) : 4 * (tid >> 6) + ((tid >> 1) & 3);
    const int wc = (NT == 2) ? ((tid >> 2) & 7) : ((tid >> 3) & 1) + 2 * ((tid >> 4) & 3);
    const int wt = 8 * wr + wc;
    unsigned rbase = lds0 + (unsigned)((2 * wr * RP + wc) * PB + qc * 16);
    unsigned vwbase = (NT == 2) ? lds0 + RAW_BYTES + (unsigned)(wt * 64 + ((qc ^ (wr & 3)) << 4))
                                : lds0 + RAW_BYTES + (unsigned)((8 * (4 * (wr >> 2) + (wc >> 1)) + (wr & 3) + 4 * (wc & 1)) * 32 + ((qc ^ ((wc >> 1) & 1)) << 4));
    asm volatile("" : "+v"(rbase));
    asm volatile("" : "+v"(vwbase));

    const int ecol = n_tile * 32 * NT + wn * 32 + li;
    const float esc = a.scale[ecol], esh = a.shift[ecol];

    f32x16 acc[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
        // pin the 256 v_accvgpr_write of the zeroing HERE, under the latency of the first DMAs: hipcc otherwise sinks them to the
        // first MFMA, i.e. into the main loop of the first chunk (timeline: the first halo takes 1.3 - 9 k cycles to land)
        asm volatile("" : "+a"(acc[k]));
    }

    // One wave per SIMD: nothing but this wave's own instruction stream can fill the matrix pipe's shadow, so everything that
    // is not an MFMA is cut into micro-ops and issued between the four MFMAs of a step (64 cycles each), in a fixed order:
    //   group g = 0 of chunk c (positions 0-7):  the hi half of V (positions 8-15) of chunk c from the registers `tv`; LDS-DMA of
    //                                            the next U group and of the raw halo of chunk c + 1
    //   group g = 1 (positions 8-15):            LDS-DMA of the next U group; raw halo of chunk c + 1 -> registers -> B^T d B ->
    //                                            `tv`; its lo half (positions 0-7) into V
    // Two barriers per chunk (one per group): every LDS region is written in the group after its last reader's group.
    f32x4 d[16], t[16], tv[16];
    auto raw_read = [&](int n) {   // pixel (2 wr + i, 2 wc + j) of the halo: index (2 wr + i) * RP + (j & 1) * 9 + wc + (j >> 1)
        const int i = n >> 2, j = n & 3;
        d[n] = og_lds_read16(rbase + (unsigned)((i * RP + (j & 1) * 9 + (j >> 1)) * PB));
    };
    auto row_op = [&](int n) {   // V = B^T d B, B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]: rows first (n = 4 * row + j)
        const int r = n >> 2, j = n & 3;
        t[n] = (r == 0) ? og_sub4(d[0 + j], d[8 + j]) : (r == 1) ? d[4 + j] + d[8 + j] : (r == 2) ? og_sub4(d[8 + j], d[4 + j]) : og_sub4(d[4 + j], d[12 + j]);
    };
    auto col_op = [&](int n) {   // then columns (n = 4 * i + column)
        const int i = n >> 2, cc = n & 3;
        tv[n] = (cc == 0) ? og_sub4(t[4 * i + 0], t[4 * i + 2]) : (cc == 1) ? t[4 * i + 1] + t[4 * i + 2] : (cc == 2) ? og_sub4(t[4 * i + 2], t[4 * i + 1])
                                                                                                                     : og_sub4(t[4 * i + 1], t[4 * i + 3]);
    };
    auto v_write = [&](int k) { *(OG_LDS_AS f32x4*)(unsigned long long)(vwbase + (unsigned)(k * 4096)) = tv[k]; };
    // micro-op n of the transform pipeline of a group-1 (32 raw reads are 16: reads 0-15, row ops 16-31, column ops 32-47, lo writes 48-55)
    auto xf_op = [&](int n) {
        if (n < 16) raw_read(n);
        else if (n < 32) row_op(n - 16);
        else if (n < 48) col_op(n - 32);
        else if (n < 56) v_write(n - 48);
    };
    auto xf4 = [&](int n) { xf_op(n); xf_op(n + 1); xf_op(n + 2); xf_op(n + 3); };

    og_wait_dma();
    if (st != nullptr && tid == 0) st[2] = __builtin_amdgcn_s_memtime();   // first halo and weight group landed
    __syncthreads();
#pragma unroll
    for (int n = 0; n < 48; ++n) xf_op(n);
#pragma unroll
    for (int k = 0; k < 8; ++k) v_write(k);
    __syncthreads();   // lo half of V (chunk 0) complete; every read of the raw buffer done
    if (st != nullptr && tid == 0) st[3] = __builtin_amdgcn_s_memtime();   // main loop starts

    for (int c = 0; c < n_ck; ++c) {
        const bool nxt = c + 1 < n_ck;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const int grp = 2 * c + g;
            const bool more_u = grp + 1 < 2 * n_ck;
            f32x4 fa[2], fb[2];
            fa[0] = og_lds_read16(abase[0] + (unsigned)(8 * g * 4096));
            fb[0] = og_lds_read16(bbase[0] + (unsigned)(g * UG_BYTES));
#pragma unroll
            for (int st = 0; st < NST; ++st) {
                const int k = 8 * g + st / KS;
                const f32x4 av = fa[st & 1], bv = fb[st & 1];
                acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc[k], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (st + 1 < NST) {   // slot 0: the fragments of the next step
                    const int kn = (st + 1) / KS, jn = (st + 1) % KS;
                    fa[(st + 1) & 1] = og_lds_read16(abase[jn] + (unsigned)((8 * g + kn) * 4096));
                    fb[(st + 1) & 1] = og_lds_read16(bbase[jn] + (unsigned)(g * UG_BYTES + kn * UP_BYTES));
                }
                __builtin_amdgcn_sched_barrier(0);
                acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc[k], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                // slot 1
                if (g == 0) {
                    if (NT == 2) { if (st < 8) v_write(8 + st); }                 // hi half of this chunk's V
                    else v_write(8 + st);
                } else if (nxt) {
                    if (NT == 2) {   // reads in steps 0-1, row ops 3-6, column ops 7-10, lo writes 11-14
                        if (st < 2) xf4(8 * st);
                        else if (st >= 3 && st < 7) { xf_op(16 + 4 * (st - 3)); xf_op(16 + 4 * (st - 3) + 1); }
                        else if (st >= 7 && st < 11) { xf_op(32 + 4 * (st - 7)); xf_op(32 + 4 * (st - 7) + 1); }
                        else if (st >= 11 && st < 15) xf_op(48 + 2 * (st - 11));
                    } else {         // reads in steps 0-1, row ops 2-3, column ops 4-5, lo writes 6-7
                        if (st < 2) xf4(8 * st);
                        else if (st < 4) xf4(16 + 8 * (st - 2));
                        else if (st < 6) xf4(32 + 8 * (st - 4));
                        else { xf_op(48 + 4 * (st - 6)); xf_op(48 + 4 * (st - 6) + 1); }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc[k], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                // slot 2
                if (st < U_IT && more_u) u_piece(g ^ 1, grp + 1, st);
                if (g == 1 && nxt) {
                    if (NT == 2) {
                        if (st < 2) xf4(8 * st + 4);
                        else if (st >= 3 && st < 7) { xf_op(16 + 4 * (st - 3) + 2); xf_op(16 + 4 * (st - 3) + 3); }
                        else if (st >= 7 && st < 11) { xf_op(32 + 4 * (st - 7) + 2); xf_op(32 + 4 * (st - 7) + 3); }
                        else if (st >= 11 && st < 15) xf_op(48 + 2 * (st - 11) + 1);
                    } else {
                        if (st < 2) xf4(8 * st + 4);
                        else if (st < 4) xf4(16 + 8 * (st - 2) + 4);
                        else if (st < 6) xf4(32 + 8 * (st - 4) + 4);
                        else { xf_op(48 + 4 * (st - 6) + 2); xf_op(48 + 4 * (st - 6) + 3); }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc[k], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                // slot 3
                if (g == 0 && nxt && st < RAW_IT) raw_piece(c + 1, st);   // early in the group: waited at its end
                __builtin_amdgcn_sched_barrier(0);
            }
            og_wait_dma();
            __syncthreads();
        }
    }

    if (st != nullptr && tid == 0) st[4] = __builtin_amdgcn_s_memtime();   // main loop done
    // ---- output transform Y = A^T M A, A^T = [1 1 1 0; 0 1 -1 -1], in registers; then the shared epilogue ----
    f32x16 o[4];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float tm[2][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            tm[0][j] = acc[0 + j][r] + acc[4 + j][r] + acc[8 + j][r];
            tm[1][j] = acc[4 + j][r] - acc[8 + j][r] - acc[12 + j][r];
        }
#pragma unroll
        for (int y = 0; y < 2; ++y) {
            o[r & 3][4 * (r >> 2) + 2 * y + 0] = tm[y][0] + tm[y][1] + tm[y][2];
            o[r & 3][4 * (r >> 2) + 2 * y + 1] = tm[y][1] - tm[y][2] - tm[y][3];
        }
    }
    if (st != nullptr && tid == 0) st[5] = __builtin_amdgcn_s_memtime();   // output transform done (wave 0)
    unsigned char* const scr = smem + wave * 5120;   // the raw buffer is dead (>= 19 KB; the epilogue's scratch runs into V, dead too)
    if (a.act == 1) conv_epilogue_b<NT, 0, TH, 1, false>(a, o, n_tile, b, ty0, tx0, wm, wn, li, lh, esc, esh, scr);
    else conv_epilogue_b<NT, 0, TH, 0, false>(a, o, n_tile, b, ty0, tx0, wm, wn, li, lh, esc, esh, scr);
    if (st != nullptr && tid == 0) {
        st[6] = __builtin_amdgcn_s_memtime();                               // epilogue issued
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        st[7] = __builtin_amdgcn_s_memtime();                               // stores acknowledged
    }
}

// 16-byte device-scope (sc1: written through / read past this XCD's L2) store and load for the exchange of raw accumulators
// between workgroups that may sit on different XCDs -- the 128-bit form of what __hip_atomic_store / __hip_atomic_load at agent
// scope compile to (global_store_dword ... sc1).  The store is followed by one wait state for the same reason as
// og_buffer_store16 (hipcc cannot see the data registers of an inline-asm store being reused).
__device__ __forceinline__ void og_store16_dev(float* p, f32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 0" : : "v"(p), "v"(v) : "memory");
}
// The matching load goes through the compiler's own buffer-load builtin with the sc1 cache-policy bit (aux 16 on gfx94x/95x):
// hipcc then tracks the outstanding load itself (an inline-asm load returns "immediately" as far as the register allocator
// is concerned, and anything it does with the destination registers before our s_waitcnt is undefined behaviour).
__device__ __forceinline__ f32x4 og_load16_dev(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 16));
}

// POSITION-SPLIT form of k_conv_wino for launches that cannot fill the chip (one frame per kernel chain: the 16x16 map of the
// bottleneck is ONE tile x 8 column tiles = 8 workgroups of k_conv_wino on 256 CUs, each with a serial loop of 32 channel
// chunks x 128 MFMAs).  The 16 Winograd positions of a tile are 16 INDEPENDENT GEMMs over the input channels, so they can go
// to different workgroups without touching any sum: workgroup (tile, column tile, position group pg) runs positions
// pg PN .. pg PN + PN - 1 -- for each the same V entries (same adds in the same order), the same U entries and the same MFMA
// sequence (chunk by chunk, k pairs in the same order) as k_conv_wino -- and writes its raw accumulators to a.partial; the
// LAST workgroup of a tile to arrive (a.tile_counter) reads all 16 positions back, runs k_conv_wino's output transform and the
// shared epilogue.  Every accumulator holds the bits k_conv_wino computes, so the result is BIT-IDENTICAL to it whatever
// arrives first: 16 / PN times the workgroups, no change of arithmetic -- unlike split-K, which re-associates the channel sum.
//   V for one position (i, cc) needs only rows (ra, rb) and columns (ca, cb) of the 4x4 patch: B^T d B is, per axis,
//   {x0 - x2, x1 + x2, x2 - x1, x1 - x3}; a workgroup's PN <= 4 positions share the row i, so it reads 8 of the 16 patch pixels.
//   LDS: raw halo ring LA x 24 KB | V 2 x PN x 4 KB | U (LA + 1) x max(PN x UP, 4 KB).  Stage c + LA is fetched (LDS-DMA, counted
//   vmcnt) while chunk c multiplies and chunk c + 1 is transformed; two barriers per chunk; one workgroup per CU.
template <int NT, int PN>
__global__ __launch_bounds__(256, 1) void k_conv_wino_ps(ConvArgs a) {
    static_assert(PN == 1 || PN == 2 || PN == 4, "a workgroup's positions share one row of the 4x4 position grid");
    constexpr int KC = 8 * NT;
    constexpr int PB = KC * 4;
    constexpr int SL = KC / 4;
    constexpr int KS = KC / 8;
    constexpr int TH = 32 / NT;
    constexpr int RP = 18;
    constexpr int RAW_PIX = (TH + 2) * RP;
    constexpr int RAW_PIECES = RAW_PIX * SL;
    constexpr int RAW_IT = (RAW_PIECES + 255) / 256;   // 6 | 5
    constexpr int RAW_PAD = RAW_IT * 4096;             // every lane of every DMA instruction lands inside its buffer
    constexpr int UP_BYTES = 32 * NT * PB;             // one position's weights: 4096 | 1024
    constexpr int U_STAGE = (PN * UP_BYTES > 4096) ? PN * UP_BYTES : 4096;
    constexpr int U_IT = U_STAGE / 4096;               // DMA instructions per wave and stage (all four waves issue all of them)
    constexpr int NDMA = RAW_IT + U_IT;                // vector-memory instructions per wave and stage: what s_waitcnt counts
    constexpr int NPG = 16 / PN;
    // look-ahead: stage c + LA is requested while chunk c multiplies (measured with LA = 2: 0.76 us per chunk against ~0.35 us of
    // work -- the loop waits for memory latency, one workgroup per CU has nobody else to hide it); PN = 4 has no LDS left for 3
    constexpr int LA = (PN <= 2) ? 3 : 2;
    constexpr int RS = LA, US = LA + 1;                // raw / U ring depths
    constexpr int VB = RS * RAW_PAD;
    constexpr int UB = VB + 2 * PN * 4096;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = (NT == 2) ? (wave & 1) : 0, wm = (NT == 2) ? (wave >> 1) : wave;
    const int li = lane & 31, lh = lane >> 5;

    // grid.z = k_conv_wino's z (frame group x column tile x frame-in-group) x position group
    const int bzf = (int)blockIdx.z;
    const int pg = bzf % NPG, bz = bzf / NPG;
    const int gz = a.zdiv << a.zgroup_shift;
    const int q_ = (gz == 1) ? bz : (int)(((float)bz + 0.5f) * a.zrcp);
    const int rz = bz - q_ * gz;
    const int n_tile = rz >> a.zgroup_shift;
    const int b = (q_ << a.zgroup_shift) + (rz & ((1 << a.zgroup_shift) - 1));
    if (b >= a.frames) return;   // every position group of such a tile returns: nobody waits for it
    // diagnostic timeline (og_unet_clock_probe only; nullptr on every product path): entry, loop start, loop end, exit
    unsigned long long* const st = a.stamps ? a.stamps + 4ull * (((unsigned)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x < 1023u
                                                                    ? ((unsigned)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x : 1023u) : nullptr;
    if (st != nullptr && tid == 0) st[0] = __builtin_amdgcn_s_memtime();
    const int ty0 = (int)blockIdx.y * TH, tx0 = (int)blockIdx.x * 16;
    const int tile_id = ((b * a.tiles_y + (int)blockIdx.y) * a.tiles_x + (int)blockIdx.x) * a.zdiv + n_tile;
    const int p0 = pg * PN;
    const int pi = p0 >> 2, pc0 = p0 & 3;                 // row of the position grid, first column
    const int ra = (pi == 0) ? 0 : (pi == 2) ? 2 : 1;     // B^T row pi = x[ra] (+ if pi == 1, else -) x[rb]
    const int rb = (pi == 0) ? 2 : (pi == 1) ? 2 : (pi == 2) ? 1 : 3;
    const bool rplus = pi == 1;

    const og_i32x4 in_rsrc = og_make_rsrc(a.in + (long long)b * a.in_frame_stride + a.in_ch_off,
                                          (unsigned)(a.in_frame_stride - a.in_ch_off) * 4u);
    unsigned hoff[RAW_IT];   // as k_conv_wino: pixel (hy, hx) at index hy * RP + (hx & 1) * 9 + (hx >> 1)
    {   // incremental walk of the piece id, as in k_conv_wino
        constexpr int QY = (256 / SL) / RP, QR = (256 / SL) % RP;
        const int pc = tid % SL;
        int hy = (tid / SL) / RP, r = (tid / SL) - hy * RP;
#pragma unroll
        for (int it = 0; it < RAW_IT; ++it) {
            const int hx = (r >= 9) ? 2 * (r - 9) + 1 : 2 * r;
            const int gy = ty0 + hy - 1, gx = tx0 + hx - 1;
            const bool inb = it * 256 + tid < RAW_PIECES && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
            hoff[it] = inb ? (unsigned)((gy * a.W + gx) * a.in_pix_stride * 4 + pc * 16) : OG_OOB;
            hy += QY;
            r += QR;
            if (r >= RP) { r -= RP; hy += 1; }
        }
    }
    const unsigned lds0 = og_lds_addr(smem);
    const int n_ck = a.n_chunks * (32 / KC);
    const og_i32x4 w_rsrc = og_make_rsrc(a.wpk + (long long)n_tile * n_ck * (16 * UP_BYTES / 4), (unsigned)n_ck * (16u * UP_BYTES));
    auto stage = [&](int ck) {   // raw halo of chunk ck -> raw[ck % RS], U of (chunk ck, positions p0 ..) -> U[ck % US]
        const unsigned rbase_ = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)((ck % RS) * RAW_PAD) + wave * 1024);
#pragma unroll
        for (int it = 0; it < RAW_IT; ++it) glds16b(hoff[it], in_rsrc, (unsigned)ck * PB, rbase_ + it * 4096);
        const unsigned ubase_ = __builtin_amdgcn_readfirstlane(lds0 + UB + (unsigned)((ck % US) * U_STAGE) + wave * 1024);
#pragma unroll
        for (int i = 0; i < U_IT; ++i)   // (PN x UP < 4 KB: the tail of the 4 KB belongs to the next positions / chunk, or reads as zeros)
            glds16b((unsigned)tid * 16u, w_rsrc, (unsigned)((ck * 16 + p0) * UP_BYTES + i * 4096), ubase_ + i * 4096);
    };
#pragma unroll
    for (int i = 0; i < LA; ++i)
        if (i < n_ck) stage(i);

    // fragment / transform addressing: k_conv_wino's, with V and U counted from the start of a position's 4 KB / UP block
    const int wxa = 2 * (li >> 3) + ((li >> 2) & 1);
    const int nb = wn * 32 + li;
    unsigned abase[KS], bbase[KS];
#pragma unroll
    for (int j = 0; j < KS; ++j) {
        if (NT == 2) {
            abase[j] = lds0 + VB + ((4 * wm + (li & 3)) * 8 + wxa) * 64 + (((2 * j + lh) ^ (li & 3)) << 4);
            bbase[j] = lds0 + UB + nb * 64 + (((2 * j + lh) ^ ((nb >> 2) & 3)) << 4);
        } else {
            abase[j] = lds0 + VB + (8 * (4 * wm + (wxa >> 1)) + (li & 3) + 4 * (wxa & 1)) * 32 + ((lh ^ ((wxa >> 1) & 1)) << 4);
            bbase[j] = lds0 + UB + nb * 32 + ((lh ^ ((nb >> 3) & 1)) << 4);
        }
    }
    const int qc = tid % SL;
    const int wr = (NT == 2) ? (tid >> 5) : 4 * (tid >> 6) + ((tid >> 1) & 3);
    const int wc = (NT == 2) ? ((tid >> 2) & 7) : ((tid >> 3) & 1) + 2 * ((tid >> 4) & 3);
    const int wt = 8 * wr + wc;
    const unsigned rbase = lds0 + (unsigned)((2 * wr * RP + wc) * PB + qc * 16);
    const unsigned vwbase = (NT == 2) ? lds0 + VB + (unsigned)(wt * 64 + ((qc ^ (wr & 3)) << 4))
                                      : lds0 + VB + (unsigned)((8 * (4 * (wr >> 2) + (wc >> 1)) + (wr & 3) + 4 * (wc & 1)) * 32 + ((qc ^ ((wc >> 1) & 1)) << 4));

    int issued = ((n_ck < LA) ? n_ck : LA) - 1;   // newest stage requested so far
    auto wait_landed = [&](int ck) {   // stage ck has landed; the `issued - ck` younger ones may still be in flight (vmcnt retires in order)
        const int newer = issued - ck;
        if (LA >= 3 && newer >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NDMA) : "memory");
        else if (newer == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
        else og_wait_dma();
    };
    // transform of chunk ck, in two halves so that the LDS latency of its reads hides under the MFMAs of chunk ck - 1.
    // B^T d B per axis is x[ia] +- x[ib] with (ia, ib, sign) = (0,2,-) (1,2,+) (2,1,-) (1,3,-): the sign rides as a +-1 multiplier in
    // ONE fma -- fma(+-1, b, a) rounds a +- b once, exactly like the add / subtract k_conv_wino issues -- so a runtime position
    // costs no select; PN = 1 reads just the 4 patch pixels (rows ra, rb x columns ca, cb) of its position, PN = 2 the 3 columns
    // of its two positions, PN = 4 all four.  (Measured before: the transform was 8.7 of the 22.8 us of the deepest layer's loop.)
    const float rsgn = rplus ? 1.0f : -1.0f;
    constexpr int NCOL = (PN == 1) ? 2 : (PN == 2) ? 3 : 4;
    int colj[NCOL];   // patch columns read, in register order
    if (PN == 1) {
        colj[0] = (pc0 == 0) ? 0 : (pc0 == 2) ? 2 : 1;                    // ca
        colj[1] = (pc0 == 0) ? 2 : (pc0 == 1) ? 2 : (pc0 == 2) ? 1 : 3;    // cb
    } else if (PN == 2) {   // positions (0,1): columns 0,1,2; positions (2,3): columns 1,2,3
        colj[0] = pc0 ? 1 : 0;
        colj[1] = pc0 ? 2 : 1;
        colj[NCOL - 1] = pc0 ? 3 : 2;
    } else {
#pragma unroll
        for (int j = 0; j < NCOL; ++j) colj[j] = j;
    }
    f32x4 xa[NCOL], xb[NCOL];
    auto tr_read = [&](int ck) {   // patch pixel (row, j) of the window: index (2 wr + row) * RP + (j & 1) * 9 + wc + (j >> 1)
        const unsigned rb_ = rbase + (unsigned)((ck % RS) * RAW_PAD);
#pragma unroll
        for (int n = 0; n < NCOL; ++n) {
            const int j = colj[n];
            xa[n] = og_lds_read16(rb_ + (unsigned)((ra * RP + (j & 1) * 9 + (j >> 1)) * PB));
            xb[n] = og_lds_read16(rb_ + (unsigned)((rb * RP + (j & 1) * 9 + (j >> 1)) * PB));
        }
    };
    auto fma4 = [](float sgn, f32x4 b_, f32x4 a_) {   // sgn * b + a as two v_pk_fma_f32 (hipcc emits four v_fma_f32 for it; same fused rounding)
        const f32x2 s2 = {sgn, sgn};
        f32x2 lo, hi;
        asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(lo) : "v"(s2), "v"(f32x2{b_.x, b_.y}), "v"(f32x2{a_.x, a_.y}));
        asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(hi) : "v"(s2), "v"(f32x2{b_.z, b_.w}), "v"(f32x2{a_.z, a_.w}));
        return f32x4{lo.x, lo.y, hi.x, hi.y};
    };
    auto tr_write = [&](int ck) {   // -> V[ck & 1][0 .. PN)
        f32x4 t[NCOL];
#pragma unroll
        for (int n = 0; n < NCOL; ++n) t[n] = fma4(rsgn, xb[n], xa[n]);
#pragma unroll
        for (int k = 0; k < PN; ++k) {
            f32x4 v;
            if (PN == 1) v = fma4((pc0 == 1) ? 1.0f : -1.0f, t[1], t[0]);
            else if (PN == 2) {   // t = columns (0,1,2) -> c0 = t0 - t2, c1 = t1 + t2 | columns (1,2,3) -> c2 = t2 - t1 = t[1] - t[0], c3 = t1 - t3 = t[0] - t[2]
                if (k == 0) v = pc0 ? fma4(-1.0f, t[0], t[1]) : fma4(-1.0f, t[2], t[0]);
                else v = pc0 ? fma4(-1.0f, t[2], t[0]) : fma4(1.0f, t[2], t[1]);
            } else v = (k == 0) ? og_sub4(t[0], t[2]) : (k == 1) ? og_add4(t[1], t[2]) : (k == 2) ? og_sub4(t[2], t[1]) : og_sub4(t[1], t[3]);
            *(OG_LDS_AS f32x4*)(unsigned long long)(vwbase + (unsigned)(((ck & 1) * PN + k) * 4096)) = v;
        }
    };

    f32x16 acc[PN];
#pragma unroll
    for (int k = 0; k < PN; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;

    wait_landed(0);
    __syncthreads();
    tr_read(0);
    tr_write(0);
    if (n_ck > 1) wait_landed(1);
    if (st != nullptr && tid == 0) st[1] = __builtin_amdgcn_s_memtime();

    // ONE barrier per chunk: behind it V[c & 1] is complete, the raw halo of chunk c + 1 has landed for every wave, and the
    // buffers of stage c + LA (raw[c % RS]: transformed an iteration ago; U[(c - 1) % US]: multiplied an iteration ago) are free
    for (int c = 0; c < n_ck; ++c) {
        __syncthreads();
        if (c + LA < n_ck) {
            stage(c + LA);
            issued = c + LA;
        }
        if (c + 1 < n_ck) tr_read(c + 1);
        const unsigned vo = (unsigned)((c & 1) * PN * 4096), uo = (unsigned)((c % US) * U_STAGE);
        // fragments of step s + 1 are requested before the MFMAs of step s (measured without: 770 cycles per position and chunk
        // for 512 cycles of MFMA -- the LDS latency of every fragment pair was exposed)
        f32x4 fa[2], fb[2];
        fa[0] = og_lds_read16(abase[0] + vo);
        fb[0] = og_lds_read16(bbase[0] + uo);
#pragma unroll
        for (int s_ = 0; s_ < PN * KS; ++s_) {
            const int k = s_ / KS;
            if (s_ + 1 < PN * KS) {
                const int kn = (s_ + 1) / KS, jn = (s_ + 1) % KS;
                fa[(s_ + 1) & 1] = og_lds_read16(abase[jn] + vo + (unsigned)(kn * 4096));
                fb[(s_ + 1) & 1] = og_lds_read16(bbase[jn] + uo + (unsigned)(kn * UP_BYTES));
            }
            const f32x4 av = fa[s_ & 1], bv = fb[s_ & 1];
            __builtin_amdgcn_sched_barrier(0);
            acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc[k], 0, 0, 0);
            acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc[k], 0, 0, 0);
            acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc[k], 0, 0, 0);
            acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc[k], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (c + 1 < n_ck) tr_write(c + 1);
        if (c + 2 < n_ck) wait_landed(c + 2);
    }

    if (st != nullptr && tid == 0) st[2] = __builtin_amdgcn_s_memtime();
    // ---- raw accumulators out (16 B per lane and store, device scope), arrival, and for the last arriver: k_conv_wino's tail ----
    float* const part = a.partial + (long long)tile_id * (16 * 4 * 1024) + wave * 1024 + lane * 4;
#pragma unroll
    for (int k = 0; k < PN; ++k)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            og_store16_dev(part + (p0 + k) * 4096 + q * 256, f32x4{acc[k][4 * q], acc[k][4 * q + 1], acc[k][4 * q + 2], acc[k][4 * q + 3]});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's part is written through
    __syncthreads();
    int* const flag = (int*)smem;
    if (tid == 0) *flag = (atomicAdd(a.tile_counter + tile_id, 1) == NPG - 1) ? 1 : 0;
    __syncthreads();
    if (*(volatile int*)flag == 0) {
        if (st != nullptr && tid == 0) st[3] = __builtin_amdgcn_s_memtime();
        return;
    }
    __syncthreads();   // the flag word is part of wave 0's epilogue scratch

    const int ecol = n_tile * 32 * NT + wn * 32 + li;
    const float esc = a.scale[ecol], esh = a.shift[ecol];
    const __amdgpu_buffer_rsrc_t part_rs = og_rsrc(a.partial + (long long)tile_id * (16 * 4 * 1024), 16u * 4u * 1024u * 4u);
    const unsigned part_voff = (unsigned)((wave * 1024 + lane * 4) * 4);
    f32x16 o[4];
    {   // all 16 positions x 16 accumulator registers of this lane (its own part from memory too): 64 loads in flight together --
        // ONE memory round trip for the tile (the wave has 512 registers to itself)
        f32x4 mm[4][16];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int k = 0; k < 16; ++k) mm[q][k] = og_load16_dev(part_rs, part_voff, (unsigned)((k * 4096 + q * 256) * 4));
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4* const m = mm[q];
#pragma unroll
            for (int e = 0; e < 4; ++e) {   // Y = A^T M A exactly as k_conv_wino writes it (register r = 4q + e)
                float tm[2][4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    tm[0][j] = m[0 + j][e] + m[4 + j][e] + m[8 + j][e];
                    tm[1][j] = m[4 + j][e] - m[8 + j][e] - m[12 + j][e];
                }
#pragma unroll
                for (int y = 0; y < 2; ++y) {
                    o[e][4 * q + 2 * y + 0] = tm[y][0] + tm[y][1] + tm[y][2];
                    o[e][4 * q + 2 * y + 1] = tm[y][1] - tm[y][2] - tm[y][3];
                }
            }
        }
    }
    unsigned char* const scr = smem + wave * 5120;
    if (a.act == 1) conv_epilogue_b<NT, 0, TH, 1, false>(a, o, n_tile, b, ty0, tx0, wm, wn, li, lh, esc, esh, scr);
    else conv_epilogue_b<NT, 0, TH, 0, false>(a, o, n_tile, b, ty0, tx0, wm, wn, li, lh, esc, esh, scr);
    if (tid == 0) a.tile_counter[tile_id] = 0;   // ready for the next launch on this stream
    if (st != nullptr && tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        st[3] = __builtin_amdgcn_s_memtime() | (1ull << 63);   // top bit: this workgroup reduced its tile
    }
}

// WAVE-OWNS-POSITIONS form of k_conv_wino for launches that cannot fill the chip with 64-window tiles (one frame per kernel
// chain, BASELINE configs[1] / utils.py:235-237): the 16 Winograd positions of a tile are split over the four WAVES of one
// workgroup instead of over workgroups (k_conv_wino_ps), so the raw accumulators meet in LDS, not in memory -- no 256 KB
// round trip through the split-K workspace, no arrival counter, no serial reducer.
//   tile      8 WB x 16 output pixels = WB blocks of 32 windows (4 window rows x 8 window columns) x 32 output channels;
//             grid = (W / 16, H / (8 WB), frames x Cout / 32): 4 (WB 1) or 2 (WB 2) times k_conv_wino<2>'s workgroups
//   wave w    row w of the 4x4 position grid: positions 4w .. 4w + 3 = 4 WB accumulators of 32 windows x 32 channels
//   V         never touches LDS: lane (window, k half) reads the 2 x 4 patch pixels its position row needs (B^T d B per axis is
//             x[ia] +- x[ib]) from the raw halo, 16 bytes = 4 channels each, and transforms them IN REGISTERS straight into the
//             MFMA A-fragment layout (the sign of the row op rides in one fma, exactly the rounding of k_conv_wino's add / subtract)
//   U         a wave's 4 positions of an 8-channel chunk = 4 KB, fetched by that wave alone (LDS-DMA ring, no barrier: only the
//             issuing wave reads it); the weights are the NT = 1 Winograd image (pack_wino) for every layer
//   raw halo  32-channel stages, chunk-major image [8-channel chunk][pixel][2 x 16 B] (tools/wino_w_layout_check.py: consistent
//             and bank-conflict-free), two stages deep; ONE barrier per 32 channels = per 64 WB MFMAs
//   tail      accumulators -> LDS (lane-linear, 64 KB per block) -> wave e takes window row e of all 16 positions, runs
//             k_conv_wino's output transform on it and the shared epilogue (conv_epilogue_b on an 8 x 16 tile, one sub-tile per wave)
// Every per-output sum is k_conv_wino's: channels in groups of 8 in the k order (0,4,1,5,2,6,3,7) of the fragment layout, V and
// Y = A^T M A by the same adds in the same order -- BIT-IDENTICAL to k_conv_wino<1> and <2> (tests/test_gpu_wino_w.py), so
// choosing it by micro-batch size is a scheduling choice.
//   vmcnt     counted, with a uniform issue pattern: per chunk 4 U pieces (UD chunks ahead), per stage RAW_IT raw pieces (one
//             stage ahead); past the end the pieces are issued with out-of-range offsets (zeros, no traffic) so that the counts
//             stay compile-time constants.
template <int WB>
__global__ __launch_bounds__(256, 1) void k_conv_wino_w(ConvArgs a) {
    static_assert(WB == 1 || WB == 2, "one or two 32-window blocks per workgroup");
    constexpr int TH = 8 * WB, RP = 18;
    constexpr int RAW_PIX = (TH + 2) * RP;               // 180 | 324
    constexpr int RAW_PIECES = RAW_PIX * 8;              // 16-byte pieces of a 32-channel stage
    constexpr int RAW_IT = (RAW_PIECES + 255) / 256;     // 6 | 11 LDS-DMA instructions per wave and stage
    constexpr int RAW_PAD = RAW_IT * 4096;
    constexpr int GSTRIDE = RAW_PIX * 32;                // bytes between the 8-channel chunks of a stage
    constexpr int WBSTRIDE = 8 * RP * 32;                // bytes between the window blocks (8 pixel rows)
    constexpr int UD = (WB == 1) ? 4 : 3;                // U look-ahead in chunks; ring of UD + 1
    constexpr int US = UD + 1;
    constexpr int UCH = 16 * 1024;                       // one chunk's weights: [16 positions][32 cout][8 ch]
    constexpr int UBASE = 2 * RAW_PAD;
    static_assert(UBASE + US * UCH <= 160 * 1024 && WB * 65536 <= UBASE + US * UCH, "LDS budget / exchange buffer");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;

    // tile decode: as k_conv_wino (grid.z = frame group x column tile x frame-in-group)
    const int bz = (int)blockIdx.z;
    const int gz = a.zdiv << a.zgroup_shift;
    const int q_ = (gz == 1) ? bz : (int)(((float)bz + 0.5f) * a.zrcp);
    const int rz = bz - q_ * gz;
    const int n_tile = rz >> a.zgroup_shift;
    const int b = (q_ << a.zgroup_shift) + (rz & ((1 << a.zgroup_shift) - 1));
    if (b >= a.frames) return;
    const int ty0 = (int)blockIdx.y * TH, tx0 = (int)blockIdx.x * 16;
    // diagnostic timeline (og_unet_clock_probe only; nullptr on every product path): entry, loop start, loop end, exit
    unsigned long long* st = nullptr;
    if (a.stamps != nullptr) {
        const unsigned wg = ((unsigned)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        st = a.stamps + 4ull * (wg < 1023u ? wg : 1023u);
        if (tid == 0) st[0] = __builtin_amdgcn_s_memtime();
    }

    const int n_st = a.n_chunks, n_ck = 4 * a.n_chunks;   // 32-channel stages, 8-channel chunks
    const og_i32x4 in_rsrc = og_make_rsrc(a.in + (long long)b * a.in_frame_stride + a.in_ch_off,
                                          (unsigned)(a.in_frame_stride - a.in_ch_off) * 4u);
    const og_i32x4 w_rsrc = og_make_rsrc(a.wpk + (long long)n_tile * n_ck * (UCH / 4), (unsigned)n_ck * (unsigned)UCH);
    const unsigned lds0 = og_lds_addr(smem);

    // raw halo pieces of this thread: piece q = it * 256 + tid lands at LDS byte 16 q of the stage's buffer and is
    // (chunk g, pixel P = hy * RP + (hx & 1) * 9 + (hx >> 1), half hp) holding channels 8 g + 4 (hp ^ (hy >> 2 & 1)) ..
    unsigned hoff[RAW_IT];
    {   // q walks in steps of 256 pieces = 128 pixels = 7 image rows + 2 pixels: one division for the first piece, then adds
        int g = 0, hy = (tid >> 1) / RP, r18 = (tid >> 1) - hy * RP;
        const int hp = tid & 1;
#pragma unroll
        for (int it = 0; it < RAW_IT; ++it) {
            const int hx = (r18 >= 9) ? 2 * (r18 - 9) + 1 : 2 * r18;
            const int lg = hp ^ ((hy >> 2) & 1);
            const int gy = ty0 + hy - 1, gx = tx0 + hx - 1;
            const bool inb = it * 256 + tid < RAW_PIECES && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
            hoff[it] = inb ? (unsigned)((gy * a.W + gx) * a.in_pix_stride * 4 + g * 32 + lg * 16) : OG_OOB;
            r18 += 128 % RP;
            hy += 128 / RP;
            if (r18 >= RP) { r18 -= RP; hy += 1; }
            if (hy >= TH + 2) { hy -= TH + 2; g += 1; }
        }
    }
    auto issue_raw = [&](int s) {   // stage s -> raw[s & 1]; past the last stage: zero records (zeros, no traffic), same count
        og_i32x4 rs = in_rsrc;
        rs.z = (s < n_st) ? in_rsrc.z : 0;
        const unsigned base = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)((s & 1) * RAW_PAD) + wave * 1024);
#pragma unroll
        for (int it = 0; it < RAW_IT; ++it) glds16b(hoff[it], rs, (unsigned)s * 128u, base + it * 4096);
    };
    auto issue_u = [&](int c) {   // positions 4 wave .. + 3 of chunk c -> U[c % US]; past the last chunk: zero records, as above
        og_i32x4 ws = w_rsrc;
        ws.z = (c < n_ck) ? w_rsrc.z : 0;
        const unsigned base = __builtin_amdgcn_readfirstlane(lds0 + UBASE + (unsigned)((c % US) * UCH) + wave * 4096);
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) glds16b((unsigned)lane * 16u, ws, (unsigned)((c * 16 + 4 * wave + cc) * 1024), base + cc * 1024);
    };
    // only what the first chunk needs is requested up front (raw(0), U(0), U(1): 14 / 19 pieces); the rest of the look-ahead --
    // U(2 .. UD), raw(1) -- is issued behind the MFMAs of chunk 0 (MODE 3 below), in the order the counted waits assume
    issue_raw(0);
    issue_u(0);
    issue_u(1);

    // position row of this wave: B^T row i = x[ra] (+ if i == 1, else -) x[rb]
    const int ra = (wave == 0) ? 0 : (wave == 2) ? 2 : 1;
    const int rb = (wave == 0) ? 2 : (wave == 1) ? 2 : (wave == 2) ? 1 : 3;
    const float rsgn = (wave == 1) ? 1.0f : -1.0f;
    // MFMA row li = window (row wr, column wc) of a block; patch pixel (r, pc) of it at P = (2 wr + r) RP + (pc & 1) 9 + wc + (pc >> 1)
    const int wr = li & 3, wc = 2 * (li >> 3) + ((li >> 2) & 1);
    unsigned rbase[2][4];
#pragma unroll
    for (int rsel = 0; rsel < 2; ++rsel) {
        const int hy = 2 * wr + (rsel ? rb : ra);
#pragma unroll
        for (int pc = 0; pc < 4; ++pc) {
            rbase[rsel][pc] = lds0 + (unsigned)((hy * RP + (pc & 1) * 9 + wc + (pc >> 1)) * 32 + ((lh ^ ((hy >> 2) & 1)) << 4));
            asm volatile("" : "+v"(rbase[rsel][pc]));
        }
    }
    unsigned ubase = lds0 + UBASE + (unsigned)(wave * 4096 + li * 32 + ((lh ^ ((li >> 3) & 1)) << 4));
    asm volatile("" : "+v"(ubase));
    const int ecol = n_tile * 32 + li;
    const float esc = a.scale[ecol], esh = a.shift[ecol];

    f32x16 acc[WB][4];
#pragma unroll
    for (int wb = 0; wb < WB; ++wb)
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[wb][cc][r] = 0.f;
            asm volatile("" : "+a"(acc[wb][cc]));   // zeroed here, under the latency of the first DMAs (as k_conv_wino)
        }

    f32x4 tv[2][WB][4], bv[2][4], rd[WB][2][4];
    auto fma4 = [](float sgn, f32x4 b_, f32x4 a_) {   // sgn * b + a as two v_pk_fma_f32 (hipcc emits four v_fma_f32 for it; same fused rounding)
        const f32x2 s2 = {sgn, sgn};
        f32x2 lo, hi;
        asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(lo) : "v"(s2), "v"(f32x2{b_.x, b_.y}), "v"(f32x2{a_.x, a_.y}));
        asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(hi) : "v"(s2), "v"(f32x2{b_.z, b_.w}), "v"(f32x2{a_.z, a_.w}));
        return f32x4{lo.x, lo.y, hi.x, hi.y};
    };
    constexpr int NR = 8 * WB;   // raw patch reads per chunk: index r = (block, row select, patch column)
    unsigned rcur[2][4];   // rbase + the ring buffer of the stage being read (8 adds per stage, not one per read)
#pragma unroll
    for (int i = 0; i < 8; ++i) rcur[i >> 2][i & 3] = rbase[i >> 2][i & 3];
    auto read_raw = [&](int r, int j) {   // chunk j of the stage: a compile-time offset of the read
        rd[r >> 3][(r >> 2) & 1][r & 3] = og_lds_read16(rcur[(r >> 2) & 1][r & 3] + (unsigned)(j * GSTRIDE + (r >> 3) * WBSTRIDE));
    };
    // quad-op n of the transform of block wb: 0-3 row op of patch column n (t = x[ra] +- x[rb]), 4-7 column op -> position 4 w + (n - 4)
    f32x4 t[WB][4];
    auto xf = [&](int wb, int n, int to) {
        if (n < 4) t[wb][n] = fma4(rsgn, rd[wb][1][n], rd[wb][0][n]);
        else if (n == 4) tv[to][wb][0] = og_sub4(t[wb][0], t[wb][2]);
        else if (n == 5) tv[to][wb][1] = og_add4(t[wb][1], t[wb][2]);
        else if (n == 6) tv[to][wb][2] = og_sub4(t[wb][2], t[wb][1]);
        else tv[to][wb][3] = og_sub4(t[wb][1], t[wb][3]);
    };

    // workgroup barrier WITHOUT a fence: __syncthreads() would make hipcc wait vmcnt(0) for the loads it tracks (scale / shift), i.e.
    // for every LDS-DMA queued behind them; hipcc's own LDS reads are waited for here, the DMA by the counted vmcnt in front
    auto barrier = [] { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // raw(0) and U(0) landed; U(1) in flight
    barrier();
#pragma unroll
    for (int r = 0; r < NR; ++r) read_raw(r, 0);
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) bv[0][cc] = og_lds_read16(ubase + (unsigned)(cc * 1024));
#pragma unroll
    for (int wb = 0; wb < WB; ++wb)
#pragma unroll
        for (int n = 0; n < 8; ++n) xf(wb, n, 0);
    if (st != nullptr && tid == 0) st[1] = __builtin_amdgcn_s_memtime();

    // One chunk: 16 WB MFMAs on the fragments tv / bv[cur].  One wave per SIMD: nothing but this wave's own instruction stream can
    // fill the matrix pipe's shadow, so the side work of the NEXT chunk (c_rd = 4 s_rd + j_rd) is cut into micro-ops with a fixed
    // slot behind the MFMAs (as k_conv_wino does): slot 0 the waits (+ the stage barrier), then 2 LDS reads per slot, the 4 U pieces
    // of chunk c_rd + UD one per slot, the transform one quad-op per slot, and on a stage boundary (MODE 2) the raw pieces of stage
    // s_rd + 1 one per slot.  MODE 0: last chunk, MFMAs only.  MODE 3: chunk 0 -- it also issues the look-ahead the prologue left
    // out, U(2 .. UD) then raw(1), in front of its own U(UD + 1), spread over its slots.
    constexpr int S_DU = NR / 2 + 2, S_X = NR / 2 + 3, S_DR = NR / 2 + 6;
    static_assert(S_X + 8 * WB <= 16 * WB && S_DR + RAW_IT <= 16 * WB, "side work of a chunk must fit its MFMA slots");
    auto chunk = [&](int cur, int mode, int s_rd, int j_rd, int c_rd) {
        const unsigned ucur = ubase + (unsigned)((c_rd % US) * UCH);
        const int cu = c_rd + UD;
        og_i32x4 ws = w_rsrc, rs = in_rsrc;
        ws.z = (cu < n_ck) ? w_rsrc.z : 0;            // past the end: zero records -> zeros, no traffic, same vmcnt count
        rs.z = (s_rd + 1 < n_st) ? in_rsrc.z : 0;
        const unsigned ub = __builtin_amdgcn_readfirstlane(lds0 + UBASE + (unsigned)((cu % US) * UCH) + wave * 4096);
        const unsigned us = (unsigned)((cu * 16 + 4 * wave) * 1024);
        const unsigned rbs = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(((s_rd + 1) & 1) * RAW_PAD) + wave * 1024);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int wb = 0; wb < WB; ++wb)
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) {
                    const int n = (e * WB + wb) * 4 + cc;   // slot 0 .. 16 WB - 1
                    const float av = tv[cur][wb][cc][e], bw = bv[cur][cc][e];
                    acc[wb][cc] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bw, acc[wb][cc], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (mode == 3 && n >= 1) {   // pieces [p0, p1) of: U(2 .. UD) | raw(1) | U(UD + 1)
                        constexpr int NP = 4 * UD + RAW_IT, NSL = 16 * WB - 1;
                        const int p0 = ((n - 1) * NP) / NSL, p1 = (n * NP) / NSL;
#pragma unroll
                        for (int pp = 0; pp < 3; ++pp) {
                            const int q = p0 + pp;
                            if (q < p1) {
                                if (q < 4 * (UD - 1) || q >= 4 * (UD - 1) + RAW_IT) {
                                    const int qu = (q < 4 * (UD - 1)) ? q : q - RAW_IT;      // piece index among the U pieces of chunks 2 .. UD + 1
                                    const int cq = 2 + qu / 4, pc_ = qu % 4;
                                    og_i32x4 wq = w_rsrc;
                                    wq.z = (cq < n_ck) ? w_rsrc.z : 0;
                                    const unsigned bq = __builtin_amdgcn_readfirstlane(lds0 + UBASE + (unsigned)((cq % US) * UCH) + wave * 4096);
                                    glds16b((unsigned)lane * 16u, wq, (unsigned)((cq * 16 + 4 * wave + pc_) * 1024), bq + pc_ * 1024);
                                } else {
                                    const int it = q - 4 * (UD - 1);
                                    og_i32x4 rq = in_rsrc;
                                    rq.z = (1 < n_st) ? in_rsrc.z : 0;
                                    const unsigned bq = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)RAW_PAD + wave * 1024);
                                    glds16b(hoff[it], rq, 128u, bq + it * 4096);
                                }
                            }
                        }
                    }
                    if (mode != 0) {
                        if (n == 0) {
                            if (mode == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // U(1): nothing younger has been issued yet
                            if (mode == 2) {
                                // the next chunk opens stage s_rd: its raw halo (issued 4 chunks = 12 younger U pieces ago) has landed
                                // for every wave behind this barrier, and every read of stage s_rd - 1 (taken a chunk ago) has
                                // returned, so that buffer takes stage s_rd + 1 below
                                asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
                                barrier();
                            }
                            // U(c_rd): younger than it are U(c_rd + 1 .. c_rd + UD - 1) and the raw stage issued in the UD chunks
                            // before this one, if any
                            if (mode == 3) {
                            } else if (UD == 4 || mode != 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (UD - 1) + RAW_IT) : "memory");
                            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (UD - 1)) : "memory");
                            if (mode == 2) {
                                const unsigned ring = (unsigned)((s_rd & 1) * RAW_PAD);
#pragma unroll
                                for (int i = 0; i < 8; ++i) rcur[i >> 2][i & 3] = rbase[i >> 2][i & 3] + ring;
                            }
                        }
                        if (n < NR / 2) {
                            read_raw(2 * n, j_rd);
                            read_raw(2 * n + 1, j_rd);
                        } else if (n < NR / 2 + 2) {
                            const int c0 = 2 * (n - NR / 2);
                            bv[cur ^ 1][c0] = og_lds_read16(ucur + (unsigned)(c0 * 1024));
                            bv[cur ^ 1][c0 + 1] = og_lds_read16(ucur + (unsigned)((c0 + 1) * 1024));
                        }
                        if (mode != 3 && n >= S_DU && n < S_DU + 4) glds16b((unsigned)lane * 16u, ws, us + (unsigned)((n - S_DU) * 1024), ub + (n - S_DU) * 1024);
                        if (n >= S_X && n < S_X + 8 * WB) xf((n - S_X) >> 3, (n - S_X) & 7, cur ^ 1);
                        if (mode == 2 && n >= S_DR && n < S_DR + RAW_IT) glds16b(hoff[n - S_DR], rs, (unsigned)(s_rd + 1) * 128u, rbs + (n - S_DR) * 4096);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
    };
    // (straight-line stages: a branch around a chunk inside the loop made hipcc keep two copies of the accumulators and move all
    //  64 WB registers there and back once per stage)
    chunk(0, 3, 0, 1, 1);                                                        // chunk 0: also the rest of the look-ahead
    chunk(1, 1, 0, 2, 2);
    chunk(0, 1, 0, 3, 3);
    if (n_st > 1) {
        chunk(1, 2, 1, 0, 4);                                                    // chunk 3 before stage 1
        for (int s = 1; s + 1 < n_st; ++s) {
#pragma unroll
            for (int j = 0; j < 3; ++j) chunk(j & 1, 1, s, j + 1, 4 * s + j + 1);   // chunks 4 s + j, j < 3: the next chunk is in the same stage
            chunk(1, 2, s + 1, 0, 4 * s + 4);                                    // chunk 4 s + 3 before another stage
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) chunk(j & 1, 1, n_st - 1, j + 1, 4 * (n_st - 1) + j + 1);
    }
    chunk(1, 0, 0, 0, 0);                                                        // the last chunk: MFMAs only
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the out-of-range tail pieces still write (zeros) into the rings
    barrier();
    if (st != nullptr && tid == 0) st[2] = __builtin_amdgcn_s_memtime();

    // ---- exchange: every wave's 4 positions -> LDS, lane-linear; wave e takes window row e (accumulator registers 4 q + e) of all
    //      16 positions = the register set k_conv_wino's output transform combines into sub-tile e of the direct accumulator layout ----
#pragma unroll
    for (int wb = 0; wb < WB; ++wb)
#pragma unroll
        for (int cc = 0; cc < 4; ++cc)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const f32x4 v = {acc[wb][cc][e], acc[wb][cc][4 + e], acc[wb][cc][8 + e], acc[wb][cc][12 + e]};
                *(OG_LDS_AS f32x4*)(unsigned long long)(lds0 + (unsigned)((((wb * 4 + e) * 16 + cc) * 64) * 16) + (unsigned)(wave * 4 * 64 * 16 + lane * 16)) = v;
            }
    __syncthreads();
    f32x16 o[WB];
#pragma unroll
    for (int wb = 0; wb < WB; ++wb) {
        f32x4 m[16];
#pragma unroll
        for (int p = 0; p < 16; ++p) m[p] = og_lds_read16(lds0 + (unsigned)(((wb * 4) * 16 + p) * 64 * 16) + (unsigned)(wave * 16 * 64 * 16 + lane * 16));
#pragma unroll
        for (int q = 0; q < 4; ++q) {   // Y = A^T M A exactly as k_conv_wino writes it (its register r = 4 q + wave)
            float tm[2][4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                tm[0][j] = m[0 + j][q] + m[4 + j][q] + m[8 + j][q];
                tm[1][j] = m[4 + j][q] - m[8 + j][q] - m[12 + j][q];
            }
#pragma unroll
            for (int y = 0; y < 2; ++y) {
                o[wb][4 * q + 2 * y + 0] = tm[y][0] + tm[y][1] + tm[y][2];
                o[wb][4 * q + 2 * y + 1] = tm[y][1] - tm[y][2] - tm[y][3];
            }
        }
    }
    __syncthreads();   // the exchange buffer is dead: the epilogue's per-wave scratch overlays it
    unsigned char* const scr = smem + wave * 5120;
#pragma unroll
    for (int wb = 0; wb < WB; ++wb) {
        if (a.act == 1) conv_epilogue_b<1, 0, 8, 1, false>(a, &o[wb], n_tile, b, ty0 + 8 * wb, tx0, wave, 0, li, lh, esc, esh, scr);
        else conv_epilogue_b<1, 0, 8, 0, false>(a, &o[wb], n_tile, b, ty0 + 8 * wb, tx0, wave, 0, li, lh, esc, esh, scr);
    }
    if (st != nullptr && tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        st[3] = __builtin_amdgcn_s_memtime();
    }
}

// POSITION-ROW-SPLIT form of k_conv_wino_w for the deep layers of a one-frame chain (maps of <= 64 x 64 pixels with hundreds of input
// channels: a 32-window tile's serial loop of Cin / 8 chunks x 16 MFMAs per wave is what bounds them, not the chip).  The same
// 8 x 16-pixel x 32-channel tiles, but a tile's four position ROWS go to four workgroups (grid.z x 4) and the four positions of a row
// to the four waves: 4 MFMAs per wave and chunk.  Each wave transforms only its own position -- 4 patch pixels, both axes' signs as
// fma multipliers (exactly the rounding of k_conv_wino's add / subtract) -- so every accumulator holds the bits k_conv_wino computes.
// The raw accumulators meet in the split-K workspace as in k_conv_wino_ps, but a tile is 64 KB instead of 256 KB and there are four
// times as many reducing workgroups: the last of a tile's four workgroups to arrive reads all 16 positions back (16 loads per lane),
// runs k_conv_wino's output transform and the shared epilogue.  BIT-IDENTICAL to k_conv_wino whatever arrives first.
//   LDS: raw halo ring 3 x 24 KB (32-channel stages, two ahead) | U ring 9 x 4 KB (a wave's position of an 8-channel chunk, 8 chunks ahead)
//   vmcnt: counted against a uniform issue pattern (per chunk 1 U piece, per stage 6 raw pieces, out-of-range past the end); the counts
//   are the smallest the pattern ever leaves behind a piece, so a wait may cover older pieces too, never fewer.
__global__ __launch_bounds__(256, 1) void k_conv_wino_wp(ConvArgs a) {
    constexpr int RP = 18, RAW_PIX = 10 * RP, RAW_PIECES = RAW_PIX * 8, RAW_IT = (RAW_PIECES + 255) / 256;   // 180, 1440, 6
    constexpr int RAW_PAD = RAW_IT * 4096, GSTRIDE = RAW_PIX * 32;
    constexpr int RS = 3, UD = 8, US = UD + 1, UCH = 4096, UBASE = RS * RAW_PAD;
    static_assert(UBASE + US * UCH <= 160 * 1024 && 4 * 5120 <= UBASE, "LDS budget / epilogue scratch");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;

    // grid.z = k_conv_wino_w's z (frame x column tile) x position row: a tile's four workgroups are neighbours in the dispatch order
    const int prow = (int)blockIdx.z & 3, bz = (int)blockIdx.z >> 2;
    const int q_ = (a.zdiv == 1) ? bz : (int)(((float)bz + 0.5f) * a.zrcp);
    const int n_tile = bz - q_ * a.zdiv;
    const int b = q_;
    if (b >= a.frames) return;   // every position row of such a tile returns: nobody waits for it
    const int ty0 = (int)blockIdx.y * 8, tx0 = (int)blockIdx.x * 16;
    const int tile_id = ((b * a.tiles_y + (int)blockIdx.y) * a.tiles_x + (int)blockIdx.x) * a.zdiv + n_tile;
    unsigned long long* st = nullptr;   // diagnostic timeline (og_unet_clock_probe only): entry, loop start, loop end, exit (top bit: reducer)
    if (a.stamps != nullptr) {
        const unsigned wg = ((unsigned)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        st = a.stamps + 4ull * (wg < 1023u ? wg : 1023u);
        if (tid == 0) st[0] = __builtin_amdgcn_s_memtime();
    }

    const int n_st = a.n_chunks, n_ck = 4 * a.n_chunks;
    const og_i32x4 in_rsrc = og_make_rsrc(a.in + (long long)b * a.in_frame_stride + a.in_ch_off,
                                          (unsigned)(a.in_frame_stride - a.in_ch_off) * 4u);
    const og_i32x4 w_rsrc = og_make_rsrc(a.wpk + (long long)n_tile * n_ck * 4096, (unsigned)n_ck * 16384u);
    const unsigned lds0 = og_lds_addr(smem);
    unsigned hoff[RAW_IT];   // as k_conv_wino_w<1>
    {
        int g = 0, hy = (tid >> 1) / RP, r18 = (tid >> 1) - hy * RP;
        const int hp = tid & 1;
#pragma unroll
        for (int it = 0; it < RAW_IT; ++it) {
            const int hx = (r18 >= 9) ? 2 * (r18 - 9) + 1 : 2 * r18;
            const int lg = hp ^ ((hy >> 2) & 1);
            const int gy = ty0 + hy - 1, gx = tx0 + hx - 1;
            const bool inb = it * 256 + tid < RAW_PIECES && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
            hoff[it] = inb ? (unsigned)((gy * a.W + gx) * a.in_pix_stride * 4 + g * 32 + lg * 16) : OG_OOB;
            r18 += 128 % RP;
            hy += 128 / RP;
            if (r18 >= RP) { r18 -= RP; hy += 1; }
            if (hy >= 10) { hy -= 10; g += 1; }
        }
    }
    auto raw_piece = [&](int s, int it) {   // stage s -> raw[s % RS]; past the last stage: zero records (zeros, no traffic), same count
        og_i32x4 rs = in_rsrc;
        rs.z = (s < n_st) ? in_rsrc.z : 0;
        const unsigned base = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)((s % RS) * RAW_PAD) + wave * 1024);
        glds16b(hoff[it], rs, (unsigned)s * 128u, base + it * 4096);
    };
    auto u_piece = [&](int c) {   // position 4 prow + wave of chunk c -> U[c % US]
        og_i32x4 ws = w_rsrc;
        ws.z = (c < n_ck) ? w_rsrc.z : 0;
        const unsigned base = __builtin_amdgcn_readfirstlane(lds0 + UBASE + (unsigned)((c % US) * UCH) + wave * 1024);
        glds16b((unsigned)lane * 16u, ws, (unsigned)((c * 16 + 4 * prow + wave) * 1024), base);
    };
#pragma unroll
    for (int it = 0; it < RAW_IT; ++it) raw_piece(0, it);
#pragma unroll
    for (int c = 0; c <= UD; ++c) u_piece(c);
#pragma unroll
    for (int it = 0; it < RAW_IT; ++it) raw_piece(1, it);
#pragma unroll
    for (int it = 0; it < RAW_IT; ++it) raw_piece(2, it);

    // position (prow, wave): B^T d B per axis is x[ia] +- x[ib] with (ia, ib, sign) = (0,2,-) (1,2,+) (2,1,-) (1,3,-)
    auto ia_of = [](int i) { return (i == 0) ? 0 : (i == 2) ? 2 : 1; };
    auto ib_of = [](int i) { return (i == 0) ? 2 : (i == 1) ? 2 : (i == 2) ? 1 : 3; };
    const int ra = ia_of(prow), rb = ib_of(prow), ca = ia_of(wave), cb = ib_of(wave);
    const float rsgn = (prow == 1) ? 1.0f : -1.0f, csgn = (wave == 1) ? 1.0f : -1.0f;
    const int wr = li & 3, wc = 2 * (li >> 3) + ((li >> 2) & 1);
    unsigned rbase[2][2], rcur[2][2];
#pragma unroll
    for (int rsel = 0; rsel < 2; ++rsel)
#pragma unroll
        for (int csel = 0; csel < 2; ++csel) {
            const int hy = 2 * wr + (rsel ? rb : ra), pc = csel ? cb : ca;
            rbase[rsel][csel] = lds0 + (unsigned)((hy * RP + (pc & 1) * 9 + wc + (pc >> 1)) * 32 + ((lh ^ ((hy >> 2) & 1)) << 4));
            rcur[rsel][csel] = rbase[rsel][csel];
        }
    const unsigned ubase = lds0 + UBASE + (unsigned)(wave * 1024 + li * 32 + ((lh ^ ((li >> 3) & 1)) << 4));

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    asm volatile("" : "+a"(acc));
    f32x4 tv[2], bv[2], rd[2][2];
    auto fma4 = [](float sgn, f32x4 b_, f32x4 a_) {   // sgn * b + a as two v_pk_fma_f32 (hipcc emits four v_fma_f32 for it; same fused rounding)
        const f32x2 s2 = {sgn, sgn};
        f32x2 lo, hi;
        asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(lo) : "v"(s2), "v"(f32x2{b_.x, b_.y}), "v"(f32x2{a_.x, a_.y}));
        asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(hi) : "v"(s2), "v"(f32x2{b_.z, b_.w}), "v"(f32x2{a_.z, a_.w}));
        return f32x4{lo.x, lo.y, hi.x, hi.y};
    };
    auto read_raw = [&](int r, int j) { rd[r >> 1][r & 1] = og_lds_read16(rcur[r >> 1][r & 1] + (unsigned)(j * GSTRIDE)); };
    auto xform = [&](int to) {   // rows first (t = x[ra] +- x[rb] for the two columns), then the column op: k_conv_wino's order
        const f32x4 ta = fma4(rsgn, rd[1][0], rd[0][0]), tb = fma4(rsgn, rd[1][1], rd[0][1]);
        tv[to] = fma4(csgn, tb, ta);
    };
    auto barrier = [] { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(UD + 2 * RAW_IT) : "memory");   // raw(0) and U(0) landed
    barrier();
#pragma unroll
    for (int r = 0; r < 4; ++r) read_raw(r, 0);
    bv[0] = og_lds_read16(ubase);
    xform(0);
    if (st != nullptr && tid == 0) st[1] = __builtin_amdgcn_s_memtime();

    // one chunk: 4 MFMAs on tv / bv[cur]; behind them the side work of the NEXT chunk (c_rd = 4 s_rd + j_rd), in fixed slots.
    // MODE 1: same stage; MODE 2: the next chunk opens stage s_rd (barrier, stage s_rd + 2 requested); MODE 0: last chunk
    auto chunk = [&](int cur, int mode, int s_rd, int j_rd, int c_rd) {
        const unsigned ucur = ubase + (unsigned)((c_rd % US) * UCH);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(tv[cur][e], bv[cur][e], acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (mode != 0) {
                if (e == 0) {
                    if (mode == 2) {
                        // raw(s_rd): at least the 3 U pieces of the last three chunks and the next stage's 6 raw pieces are younger
                        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 + RAW_IT) : "memory");
                        barrier();
                        const unsigned ring = (unsigned)((s_rd % RS) * RAW_PAD);
#pragma unroll
                        for (int i = 0; i < 4; ++i) rcur[i >> 1][i & 1] = rbase[i >> 1][i & 1] + ring;
                    }
                    // U(c_rd): at least U(c_rd + 1 .. c_rd + UD - 1) and one raw stage are younger
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(UD - 1 + RAW_IT) : "memory");
                    read_raw(0, j_rd);
                    read_raw(1, j_rd);
                } else if (e == 1) {
                    read_raw(2, j_rd);
                    read_raw(3, j_rd);
                    bv[cur ^ 1] = og_lds_read16(ucur);
                } else if (e == 2) {
                    u_piece(c_rd + UD);
                    if (mode == 2) { raw_piece(s_rd + 2, 0); raw_piece(s_rd + 2, 1); raw_piece(s_rd + 2, 2); }
                } else {
                    xform(cur ^ 1);
                    if (mode == 2) { raw_piece(s_rd + 2, 3); raw_piece(s_rd + 2, 4); raw_piece(s_rd + 2, 5); }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    for (int s = 0; s + 1 < n_st; ++s) {
#pragma unroll
        for (int j = 0; j < 3; ++j) chunk(j & 1, 1, s, j + 1, 4 * s + j + 1);
        chunk(1, 2, s + 1, 0, 4 * s + 4);
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) chunk(j & 1, 1, n_st - 1, j + 1, 4 * (n_st - 1) + j + 1);
    chunk(1, 0, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the out-of-range tail pieces still write (zeros) into the rings
    barrier();
    if (st != nullptr && tid == 0) st[2] = __builtin_amdgcn_s_memtime();

    // ---- raw accumulators out: [window row e][position][lane] x 16 B (registers e, 4 + e, 8 + e, 12 + e), device scope; arrival;
    //      the last of the tile's four workgroups: k_conv_wino's tail on window row `wave` of all 16 positions ----
    float* const part = a.partial + (long long)tile_id * 16384 + lane * 4;
#pragma unroll
    for (int e = 0; e < 4; ++e)
        og_store16_dev(part + ((e * 16 + 4 * prow + wave) * 64) * 4, f32x4{acc[e], acc[4 + e], acc[8 + e], acc[12 + e]});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's part is written through
    __syncthreads();
    int* const flag = (int*)smem;
    if (tid == 0) *flag = (atomicAdd(a.tile_counter + tile_id, 1) == 3) ? 1 : 0;
    __syncthreads();
    if (*(volatile int*)flag == 0) {
        if (st != nullptr && tid == 0) st[3] = __builtin_amdgcn_s_memtime();
        return;
    }
    __syncthreads();   // the flag word is part of wave 0's epilogue scratch
    const int ecol = n_tile * 32 + li;
    const float esc = a.scale[ecol], esh = a.shift[ecol];
    const __amdgpu_buffer_rsrc_t part_rs = og_rsrc(a.partial + (long long)tile_id * 16384, 65536u);
    const unsigned part_voff = (unsigned)((wave * 16 * 64 + lane) * 16);
    f32x16 o;
    {
        f32x4 m[16];
#pragma unroll
        for (int p = 0; p < 16; ++p) m[p] = og_load16_dev(part_rs, part_voff, (unsigned)(p * 1024));
#pragma unroll
        for (int q = 0; q < 4; ++q) {   // Y = A^T M A exactly as k_conv_wino writes it (its register r = 4 q + wave)
            float tm[2][4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                tm[0][j] = m[0 + j][q] + m[4 + j][q] + m[8 + j][q];
                tm[1][j] = m[4 + j][q] - m[8 + j][q] - m[12 + j][q];
            }
#pragma unroll
            for (int y = 0; y < 2; ++y) {
                o[4 * q + 2 * y + 0] = tm[y][0] + tm[y][1] + tm[y][2];
                o[4 * q + 2 * y + 1] = tm[y][1] - tm[y][2] - tm[y][3];
            }
        }
    }
    unsigned char* const scr = smem + wave * 5120;
    if (a.act == 1) conv_epilogue_b<1, 0, 8, 1, false>(a, &o, n_tile, b, ty0, tx0, wave, 0, li, lh, esc, esh, scr);
    else conv_epilogue_b<1, 0, 8, 0, false>(a, &o, n_tile, b, ty0, tx0, wave, 0, li, lh, esc, esh, scr);
    if (tid == 0) a.tile_counter[tile_id] = 0;   // ready for the next launch on this stream
    if (st != nullptr && tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        st[3] = __builtin_amdgcn_s_memtime() | (1ull << 63);
    }
}

// ConvTranspose2d(2f, f, 2, 2) (unet.py:69,82) for launches that cannot fill the chip (one frame per kernel chain: the 16 x 16 map of
// the deepest level is 2 tiles x 16 column tiles of k_conv_mfma_o / _p, each with a serial loop over 512 input channels).  Same GEMM
// (N = 4 Cout columns (dy, dx, co)), same k order per output (channel chunks of 32, k groups of 8, the fragment layout's pairs) and
// the same epilogue as the direct kernels -- bit-identical -- on the finest tiles the MFMA allows: a WAVE owns 32 input pixels
// (2 rows x 16) x 32 columns, a workgroup four column tiles of the same pixels (the A lines are then shared in the CU's L1).
// Operands go straight from memory to the fragment registers (16 B per lane, three chunks ahead): at this size an LDS round trip
// and its barriers cost more than the 32-byte gathers.  Weights: the NT = 1 image of pack_gemm_b (row = column n, slot' = slot ^ (n >> 1 & 7)).
__global__ __launch_bounds__(256) void k_convt_w(ConvArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[4 * 5120];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int b = (int)blockIdx.z / a.zdiv;                       // zdiv = column tiles / 4
    const int n_tile = ((int)blockIdx.z - b * a.zdiv) * 4 + wave;
    const int ty0 = (int)blockIdx.y * 2, tx0 = (int)blockIdx.x * 16;
    // A rows: i -> 2x2-window-major pixel order, as every kernel that feeds conv_epilogue_b
    const int px = tx0 + 2 * (li >> 2) + (li & 1), py = ty0 + ((li >> 1) & 1);
    const __amdgpu_buffer_rsrc_t in_rs = og_rsrc(a.in + (long long)b * a.in_frame_stride + a.in_ch_off, (unsigned)(a.in_frame_stride - a.in_ch_off) * 4u);
    const unsigned aoff = (py < a.H && px < a.W) ? (unsigned)(((py * a.W + px) * a.in_pix_stride + 4 * lh) * 4) : OG_OOB;
    const __amdgpu_buffer_rsrc_t w_rs = og_rsrc(a.wpk + (long long)n_tile * a.n_chunks * 1024, (unsigned)a.n_chunks * 4096u);
    unsigned boff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) boff[j] = (unsigned)(li * 128 + (((2 * j + lh) ^ ((li >> 1) & 7)) << 4));
    const int ecol = n_tile * 32 + li;
    const float esc = a.scale[ecol % a.aff_mod], esh = a.shift[ecol % a.aff_mod];

    f32x4 fa[4][4], fb[4][4];   // [ring slot][k group]
    auto fetch = [&](int c, int slot) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            fa[slot][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rs, aoff, (unsigned)((c * 32 + 8 * j) * 4), 0));
            fb[slot][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rs, boff[j], (unsigned)(c * 4096), 0));
        }
    };
    const int n = a.n_chunks;
#pragma unroll
    for (int c = 0; c < 3; ++c)
        if (c < n) fetch(c, c);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int c0 = 0; c0 < n; c0 += 4) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = c0 + k;
            if (c < n) {
                if (c + 3 < n) fetch(c + 3, (k + 3) & 3);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4 av = fa[k][j], bv = fb[k][j];
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc, 0, 0, 0);
                }
            }
        }
    }
    conv_epilogue_b<1, 1, 8, 0, false, false, 1>(a, &acc, n_tile, b, ty0, tx0, 0, 0, li, lh, esc, esh, smem + wave * 5120);
}

// Split-precision twin of k_conv_mfma_o (MODE 0: 3x3 conv, MODE 1: 2x2 stride-2 transposed conv): same tiles, halo /
// weight staging, LDS images, swizzles and fragment addressing; the operands are f16 hi/lo pairs in the H layout and each
// (chunk, tap) costs 6 x v_mfma_f32_32x32x16_f16 per 32-row sub-tile instead of 16 x v_mfma_f32_32x32x2_f32.  No split-K.
// Opt-in ("precision" 1); the f32 kernel stays the default and the parity reference.
// SQ (NT == 2, TH == 16 only): every wave computes 2 row sub-tiles x BOTH 32-column sub-tiles instead of 4 x 1 -- 16 instead
// of 20 fragment reads per 24 MFMAs; LDS reads are what this kernel's MFMA rate is paid with (tools/ubench/mfma_f16_split).
template <int NT, int MODE, int TH, int OCC, bool FIRST = false, bool SQ = false>
__global__ __launch_bounds__(256, OCC) void k_conv_mfma_h(ConvArgs a) {
    static_assert(MODE == 0 || MODE == 1, "split precision: 3x3 conv and transposed conv only");
    static_assert(!SQ || (NT == 2 && TH == 16 && !FIRST), "square wave tiles: 64-column kernel on 16x16 tiles");
    constexpr int TW = 16;
    constexpr int PAD = (MODE == 0) ? 1 : 0;
    constexpr int HW_ = TW + 2 * PAD;
    constexpr int HH_ = TH + 2 * PAD;
    constexpr int HALO_PIX = HW_ * HH_;
    constexpr int HALO_BYTES = HALO_PIX * 128;
    constexpr int HALO_PIECES = HALO_PIX * 8;
    constexpr int HALO_IT = (HALO_PIECES + 255) / 256;
    constexpr int TAPS = (MODE == 0) ? 9 : 1;
    constexpr int WROWS = 32 * NT;
    constexpr int WBYTES = WROWS * 128;
    constexpr int NSTG = (MODE == 0) ? 3 : 2;   // weight ring: a tap's stage is t % 3 (compile time) for the 3x3 conv
    // Weight slices are staged TWO steps ahead in the 3-stage ring: a step is ~5x shorter than in the f32 kernel (24 MFMAs
    // of 32 cycles per wave at most), less than the latency of the LDS-DMA that has to land before the next step.
    constexpr int WAHEAD = (NSTG == 3) ? 2 : 1;
    constexpr int WM = SQ ? 4 : 4 / NT;          // waves along M
    constexpr int NC = SQ ? 2 : 1;               // 32-column sub-tiles per wave
    constexpr int MS = (TH / 2) / WM;            // 32-row M sub-tiles (2 pixel rows x 16) per wave
    static_assert(MS >= 1, "tile too small");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const halo0 = smem;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = SQ ? 0 : wave % NT;
    const int wm = SQ ? wave : wave / NT;
    const int li = lane & 31;
    const int lh = lane >> 5;
    if (a.prio_mode == 3) __builtin_amdgcn_s_setprio(3);   // set-up, DMA issue and epilogue at raised priority (see k_conv_mfma_o)

    // ---- tile decode (scalar): grid = (tile column, tile row, frame group x column tile x frame in group), as k_conv_mfma_o ----
    const int bz = (int)blockIdx.z;
    const int gz = a.zdiv << a.zgroup_shift;
    const int q = (gz == 1) ? bz : (int)(((float)bz + 0.5f) * a.zrcp);   // zrcp = 1 / gz; exact for bz < 2^16
    const int rz = bz - q * gz;
    const int zr = rz >> a.zgroup_shift;                                   // column tile x K parts + K part
    const int b = (q << a.zgroup_shift) + (rz & ((1 << a.zgroup_shift) - 1));
    if (b >= a.frames) return;   // tail of the last frame group (whole workgroup, before any barrier)
    // split-K (small launches, one frame per chain): K part = a range [s_lo, s_hi) of (chunk, tap) steps; the last part of a
    // tile to arrive sums all parts in split order and runs the epilogue (as k_conv_mfma_o; not with the square wave tiles)
    const int ks_n = SQ ? 1 : a.ksplit;
    const int n_tile = (ks_n == 1) ? zr : zr / ks_n;
    const int kpart = (ks_n == 1) ? 0 : zr - n_tile * ks_n;
    const int ty0 = (int)blockIdx.y * TH;
    const int tx0 = (int)blockIdx.x * TW;
    const int n_steps = a.n_chunks * TAPS;
    const int s_lo = (ks_n == 1) ? 0 : (kpart * n_steps) / ks_n;
    const int s_hi = (ks_n == 1) ? n_steps : ((kpart + 1) * n_steps) / ks_n;
    const int c_lo = s_lo / TAPS, c_hi = (s_hi + TAPS - 1) / TAPS;
    const int tile_id = n_tile * a.n_spatial + (b * a.tiles_y + (int)blockIdx.y) * a.tiles_x + (int)blockIdx.x;

    // this frame's input as a raw buffer: offsets past num_records read as zeros (= the conv's zero padding)
    const og_i32x4 in_rsrc = og_make_rsrc(a.in + (long long)b * a.in_frame_stride + a.in_ch_off,
                                          (unsigned)(a.in_frame_stride - a.in_ch_off) * 4u);

    // ---- per-thread halo source offsets (bytes, fixed across channel chunks); pixel index walks incrementally ----
    unsigned hoff[HALO_IT];
    {
        int hy = ((tid >> 3) >= HW_) ? 1 : 0;
        int hx = (tid >> 3) - hy * HW_;
        const int row_b = a.W * a.in_pix_stride * 4;   // bytes per halo row step
        const int col_b = a.in_pix_stride * 4;         // bytes per halo column step
#pragma unroll
        for (int it = 0; it < HALO_IT; ++it) {
            const int logical = (tid & 7) ^ og_halo_swz(hy, hx);
            const int gy = ty0 + hy - PAD, gx = tx0 + hx - PAD;
            const bool inb = (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
            hoff[it] = inb ? (unsigned)(gy * row_b + gx * col_b + logical * 16) : OG_OOB;
            hx += 32 % HW_;
            hy += 32 / HW_;
            if (hx >= HW_) { hx -= HW_; hy += 1; }
        }
    }
    const bool last_valid = ((HALO_IT - 1) * 256 + tid) < HALO_PIECES;

    const unsigned lds0 = og_lds_addr(smem);
    auto stage_halo = [&](int c) {
        const unsigned base = lds0 + wave * 1024;
#pragma unroll
        for (int it = 0; it < HALO_IT; ++it) {
            if (it < HALO_IT - 1 || last_valid) glds16b(hoff[it], in_rsrc, (unsigned)c * 128u, base + it * 4096);
        }
    };
    const og_i32x4 w_rsrc = og_make_rsrc(a.wpk + (long long)n_tile * n_steps * (WROWS * 32), (unsigned)n_steps * WBYTES);
    const unsigned woff = (unsigned)tid * 16u;
    auto stage_w = [&](int stage, int step) {
        const unsigned base = lds0 + HALO_BYTES + stage * WBYTES + wave * 1024;
#pragma unroll
        for (int i = 0; i < NT; ++i) glds16b(woff, w_rsrc, (unsigned)step * WBYTES + i * 4096, base + i * 4096);
    };

    // first halo and weights are on their way before the rest of the set-up (which then hides their latency)
    if (!FIRST) {
        stage_halo(c_lo);
        stage_w((NSTG == 3) ? s_lo % 3 : (s_lo & 1), s_lo);
        if (WAHEAD == 2 && s_lo + 1 < s_hi) stage_w((s_lo + 1) % 3, s_lo + 1);
    }

    // ---- fragment addressing: exactly k_conv_mfma_o's (2x2-window-major pixel order, 12 A registers, 4 B registers) ----
    const int px0 = 2 * (li >> 2) + (li & 1);
    const int pyl = (li >> 1) & 1;
    const int brow = wn * 32 + li;
    const int boff = brow * 128 + ((lh ^ ((brow >> 1) & 7)) << 4);
    constexpr int NDX = (MODE == 0) ? 3 : 1;
    unsigned abase[NDX][4];
#pragma unroll
    for (int dx = 0; dx < NDX; ++dx) {
        const int px = px0 + dx;
        const unsigned o = (unsigned)((pyl * HW_ + px) * 128 + ((lh ^ og_halo_swz(pyl, px)) << 4) + wm * (MS * 2 * HW_ * 128));
#pragma unroll
        for (int pat = 0; pat < 4; ++pat) {
            abase[dx][pat] = lds0 + (o ^ (unsigned)(pat << 5));   // absolute LDS address
            asm volatile("" : "+v"(abase[dx][pat]));              // opaque: or hipcc re-adds the (link-time 0) smem base per read
        }
    }
    unsigned bbase[4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        bbase[jj] = lds0 + HALO_BYTES + (unsigned)(boff ^ (jj << 5));
        asm volatile("" : "+v"(bbase[jj]));
    }

    // this lane's output channel: folded BN scale / shift, loaded now so that the epilogue does not wait for them
    float esc[NC], esh[NC];
#pragma unroll
    for (int n = 0; n < NC; ++n) {
        const int ecol = n_tile * WROWS + (wn + n) * 32 + li;
        const int eco = (MODE == 1) ? ecol % a.aff_mod : ecol;
        esc[n] = a.scale[eco];
        esh[n] = a.shift[eco];
    }

    f32x16 acc[NC][MS], cor[NC][MS];   // hi*hi | hi*lo + lo*hi (scaled by 2^11)
#pragma unroll
    for (int n = 0; n < NC; ++n)
#pragma unroll
        for (int m = 0; m < MS; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc[n][m][r] = 0.f;
                cor[n][m][r] = 0.f;
            }

    if (FIRST) {
        // halo tile of the first layer's OUTPUT, computed here: 12x20 u8 patch -> /255 -> 3x3 conv -> BN -> ReLU, split and
        // written in the swizzled H-layout image the LDS-DMA would have produced
        float* patch = (float*)(smem + HALO_BYTES + NSTG * WBYTES);  // [HH_+2][HW_+2]
        float* fw = patch + (HH_ + 2) * (HW_ + 2);                // w9[9][32] | scale[32] | shift[32]
        const uint8_t* fin = a.first_u8 + (long long)b * a.H * a.W;
        for (int i = tid; i < (HH_ + 2) * (HW_ + 2); i += 256) {
            const int hy = i / (HW_ + 2), hx = i - hy * (HW_ + 2);
            const int gy = ty0 + hy - 2, gx = tx0 + hx - 2;
            patch[i] = (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) ? (float)fin[(long long)gy * a.W + gx] / 255.0f : 0.f;
        }
        for (int i = tid; i < 352; i += 256) fw[i] = (i < 288) ? a.first_w9[i] : (i < 320 ? a.first_scale[i - 288] : a.first_shift[i - 320]);
        stage_w(0, 0);
        if (WAHEAD == 2 && 1 < n_steps) stage_w(1, 1);
        __syncthreads();
        // one item = (halo pixel, 8-channel group L): the fma chain of k_conv_first for 8 channels, then hi -> slot L, lo -> slot 4+L
        float first_absmax = 0.f;
        for (int q2 = tid; q2 < HALO_PIX * 4; q2 += 256) {
            const int p = q2 >> 2, L = q2 & 3;
            const int hy = p / HW_, hx = p - hy * HW_;
            const int gy = ty0 + hy - 1, gx = tx0 + hx - 1;
            _Float16 hi[8], lo[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { hi[e] = (_Float16)0.f; lo[e] = (_Float16)0.f; }   // outside the image: the second conv's zero padding
            if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) {
                float sacc[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) sacc[e] = 0.f;
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const float xv = patch[(hy + t / 3) * (HW_ + 2) + hx + t % 3];
                    const f32x4 w0 = *(const f32x4*)(fw + t * 32 + 8 * L), w1 = *(const f32x4*)(fw + t * 32 + 8 * L + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        sacc[e] = fmaf(xv, w0[e], sacc[e]);
                        sacc[4 + e] = fmaf(xv, w1[e], sacc[4 + e]);
                    }
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float v = fmaxf(fmaf(sacc[e], fw[288 + 8 * L + e], fw[320 + 8 * L + e]), 0.f);
                    og_split(v, hi[e], lo[e]);
                    first_absmax = fmaxf(first_absmax, v);
                }
            }
            const int swz = og_halo_swz(hy, hx);
            *(f32x4*)(halo0 + p * 128 + ((L ^ swz) << 4)) = og_pack8(hi);
            *(f32x4*)(halo0 + p * 128 + (((L + 4) ^ swz) << 4)) = og_pack8(lo);
        }
        og_flag_range(first_absmax, a.range_flag);
    }
    og_wait_dma();
    __syncthreads();
    if (a.prio_mode == 3) __builtin_amdgcn_s_setprio(0);

    int step = s_lo;   // absolute (chunk, tap) index: also the weight block's index (stage = step % 3 = t % 3: 9 taps per chunk)
    for (int c = c_lo; c < c_hi; ++c) {
#pragma unroll
        for (int t = 0; t < TAPS; ++t) {
            if (ks_n > 1 && (c * TAPS + t < s_lo || c * TAPS + t >= s_hi)) continue;  // another K part's step
            const int stg = (NSTG == 3) ? t % 3 : (step & 1), stg_next = (NSTG == 3) ? (t + WAHEAD) % 3 : ((step + 1) & 1);
            const bool more = (step + WAHEAD < s_hi);
            if (more) stage_w(stg_next, step + WAHEAD);

            const unsigned wb = (unsigned)stg * WBYTES;
            const int dy = (MODE == 0) ? t / 3 : 0;
            const int dx = (MODE == 0) ? t % 3 : 0;
            // k-group j -> 16-byte slot 2j + lh of the 128-byte row: j = 0,1 the hi halves of k-steps 0,1 (channels 16j + 8lh ..),
            // j = 2,3 their lo halves -- the same four reads per fragment as the f32 kernel, other contents
            f32x4 bv[NC][4];   // column sub-tile n: rows 32n.. of the weight slice = +4096 bytes (same swizzle: (32n + li) >> 1 & 7 = li >> 1 & 7)
#pragma unroll
            for (int n = 0; n < NC; ++n)
#pragma unroll
                for (int j = 0; j < 4; ++j) bv[n][j] = og_lds_read16(bbase[j] + wb + (unsigned)(n * 4096));
#pragma unroll
            for (int m = 0; m < MS; ++m) {
                f32x4 av[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) av[j] = og_lds_read16(abase[dx][j ^ ((dy & 1) << 1)] + (unsigned)((dy + 2 * m) * (HW_ * 128)));
#pragma unroll
                for (int n = 0; n < NC; ++n)
#pragma unroll
                    for (int t2 = 0; t2 < 2; ++t2) {
                        acc[n][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(og_h8, av[t2]), __builtin_bit_cast(og_h8, bv[n][t2]), acc[n][m], 0, 0, 0);
                        cor[n][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(og_h8, av[t2]), __builtin_bit_cast(og_h8, bv[n][2 + t2]), cor[n][m], 0, 0, 0);
                        cor[n][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(og_h8, av[2 + t2]), __builtin_bit_cast(og_h8, bv[n][t2]), cor[n][m], 0, 0, 0);
                    }
            }
            // the NEXT step's slice must have landed; the one staged just now (NT instructions per wave) may stay in flight
            if (WAHEAD == 2 && more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NT) : "memory");
            else og_wait_dma();
            __syncthreads();
            ++step;
        }
        if (c + 1 < c_hi) {  // every read of the halo buffer completed before the barrier above
            stage_halo(c + 1);
            og_wait_dma();
            __syncthreads();
        }
    }

    if (a.prio_mode == 3) __builtin_amdgcn_s_setprio(3);
    // ---- epilogue (all staging buffers are dead behind the last barrier: LDS is scratch now) ----
    unsigned char* const scr = smem + wave * 5120;
#pragma unroll
    for (int n = 0; n < NC; ++n)
#pragma unroll
        for (int m = 0; m < MS; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[n][m][r] = fmaf(cor[n][m][r], OG_LO_INV, acc[n][m][r]);
    if (!SQ && ks_n > 1) {
        // this part's (combined, f32) accumulators: device-scope write-through stores, then the arrival counter
        float* pw = a.partial + (((long long)tile_id * ks_n + kpart) * 4 + wave) * (MS * 16 * 64) + lane;
#pragma unroll
        for (int m = 0; m < MS; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) __hip_atomic_store(pw + (m * 16 + r) * 64, acc[0][m][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's part is written through
        __syncthreads();
        int* const flag = (int*)smem;
        if (tid == 0) *flag = (atomicAdd(a.tile_counter + tile_id, 1) == ks_n - 1) ? 1 : 0;
        __syncthreads();
        if (!*(volatile int*)flag) return;   // not the last part of this tile
#pragma unroll
        for (int m = 0; m < MS; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[0][m][r] = 0.f;
        for (int sp_ = 0; sp_ < ks_n; ++sp_) {   // every part from memory, in split order: the sum does not depend on who is last
            const float* pr = a.partial + (((long long)tile_id * ks_n + sp_) * 4 + wave) * (MS * 16 * 64) + lane;
#pragma unroll
            for (int m = 0; m < MS; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[0][m][r] += __hip_atomic_load(pr + (m * 16 + r) * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();   // the flag word is part of wave 0's scratch
        if (tid == 0) a.tile_counter[tile_id] = 0;   // ready for the next launch on this stream
    }
#pragma unroll
    for (int n = 0; n < NC; ++n) {
        // activations leave in the H layout; the launch with the fused head stores no activation and keeps the f32 scratch
        if (a.act == 1) conv_epilogue_b<NT, MODE, TH, 1, false, true, MS>(a, acc[n], n_tile, b, ty0, tx0, wm, wn + n, li, lh, esc[n], esh[n], scr);
        else conv_epilogue_b<NT, MODE, TH, 0, false, true, MS>(a, acc[n], n_tile, b, ty0, tx0, wm, wn + n, li, lh, esc[n], esh[n], scr);
    }
}

// ---------------------------------------------------------------------------------------
// Generation 2 of the implicit-GEMM conv: PERSISTENT workgroups walking a flat sequence of
// (item = (n_tile, frame, spatial tile), 32-channel chunk, tap-group) steps.
//   * the next step's weight slices and the next unit's halo (possibly of the NEXT tile) are
//     always in flight behind the current step's MFMAs -> no per-tile prologue bubble, and the
//     epilogue stores of tile i overlap the loads of tile i+1;
//   * A/B fragments are double-buffered in registers, and the one barrier per step sits in
//     front of the step's LAST k-substep: the first fragments of the next step are read right
//     behind the barrier and their LDS latency hides under that substep's MFMAs;
//   * TPS taps per step (weights of TPS taps staged together) trades LDS for barrier count.
// Same arithmetic, same accumulation order per output element as k_conv_mfma (results are
// bit-identical between the two).  MODE 2 (this kernel only): 1x1 conv, no halo, plain epilogue.
template <int NT, int MODE, int TH, int TPS>
__global__ __launch_bounds__(256, (TH >= 16) ? 1 : 2) void k_conv_mfma_p(ConvArgs a, int n_items) {
    constexpr int TW = 16;
    constexpr int PAD = (MODE == 0) ? 1 : 0;
    constexpr int HW_ = TW + 2 * PAD;
    constexpr int HH_ = TH + 2 * PAD;
    constexpr int HALO_PIX = HW_ * HH_;
    constexpr int HALO_BYTES = HALO_PIX * 128;
    constexpr int HALO_PIECES = HALO_PIX * 8;
    constexpr int HALO_IT = (HALO_PIECES + 255) / 256;
    constexpr int TAPS = (MODE == 0) ? 9 : 1;  // MODE 1 (convT) and MODE 2 (1x1): one tap
    static_assert(TAPS % TPS == 0, "TPS must divide the tap count");
    constexpr int NSTEP_U = TAPS / TPS;  // steps per (item, chunk) unit
    constexpr int NSUB = TPS * 4;        // k8 sub-steps per step
    constexpr int WROWS = 32 * NT;
    constexpr int WBYTES = WROWS * 128;  // one tap slice
    constexpr int SBYTES = TPS * WBYTES; // one stage
    constexpr int WM = 4 / NT;
    constexpr int MS = (TH / 2) / WM;
    static_assert(MS >= 1, "tile too small");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const halo0 = smem;
    unsigned char* const wbuf0 = smem + 2 * HALO_BYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave % NT;
    const int wm = wave / NT;
    const int li = lane & 31;
    const int lh = lane >> 5;
    const int G = gridDim.x;
    const int tiles_per_frame = a.tiles_x * a.tiles_y;

    // item -> (K-split, n_tile, frame, tile); chunk range of a split: [c_lo, c_hi)
    const int ks_n = a.ksplit;
    auto c_lo = [&](int item) { return (ks_n == 1) ? 0 : ((item % ks_n) * a.n_chunks) / ks_n; };
    auto c_hi = [&](int item) { return (ks_n == 1) ? a.n_chunks : ((item % ks_n + 1) * a.n_chunks) / ks_n; };
    auto decode = [&](int item, int& n_tile, int& b, int& ty0, int& tx0) {
        const int base = (ks_n == 1) ? item : item / ks_n;
        n_tile = base / a.n_spatial;
        int sp = base - n_tile * a.n_spatial;
        b = sp / tiles_per_frame;
        sp -= b * tiles_per_frame;
        const int tyi = sp / a.tiles_x;
        ty0 = tyi * TH;
        tx0 = (sp - tyi * a.tiles_x) * TW;
    };

    // ---- halo prefetch cursor: (h_item, h_c), with per-thread source pointers of h_item ----
    const float* hsrc[HALO_IT];
    int hstep[HALO_IT];
    const bool last_valid = ((HALO_IT - 1) * 256 + tid) < HALO_PIECES;
    auto set_hsrc = [&](int item) {
        int n_tile, b, ty0, tx0;
        decode(item, n_tile, b, ty0, tx0);
        const float* in_frame = a.in + (long long)b * a.in_frame_stride + a.in_ch_off;
#pragma unroll
        for (int it = 0; it < HALO_IT; ++it) {
            const int q = it * 256 + tid;
            const int p = q >> 3;
            const int logical = (q & 7) ^ ((p >> 1) & 7);
            const int hy = p / HW_;
            const int hx = p - hy * HW_;
            const int gy = ty0 + hy - PAD, gx = tx0 + hx - PAD;
            const bool inb = (q < HALO_PIECES) && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
            hsrc[it] = inb ? in_frame + ((long long)gy * a.W + gx) * a.in_pix_stride + logical * 4 : a.zero_page + logical * 4;
            hstep[it] = inb ? 32 : 0;
        }
    };
    const unsigned lds0 = og_lds_addr(smem);
    auto stage_halo = [&](int buf, int c) {
        const unsigned base = lds0 + buf * HALO_BYTES + wave * 1024;
#pragma unroll
        for (int it = 0; it < HALO_IT; ++it)
            if (it < HALO_IT - 1 || last_valid) glds16(hsrc[it] + c * hstep[it], base + it * 4096);
    };
    // weights of step (n_tile, c, s): TPS consecutive tap slices
    auto stage_w = [&](int stage, int n_tile, int c, int s) {
        const float* blk = a.wpk + ((long long)(n_tile * a.n_chunks + c) * TAPS + s * TPS) * (WROWS * 32) + tid * 4;
        const unsigned base = lds0 + 2 * HALO_BYTES + stage * SBYTES + wave * 1024;
#pragma unroll
        for (int i = 0; i < NT * TPS; ++i) glds16(blk + i * 1024, base + i * 4096);
    };

    // ---- fragment addressing ----
    const int px0 = 2 * (li >> 2) + (li & 1);
    const int pyl = (li >> 1) & 1;
    const int brow = wn * 32 + li;
    const int boff = brow * 128 + ((lh ^ ((brow >> 1) & 7)) << 4);
    auto a_addr = [&](int m, int tap) -> int {
        const int dy = (MODE == 0) ? tap / 3 : 0;
        const int dx = (MODE == 0) ? tap % 3 : 0;
        const int py = 2 * (wm * MS + m) + pyl + dy;
        const int p = py * HW_ + px0 + dx;
        return p * 128 + ((lh ^ ((p >> 1) & 7)) << 4);
    };

    int item = blockIdx.x;
    if (item >= n_items) return;
    if (a.stamps != nullptr && tid == 0) {
        a.stamps[4 * blockIdx.x + 0] = __builtin_amdgcn_s_memtime();
        // realtime (100 MHz) in the low 40 bits; HW_ID (cu/sh/se/simd...) and XCC_ID above it: lets the probe group workgroups by CU
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);    // HW_REG_HW_ID, 32 bits
        const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);   // HW_REG_XCC_ID, low 4 bits
        a.stamps[4 * blockIdx.x + 1] = (__builtin_amdgcn_s_memrealtime() & 0xFFFFFFFFFFull) |
                                       ((unsigned long long)((hw >> 8) & 0xFFFF) << 40) | ((unsigned long long)(xcc & 0xF) << 56);
    }

    // cursors
    int h_item = item, h_c = c_lo(item);           // next halo unit to stage
    int w_item = item, w_c = c_lo(item), w_s = 0;  // next weight step to stage
    auto ntile_of = [&](int it) { return ((ks_n == 1) ? it : it / ks_n) / a.n_spatial; };
    int w_ntile = ntile_of(item);
    set_hsrc(h_item);

    auto adv_h = [&]() {
        if (++h_c == c_hi(h_item)) {
            h_item += G;
            if (h_item < n_items) {
                h_c = c_lo(h_item);
                set_hsrc(h_item);
            }
        }
    };
    auto adv_w = [&]() {
        if (++w_s == NSTEP_U) {
            w_s = 0;
            if (++w_c == c_hi(w_item)) {
                w_item += G;
                if (w_item < n_items) {
                    w_c = c_lo(w_item);
                    w_ntile = ntile_of(w_item);
                }
            }
        }
    };

    // prologue: unit 0 halo, step 0 weights
    stage_halo(0, h_c);
    adv_h();
    stage_w(0, w_ntile, w_c, 0);
    adv_w();
    og_wait_dma();
    __syncthreads();

    f32x4 a_cur[MS], b_cur;
#pragma unroll
    for (int m = 0; m < MS; ++m) a_cur[m] = *(const f32x4*)(halo0 + a_addr(m, 0));
    b_cur = *(const f32x4*)(wbuf0 + boff);

    int u = 0;   // running unit counter  -> halo buffer parity
    int gs = 0;  // running step counter  -> weight stage parity
    int cached_ntile = -1;
    float sc = 0.f, sh = 0.f;

    while (item < n_items) {
        int n_tile, b, ty0, tx0;
        decode(item, n_tile, b, ty0, tx0);
        const bool last_item = (item + G >= n_items);

        f32x16 acc[MS];
#pragma unroll
        for (int m = 0; m < MS; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;

        const int c_end = c_hi(item);
        for (int c = c_lo(item); c < c_end; ++c) {
            const bool last_unit = last_item && (c + 1 == c_end);
            if (a.prio_mode) {
                // The two workgroups that share a CU do not progress at the same rate (issue arbitration is
                // priority-then-age): the favoured one finishes ~10 % early and the CU then runs one wave per
                // SIMD.  Alternating the priority per unit, in opposite phase for the two, evens them out.
                const int role = (a.prio_mode == 1) ? (int)(blockIdx.x >= (unsigned)(G + 1) / 2) : (int)(blockIdx.x & 1);
                if ((u + role) & 1)
                    __builtin_amdgcn_s_setprio(1);
                else
                    __builtin_amdgcn_s_setprio(0);
            }
#pragma unroll
            for (int s = 0; s < NSTEP_U; ++s) {
                // ---- issue the prefetches that ride behind this step's MFMAs ----
                if (w_item < n_items) {
                    stage_w((gs + 1) & 1, w_ntile, w_c, w_s);
                    adv_w();
                }
                if (s == 0 && h_item < n_items) {
                    stage_halo((u + 1) & 1, h_c);
                    adv_h();
                }
                const unsigned char* hb = halo0 + (u & 1) * HALO_BYTES;
                const unsigned char* wb = wbuf0 + (gs & 1) * SBYTES;
                const bool has_next = !(last_unit && s == NSTEP_U - 1);
#pragma unroll
                for (int sub = 0; sub < NSUB; ++sub) {
                    f32x4 a_nxt[MS], b_nxt;
                    if (sub < NSUB - 1) {
                        const int tl = (sub + 1) >> 2, j = (sub + 1) & 3;  // tap within step, k8 index
                        const int tap = s * TPS + tl;
#pragma unroll
                        for (int m = 0; m < MS; ++m) a_nxt[m] = *(const f32x4*)(hb + (a_addr(m, tap) ^ (j << 5)));
                        b_nxt = *(const f32x4*)(wb + tl * WBYTES + (boff ^ (j << 5)));
                    } else {
                        // The next step's data was issued >= one step ago: drain the DMA queue, then the
                        // barrier makes it visible to every wave.  sched_barrier keeps every MFMA of the
                        // earlier sub-steps ABOVE the barrier, so the fragment reads they were interleaved
                        // with are long complete and __syncthreads' lgkmcnt(0) is free (hipcc otherwise
                        // sinks MFMAs below the barrier, which then lands right behind fresh ds_reads).
                        __builtin_amdgcn_sched_barrier(0);
                        og_wait_dma();
                        __syncthreads();
                        __builtin_amdgcn_sched_barrier(0);
                        if (has_next) {
                            const int ns = (s + 1 == NSTEP_U) ? 0 : s + 1;
                            const unsigned char* hb2 = (s + 1 == NSTEP_U) ? halo0 + ((u + 1) & 1) * HALO_BYTES : hb;
                            const unsigned char* wb2 = wbuf0 + ((gs + 1) & 1) * SBYTES;
#pragma unroll
                            for (int m = 0; m < MS; ++m) a_nxt[m] = *(const f32x4*)(hb2 + a_addr(m, ns * TPS));
                            b_nxt = *(const f32x4*)(wb2 + boff);
                        } else {
#pragma unroll
                            for (int m = 0; m < MS; ++m) a_nxt[m] = a_cur[m];
                            b_nxt = b_cur;
                        }
                    }
#pragma unroll
                    for (int m = 0; m < MS; ++m) {
                        acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[m].x, b_cur.x, acc[m], 0, 0, 0);
                        acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[m].y, b_cur.y, acc[m], 0, 0, 0);
                        acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[m].z, b_cur.z, acc[m], 0, 0, 0);
                        acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[m].w, b_cur.w, acc[m], 0, 0, 0);
                    }
                    // Schedule: one MFMA, then one fragment ds_read_b128 behind each of the next MS+1
                    // MFMAs, then the remaining MFMAs.  A ds_read issued inside an MFMA's 64-cycle
                    // shadow is free; the same reads bunched ahead of the MFMAs cost ~10 cycles each
                    // (tools/ubench/mfma_f32_issue.hip: 67.6 -> 64.0 cycles per MFMA).
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
#pragma unroll
                    for (int q = 0; q < MS + 1; ++q) {
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, MS * 4 - (MS + 2), 0);
#pragma unroll
                    for (int m = 0; m < MS; ++m) a_cur[m] = a_nxt[m];
                    b_cur = b_nxt;
                }
                ++gs;
            }
            ++u;
        }

        // ---- epilogue ----
        if (ks_n > 1) {
            // split-K: raw accumulators, register order, lane-contiguous (256-B stores); summed by k_splitk_epilogue
            float* pw = a.partial + ((long long)item * 4 + wave) * (MS * 16 * 64) + lane;
#pragma unroll
            for (int m = 0; m < MS; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) pw[(m * 16 + r) * 64] = acc[m][r];
        } else {
            if (n_tile != cached_ntile) {  // ordinary loads drain the LDS-DMA queue: keep them rare
                const int ncol = n_tile * WROWS + wn * 32 + li;
                const int co = (MODE == 1) ? ncol % a.aff_mod : ncol;
                sc = a.scale[co];
                sh = a.shift[co];
                cached_ntile = n_tile;
            }
            // scratch: the halo buffer of the unit just finished (dead; the next unit's halo sits in the other one)
            unsigned char* const scr = halo0 + ((u + 1) & 1) * HALO_BYTES + wave * ((MODE == 0) ? 5120 : 4096);  // no pooled tile without MODE 0: 4 x 4 KB fits the 16 KB halo
            if (a.act == 1) conv_epilogue_b<NT, MODE, TH, 1, false>(a, acc, n_tile, b, ty0, tx0, wm, wn, li, lh, sc, sh, scr);
            else if (a.act == 2 && a.res != nullptr) conv_epilogue_b<NT, MODE, TH, 2, true>(a, acc, n_tile, b, ty0, tx0, wm, wn, li, lh, sc, sh, scr);
            else if (a.act == 2) conv_epilogue_b<NT, MODE, TH, 2, false>(a, acc, n_tile, b, ty0, tx0, wm, wn, li, lh, sc, sh, scr);
            else conv_epilogue_b<NT, MODE, TH, 0, false>(a, acc, n_tile, b, ty0, tx0, wm, wn, li, lh, sc, sh, scr);
            // the next item's first step DMAs the unit-after-next's halo into this very buffer: every wave must be
            // done with its scratch first
            __syncthreads();
        }
        item += G;
    }
    if (a.stamps != nullptr && tid == 0) {
        a.stamps[4 * blockIdx.x + 2] = __builtin_amdgcn_s_memtime();
        a.stamps[4 * blockIdx.x + 3] = __builtin_amdgcn_s_memrealtime() & 0xFFFFFFFFFFull;
    }
}

// Second pass of split-K: one workgroup per (n_tile, frame, tile), same thread roles as the conv
// kernel; sums the K-split partials IN SPLIT ORDER (deterministic) and runs the shared epilogue.
template <int NT, int MODE, int TH>
__global__ __launch_bounds__(256) void k_splitk_epilogue(ConvArgs a) {
    constexpr int TW = 16;
    constexpr int WROWS = 32 * NT;
    constexpr int WM = 4 / NT;
    constexpr int MS = (TH / 2) / WM;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave % NT, wm = wave / NT, li = lane & 31, lh = lane >> 5;
    const int base = blockIdx.x;
    const int n_tile = base / a.n_spatial;
    int sp = base - n_tile * a.n_spatial;
    const int tiles_per_frame = a.tiles_x * a.tiles_y;
    const int b = sp / tiles_per_frame;
    sp -= b * tiles_per_frame;
    const int tyi = sp / a.tiles_x;
    const int ty0 = tyi * TH, tx0 = (sp - tyi * a.tiles_x) * TW;
    f32x16 acc[MS];
#pragma unroll
    for (int m = 0; m < MS; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    for (int s = 0; s < a.ksplit; ++s) {
        const float* pw = a.partial + (((long long)base * a.ksplit + s) * 4 + wave) * (MS * 16 * 64) + lane;
#pragma unroll
        for (int m = 0; m < MS; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][r] += pw[(m * 16 + r) * 64];
    }
    const int ncol = n_tile * WROWS + wn * 32 + li;
    const int co = (MODE == 1) ? ncol % a.aff_mod : ncol;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    conv_epilogue<NT, MODE, TH>(a, acc, n_tile, b, ty0, tx0, wm, wn, li, lh, a.scale[co], a.shift[co], smem + wave * 5120);
}

// First layer: Conv2d(1, f0, 3, pad 1) + BN + ReLU straight from the u8 frame
// (fuses `inp.astype(float32) / 255.0`, utils.py:235) or from an f32 NCHW input
// (UNet.__call__ parity path).  Bandwidth-bound (AI 4.4 F/B): plain
// VALU, 8 lanes per pixel x 4 channels each -> 1 KiB fully coalesced stores.
template <typename IN_T, bool HOUT = false>
__global__ __launch_bounds__(256) void k_conv_first(const IN_T* __restrict__ in, float* __restrict__ out,
                                                    const float* __restrict__ w9,  // [9][Cp]
                                                    const float* __restrict__ scale, const float* __restrict__ shift,
                                                    int H, int W, int Cp, int out_pix_stride, long long out_frame_stride,
                                                    int* range_flag = nullptr) {
    __shared__ float tile[18][20];
    float first_absmax = 0.f;
    const int tiles_x = (W + 15) >> 4, tiles_y = (H + 15) >> 4;
    int sp = blockIdx.x;
    const int b = sp / (tiles_x * tiles_y);
    sp -= b * tiles_x * tiles_y;
    const int ty0 = (sp / tiles_x) * 16, tx0 = (sp % tiles_x) * 16;
    const IN_T* fin = in + (long long)b * H * W;
    for (int q = threadIdx.x; q < 18 * 18; q += 256) {
        const int hy = q / 18, hx = q - hy * 18;
        const int gy = ty0 + hy - 1, gx = tx0 + hx - 1;
        float v = 0.f;
        if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
            if (sizeof(IN_T) == 1)
                v = (float)fin[(long long)gy * W + gx] / 255.0f;  // IEEE-correct f32 divide, as numpy's
            else
                v = (float)fin[(long long)gy * W + gx];
        }
        tile[hy][hx] = v;
    }
    __syncthreads();
    const int cq = threadIdx.x & 7;
    float* fout = out + (long long)b * out_frame_stride;
    for (int cg = 0; cg < Cp; cg += 32) {
        if (HOUT) {
            // split-precision output (H layout): lane cq writes 16-byte slot cq of the pixel's 128-byte chunk = hi (cq < 4)
            // or lo (cq >= 4) of channels 8(cq&3)..+7; the same fma chain per channel as below
            const int c8 = cg + (cq & 3) * 8;
            for (int pp = threadIdx.x >> 3; pp < 256; pp += 32) {
                const int py = pp >> 4, px = pp & 15;
                const int y = ty0 + py, x = tx0 + px;
                _Float16 hl[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float sv = 0.f;
#pragma unroll
                    for (int t = 0; t < 9; ++t) sv = fmaf(tile[py + t / 3][px + t % 3], w9[t * Cp + c8 + e], sv);
                    const float v = fmaxf(fmaf(sv, scale[c8 + e], shift[c8 + e]), 0.f);
                    _Float16 hi, lo;
                    og_split(v, hi, lo);
                    hl[e] = (cq < 4) ? hi : lo;
                    first_absmax = fmaxf(first_absmax, v);
                }
                if (y < H && x < W) *(f32x4*)(fout + ((long long)y * W + x) * out_pix_stride + cg + cq * 4) = og_pack8(hl);
            }
            continue;
        }
        const int c0 = cg + cq * 4;
        f32x4 wv[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) wv[t] = *(const f32x4*)(w9 + t * Cp + c0);
        const f32x4 sc = *(const f32x4*)(scale + c0);
        const f32x4 sh = *(const f32x4*)(shift + c0);
        for (int pp = threadIdx.x >> 3; pp < 256; pp += 32) {
            const int py = pp >> 4, px = pp & 15;
            const int y = ty0 + py, x = tx0 + px;
            f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const float xv = tile[py + t / 3][px + t % 3];
                s.x = fmaf(xv, wv[t].x, s.x);
                s.y = fmaf(xv, wv[t].y, s.y);
                s.z = fmaf(xv, wv[t].z, s.z);
                s.w = fmaf(xv, wv[t].w, s.w);
            }
            f32x4 o;
            o.x = fmaxf(fmaf(s.x, sc.x, sh.x), 0.f);
            o.y = fmaxf(fmaf(s.y, sc.y, sh.y), 0.f);
            o.z = fmaxf(fmaf(s.z, sc.z, sh.z), 0.f);
            o.w = fmaxf(fmaf(s.w, sc.w, sh.w), 0.f);
            if (y < H && x < W) *(f32x4*)(fout + ((long long)y * W + x) * out_pix_stride + c0) = o;
        }
    }
    if (HOUT) og_flag_range(first_absmax, range_flag);
}

// Head: Conv2d(f0, 1, 1) + bias -> logit (unet.py:72,88); sigmoid; > threshold;
// {0,255} mask (utils.py:237,241); per-frame area = #(mask>0) [inside box]
// (features.py:238,244-245).  8 lanes per pixel, xor-shuffle reduction.
template <bool HIN = false>
__global__ __launch_bounds__(256) void k_head(const float* __restrict__ in, long long in_frame_stride, int in_pix_stride,
                                              const float* __restrict__ w, float bias, int Cp, int HW, int W,
                                              float threshold, const int32_t* __restrict__ boxes,
                                              float* __restrict__ logits, uint8_t* __restrict__ mask,
                                              int32_t* __restrict__ area, int blocks_per_frame) {
    const int b = blockIdx.x / blocks_per_frame;
    const int blk = blockIdx.x - b * blocks_per_frame;
    const int cq = threadIdx.x & 7;
    const float* fin = in + (long long)b * in_frame_stride;
    int bx1 = 0, by1 = 0, bx2 = 1 << 30, by2 = 1 << 30;
    if (boxes != nullptr) {
        bx1 = boxes[b * 4 + 0];
        by1 = boxes[b * 4 + 1];
        bx2 = boxes[b * 4 + 2];
        by2 = boxes[b * 4 + 3];
        if (bx1 < 0) { bx2 = 0; by2 = 0; bx1 = 0; by1 = 0; }  // "no detection" -> area 0 (features.py:241-242)
        // python slice semantics for mask[y1:y2, x1:x2] with non-negative bounds
    }
    int cnt = 0;
    const int pix_per_block = 1024;
    for (int pp = threadIdx.x >> 3; pp < pix_per_block; pp += 32) {
        const int p = blk * pix_per_block + pp;
        if (p < HW) {
            float s = 0.f;
            for (int c = cq * 4; c < Cp; c += 32) {
                if (HIN) {   // H layout: this lane's 16-byte slot holds hi (cq < 4) or lo (cq >= 4, scaled by 2^11) of channels 8(cq&3)..+7
                    const og_h8 hv = __builtin_bit_cast(og_h8, *(const f32x4*)(fin + (long long)p * in_pix_stride + c));
                    const float* wp = w + (c - cq * 4) + (cq & 3) * 8;
                    float part = 0.f;
#pragma unroll
                    for (int e = 0; e < 8; ++e) part = fmaf((float)hv[e], wp[e], part);
                    s += (cq < 4) ? part : part * OG_LO_INV;
                    continue;
                }
                const f32x4 xv = *(const f32x4*)(fin + (long long)p * in_pix_stride + c);
                const f32x4 wv = *(const f32x4*)(w + c);
                s = fmaf(xv.x, wv.x, s);
                s = fmaf(xv.y, wv.y, s);
                s = fmaf(xv.z, wv.z, s);
                s = fmaf(xv.w, wv.w, s);
            }
            s += __shfl_xor(s, 1);
            s += __shfl_xor(s, 2);
            s += __shfl_xor(s, 4);
            if (cq == 0) {
                const float lg = s + bias;
                const float prob = 1.0f / (1.0f + expf(-lg));
                const bool on = prob > threshold;
                if (logits) logits[(long long)b * HW + p] = lg;
                if (mask) mask[(long long)b * HW + p] = on ? 255 : 0;
                const int y = p / W, x = p - y * W;
                cnt += (on && x >= bx1 && x < bx2 && y >= by1 && y < by2) ? 1 : 0;
            }
        }
    }
    if (area != nullptr) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) cnt += __shfl_down(cnt, o);
        if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(&area[b], cnt);
    }
}

// Box-gated recount over a resident {0,255} mask (features.py:244-245).
__global__ __launch_bounds__(256) void k_mask_area(const uint8_t* __restrict__ mask, int HW, int W,
                                                   const int32_t* __restrict__ boxes, int32_t* __restrict__ area,
                                                   int blocks_per_frame) {
    const int b = blockIdx.x / blocks_per_frame;
    const int blk = blockIdx.x - b * blocks_per_frame;
    int bx1 = 0, by1 = 0, bx2 = 1 << 30, by2 = 1 << 30;
    if (boxes != nullptr) {
        bx1 = boxes[b * 4 + 0];
        by1 = boxes[b * 4 + 1];
        bx2 = boxes[b * 4 + 2];
        by2 = boxes[b * 4 + 3];
        if (bx1 < 0) { bx2 = 0; by2 = 0; bx1 = 0; by1 = 0; }
    }
    int cnt = 0;
    for (int p = blk * 4096 + threadIdx.x; p < min(HW, (blk + 1) * 4096); p += 256) {
        const int y = p / W, x = p - y * W;
        cnt += (mask[(long long)b * HW + p] > 0 && x >= bx1 && x < bx2 && y >= by1 && y < by2) ? 1 : 0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_down(cnt, o);
    if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(&area[b], cnt);
}

// cv2.cvtColor(BGR2GRAY) for u8: OpenCV's 15-bit fixed point, coefficients
// R 9798 / G 19235 / B 3735, (x + 2^14) >> 15.  OpenCV is absent in the build
// image: this restates its published algorithm; parity unpinned.
__global__ __launch_bounds__(256) void k_bgr2gray(const uint8_t* __restrict__ bgr, uint8_t* __restrict__ gray, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        const int bb = bgr[3 * i], gg = bgr[3 * i + 1], rr = bgr[3 * i + 2];
        gray[i] = (uint8_t)((bb * 3735 + gg * 19235 + rr * 9798 + (1 << 14)) >> 15);
    }
}

// =======================================================================================
// YOLOv8n detector kernels (rows D2 / Y1-Y6 of SURVEY §8a).  The network itself is
// third-party (ultralytics, not vendored by the reference): architecture restated from its
// published yolov8.yaml / modules — parity unpinned.  3x3 s1 and 1x1 convolutions reuse
// k_conv_mfma_p (MODE 0 / MODE 2, SiLU epilogue, optional Bottleneck residual).
// =======================================================================================

// Generic direct convolution (any k, stride, pad; NHWC f32, or the u8 BGR frame with the
// predictor's BGR->RGB swap and /255 fused).  Used for the seven stride-2 convs (~15 % of the
// detector's MACs); 8 lanes per output pixel x 4 output channels each.
// Weights: [k*k][Cin][Cout_p] (Cout contiguous).
template <bool IN_U8, int PX = 4>
__global__ __launch_bounds__(256) void k_conv_direct(const void* __restrict__ in_, long long in_frame_stride, int in_pix_stride,
                                                     int in_ch_off, int Hin, int Win, int Cin, const float* __restrict__ w,
                                                     const float* __restrict__ scale, const float* __restrict__ shift,
                                                     int Cout_p, float* __restrict__ out, long long out_frame_stride,
                                                     int out_pix_stride, int out_ch_off, int Hout, int Wout, int ks, int stride,
                                                     int pad, int act, int total_quads /* B*Hout*ceil(Wout/PX) */,
                                                     int cq_shift /* log2(channel quads worked on per 32-channel group): 3 = all
                                                                     eight; 2 when only 16 of the 32 padded channels are real, ... */) {
    // thread = (PX = 4 consecutive output pixels along x) x (4 output channels): 16 accumulators, so one
    // weight float4 and four input values feed 16 FMAs (the first version did 4 FMAs per 2 loads).
    // PX = 1 (one-frame calls): four times the threads, a quarter of the serial loads each; same fma chain per output.
    // Padded output channels (zero weights; their slots stay at the arena's initial zeros) get no thread at all.
    const int cq = threadIdx.x & ((1 << cq_shift) - 1);
    const int q = (blockIdx.x * 256 + threadIdx.x) >> cq_shift;
    if (q >= total_quads) return;
    const int qpr = (Wout + PX - 1) / PX;  // quads per output row
    const int row = q / qpr;
    const int ox0 = (q - row * qpr) * PX;
    const int b = row / Hout;
    const int oy = row - b * Hout;
    {
        const int c0 = blockIdx.y * 32 + cq * 4;  // one 32-channel output group per blockIdx.y: small maps still fill the chip
        f32x4 acc[PX];
#pragma unroll
        for (int j = 0; j < PX; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int ky = 0; ky < ks; ++ky) {
            const int iy = oy * stride - pad + ky;
            if (iy < 0 || iy >= Hin) continue;
            for (int kx = 0; kx < ks; ++kx) {
                const float* wp = w + (long long)((ky * ks + kx) * Cin) * Cout_p + c0;
                int ix[PX];
                bool ok[PX];
#pragma unroll
                for (int j = 0; j < PX; ++j) {
                    ix[j] = (ox0 + j) * stride - pad + kx;
                    ok[j] = ix[j] >= 0 && ix[j] < Win && (ox0 + j) < Wout;
                }
                if (IN_U8) {
                    const uint8_t* rowp = (const uint8_t*)in_ + ((long long)b * Hin + iy) * Win * 3;
#pragma unroll
                    for (int ci = 0; ci < 3; ++ci) {
                        const f32x4 wv = *(const f32x4*)(wp + ci * Cout_p);
#pragma unroll
                        for (int j = 0; j < PX; ++j) {
                            const float xv = ok[j] ? (float)rowp[ix[j] * 3 + 2 - ci] / 255.0f : 0.f;  // RGB[ci] = BGR[2-ci]
                            acc[j].x = fmaf(xv, wv.x, acc[j].x);
                            acc[j].y = fmaf(xv, wv.y, acc[j].y);
                            acc[j].z = fmaf(xv, wv.z, acc[j].z);
                            acc[j].w = fmaf(xv, wv.w, acc[j].w);
                        }
                    }
                } else {
                    const float* rowp = (const float*)in_ + (long long)b * in_frame_stride + (long long)iy * Win * in_pix_stride + in_ch_off;
                    for (int ci = 0; ci < Cin; ci += 4) {
                        f32x4 xv[PX];
#pragma unroll
                        for (int j = 0; j < PX; ++j)
                            xv[j] = ok[j] ? *(const f32x4*)(rowp + (long long)ix[j] * in_pix_stride + ci) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            if (ci + e < Cin) {
                                const f32x4 wv = *(const f32x4*)(wp + (long long)(ci + e) * Cout_p);
#pragma unroll
                                for (int j = 0; j < PX; ++j) {
                                    acc[j].x = fmaf(xv[j][e], wv.x, acc[j].x);
                                    acc[j].y = fmaf(xv[j][e], wv.y, acc[j].y);
                                    acc[j].z = fmaf(xv[j][e], wv.z, acc[j].z);
                                    acc[j].w = fmaf(xv[j][e], wv.w, acc[j].w);
                                }
                            }
                        }
                    }
                }
            }
        }
        const f32x4 sc = *(const f32x4*)(scale + c0);
        const f32x4 sh = *(const f32x4*)(shift + c0);
#pragma unroll
        for (int j = 0; j < PX; ++j) {
            if (ox0 + j < Wout) {
                f32x4 o;
                o.x = og_act(fmaf(acc[j].x, sc.x, sh.x), act);
                o.y = og_act(fmaf(acc[j].y, sc.y, sh.y), act);
                o.z = og_act(fmaf(acc[j].z, sc.z, sh.z), act);
                o.w = og_act(fmaf(acc[j].w, sc.w, sh.w), act);
                *(f32x4*)(out + (long long)b * out_frame_stride + ((long long)oy * Wout + ox0 + j) * out_pix_stride + out_ch_off + c0) = o;
            }
        }
    }
}

// SPPF's MaxPool2d(5, stride 1, pad 2) (-inf padding), channel segment -> channel segment of one buffer.
__global__ __launch_bounds__(256) void k_maxpool5(const float* __restrict__ buf_in, float* __restrict__ buf_out, long long frame_stride,
                                                  int pix_stride, int in_off, int out_off, int C4 /*channels/4*/, int H, int W, long long total) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const int c = (int)(t % C4) * 4;
    long long p = t / C4;
    const int x = (int)(p % W);
    p /= W;
    const int y = (int)(p % H);
    const long long b = p / H;
    f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    for (int dy = -2; dy <= 2; ++dy) {
        const int yy = y + dy;
        if (yy < 0 || yy >= H) continue;
        for (int dx = -2; dx <= 2; ++dx) {
            const int xx = x + dx;
            if (xx < 0 || xx >= W) continue;
            const f32x4 v = *(const f32x4*)(buf_in + b * frame_stride + ((long long)yy * W + xx) * pix_stride + in_off + c);
            m.x = fmaxf(m.x, v.x);
            m.y = fmaxf(m.y, v.y);
            m.z = fmaxf(m.z, v.z);
            m.w = fmaxf(m.w, v.w);
        }
    }
    *(f32x4*)(buf_out + b * frame_stride + ((long long)y * W + x) * pix_stride + out_off + c) = m;
}

// SPPF's three chained 5x5 pools in ONE launch (one-frame calls on small maps): a workgroup keeps a 32-channel slab of the
// whole map in LDS and pools it three times (segment j -> segment j+1 of the concat buffer, j = 0..2).  max() is exact, so
// the result is the chain's, value for value.  LDS: 2 x H*W*128 B (host checks H*W <= 224).
__global__ __launch_bounds__(256) void k_sppf_pools(float* __restrict__ buf, long long frame_stride, int pix_stride, int seg /*channels per segment*/,
                                                    int H, int W) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    f32x4* cur = (f32x4*)smem;
    f32x4* nxt = cur + H * W * 8;
    const int slab = blockIdx.x, b = blockIdx.y, n = H * W * 8;
    float* base = buf + (long long)b * frame_stride + slab * 32;
    for (int i = threadIdx.x; i < n; i += 256) cur[i] = *(const f32x4*)(base + (long long)(i >> 3) * pix_stride + (i & 7) * 4);
    __syncthreads();
    for (int j = 1; j <= 3; ++j) {
        for (int i = threadIdx.x; i < n; i += 256) {
            const int p = i >> 3, q = i & 7, y = p / W, x = p - y * W;
            f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
            for (int dy = -2; dy <= 2; ++dy) {
                const int yy = y + dy;
                if (yy < 0 || yy >= H) continue;
                for (int dx = -2; dx <= 2; ++dx) {
                    const int xx = x + dx;
                    if (xx < 0 || xx >= W) continue;
                    const f32x4 v = cur[(yy * W + xx) * 8 + q];
                    m.x = fmaxf(m.x, v.x);
                    m.y = fmaxf(m.y, v.y);
                    m.z = fmaxf(m.z, v.z);
                    m.w = fmaxf(m.w, v.w);
                }
            }
            nxt[i] = m;
            *(f32x4*)(base + (long long)p * pix_stride + j * seg + q * 4) = m;
        }
        __syncthreads();
        f32x4* t = cur;
        cur = nxt;
        nxt = t;
    }
}

// nn.Upsample(scale_factor=2, mode="nearest"): out[y][x] = in[y/2][x/2], into a channel segment.
__global__ __launch_bounds__(256) void k_upsample2(const float* __restrict__ in, long long in_frame_stride, int in_pix_stride, int in_off,
                                                   float* __restrict__ out, long long out_frame_stride, int out_pix_stride,
                                                   int out_off, int C4, int Hout, int Wout, long long total) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const int c = (int)(t % C4) * 4;
    long long p = t / C4;
    const int x = (int)(p % Wout);
    p /= Wout;
    const int y = (int)(p % Hout);
    const long long b = p / Hout;
    const f32x4 v = *(const f32x4*)(in + b * in_frame_stride + ((long long)(y >> 1) * (Wout >> 1) + (x >> 1)) * in_pix_stride + in_off + c);
    *(f32x4*)(out + b * out_frame_stride + ((long long)y * Wout + x) * out_pix_stride + out_off + c) = v;
}

// Detect head decode: DFL (softmax expectation over 16 bins per side), dist2bbox around the
// anchor centre (x+0.5, y+0.5), x stride, sigmoid class score; then the per-frame arg-max
// confidence box >= conf_thres.  NMS never changes which box has the highest confidence, and
// TemporalDetector consumes exactly `boxes.xyxy[boxes.conf.argmax()]` (detector.py:62-64), so the
// top-1 needs no NMS.  One workgroup per frame.  pred [B][A][5] = x1,y1,x2,y2,conf (optional);
// best [B][5] (conf = -1 when nothing passes).
struct YoloLevel {
    const float* box;   // [B,h,w,*] 64 DFL logits at ch 0
    const float* cls;   // [B,h,w,*] class logit at ch 0
    long long box_frame_stride, cls_frame_stride;
    int box_pix_stride, cls_pix_stride;
    int h, w;
    float stride;
};
struct YoloDecodeArgs {
    YoloLevel lv[3];
    int n_anchors;
    float conf_thres;
    float img_w, img_h;  // clip range (scale_boxes -> clip_boxes for the identity letterbox)
    float* pred;
    float* best;
};
__global__ __launch_bounds__(256) void k_yolo_decode(YoloDecodeArgs a) {
    __shared__ float s_conf[256];
    __shared__ int s_idx[256];
    const int b = blockIdx.x;
    float best_c = -1.f;
    int best_i = 0x7fffffff;
    float bx[4] = {0, 0, 0, 0};
    for (int i = threadIdx.x; i < a.n_anchors; i += 256) {
        int l = 0, j = i;
        if (j >= a.lv[0].h * a.lv[0].w) {
            j -= a.lv[0].h * a.lv[0].w;
            l = 1;
            if (j >= a.lv[1].h * a.lv[1].w) {
                j -= a.lv[1].h * a.lv[1].w;
                l = 2;
            }
        }
        const YoloLevel& L = a.lv[l];
        const int y = j / L.w, x = j - y * L.w;
        const float* bp = L.box + (long long)b * L.box_frame_stride + (long long)j * L.box_pix_stride;
        float d[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            float v[16], mx = -INFINITY;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                v[q] = bp[s * 16 + q];
                mx = fmaxf(mx, v[q]);
            }
            float se = 0.f, sw = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const float e = expf(v[q] - mx);
                se += e;
                sw += e * (float)q;
            }
            d[s] = sw / se;
        }
        const float ax = (float)x + 0.5f, ay = (float)y + 0.5f;
        const float x1a = ax - d[0], y1a = ay - d[1], x2a = ax + d[2], y2a = ay + d[3];
        const float cx = (x1a + x2a) * 0.5f * L.stride, cy = (y1a + y2a) * 0.5f * L.stride;
        const float ww = (x2a - x1a) * L.stride, hh = (y2a - y1a) * L.stride;
        const float lg = L.cls[(long long)b * L.cls_frame_stride + (long long)j * L.cls_pix_stride];
        const float conf = 1.0f / (1.0f + expf(-lg));
        float r[4] = {cx - ww * 0.5f, cy - hh * 0.5f, cx + ww * 0.5f, cy + hh * 0.5f};
        r[0] = fminf(fmaxf(r[0], 0.f), a.img_w);
        r[2] = fminf(fmaxf(r[2], 0.f), a.img_w);
        r[1] = fminf(fmaxf(r[1], 0.f), a.img_h);
        r[3] = fminf(fmaxf(r[3], 0.f), a.img_h);
        if (a.pred) {
            float* pp = a.pred + ((long long)b * a.n_anchors + i) * 5;
            pp[0] = r[0]; pp[1] = r[1]; pp[2] = r[2]; pp[3] = r[3]; pp[4] = conf;
        }
        if (conf > a.conf_thres && (conf > best_c)) {  // strict >: the first (lowest index) maximum wins, as argmax
            best_c = conf;
            best_i = i;
            bx[0] = r[0]; bx[1] = r[1]; bx[2] = r[2]; bx[3] = r[3];
        }
    }
    s_conf[threadIdx.x] = best_c;
    s_idx[threadIdx.x] = best_i;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) {
            const float c2 = s_conf[threadIdx.x + o];
            const int i2 = s_idx[threadIdx.x + o];
            if (c2 > s_conf[threadIdx.x] || (c2 == s_conf[threadIdx.x] && i2 < s_idx[threadIdx.x])) {
                s_conf[threadIdx.x] = c2;
                s_idx[threadIdx.x] = i2;
            }
        }
        __syncthreads();
    }
    const int win = s_idx[0];
    if (s_conf[0] < 0.f) {
        if (threadIdx.x == 0) {
            float* o = a.best + (long long)b * 5;
            o[0] = o[1] = o[2] = o[3] = 0.f;
            o[4] = -1.f;
        }
    } else if (best_i == win && best_c == s_conf[0]) {
        float* o = a.best + (long long)b * 5;
        o[0] = bx[0]; o[1] = bx[1]; o[2] = bx[2]; o[3] = bx[3]; o[4] = best_c;
    }
}

// One-frame calls: the same decode spread over ceil(A / 64) workgroups per frame -- thread = (anchor, box side), so a
// thread reads 16 DFL logits instead of 64 and 21 workgroups work instead of one.  Per-anchor arithmetic is the single-
// workgroup kernel's, expression for expression; every workgroup leaves its best candidate in `cand`, the last one to
// arrive (arrival counter, reset for the next launch) picks the winner by (confidence, lowest index), as above.
__global__ __launch_bounds__(256) void k_yolo_decode_mb(YoloDecodeArgs a, float* __restrict__ cand /*[B][nblk][8]*/, int* __restrict__ counter /*[B]*/) {
    __shared__ float s_conf[64];
    __shared__ int s_idx[64];
    __shared__ int s_last;
    const int b = blockIdx.y, nblk = gridDim.x;
    const int la = threadIdx.x >> 2, sd = threadIdx.x & 3;
    const int i = blockIdx.x * 64 + la;
    float conf = -1.f;
    float r[4] = {0, 0, 0, 0};
    bool cand_ok = false;
    {
        const bool live = i < a.n_anchors;
        int l = 0, j = live ? i : 0;
        if (j >= a.lv[0].h * a.lv[0].w) {
            j -= a.lv[0].h * a.lv[0].w;
            l = 1;
            if (j >= a.lv[1].h * a.lv[1].w) {
                j -= a.lv[1].h * a.lv[1].w;
                l = 2;
            }
        }
        const YoloLevel& L = a.lv[l];
        const int y = j / L.w, x = j - y * L.w;
        const float* bp = L.box + (long long)b * L.box_frame_stride + (long long)j * L.box_pix_stride;
        float v[16], mx = -INFINITY;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            v[q] = bp[sd * 16 + q];
            mx = fmaxf(mx, v[q]);
        }
        float se = 0.f, sw = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const float e = expf(v[q] - mx);
            se += e;
            sw += e * (float)q;
        }
        const float dmine = sw / se;
        float d[4];
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_) d[s_] = __shfl(dmine, (threadIdx.x & 60) + s_, 64);
        const float ax = (float)x + 0.5f, ay = (float)y + 0.5f;
        const float x1a = ax - d[0], y1a = ay - d[1], x2a = ax + d[2], y2a = ay + d[3];
        const float cx = (x1a + x2a) * 0.5f * L.stride, cy = (y1a + y2a) * 0.5f * L.stride;
        const float ww = (x2a - x1a) * L.stride, hh = (y2a - y1a) * L.stride;
        const float lg = L.cls[(long long)b * L.cls_frame_stride + (long long)j * L.cls_pix_stride];
        const float cf = 1.0f / (1.0f + expf(-lg));
        r[0] = cx - ww * 0.5f; r[1] = cy - hh * 0.5f; r[2] = cx + ww * 0.5f; r[3] = cy + hh * 0.5f;
        r[0] = fminf(fmaxf(r[0], 0.f), a.img_w);
        r[2] = fminf(fmaxf(r[2], 0.f), a.img_w);
        r[1] = fminf(fmaxf(r[1], 0.f), a.img_h);
        r[3] = fminf(fmaxf(r[3], 0.f), a.img_h);
        if (live && sd == 0) {
            if (a.pred) {
                float* pp = a.pred + ((long long)b * a.n_anchors + i) * 5;
                pp[0] = r[0]; pp[1] = r[1]; pp[2] = r[2]; pp[3] = r[3]; pp[4] = cf;
            }
            if (cf > a.conf_thres) {
                conf = cf;
                cand_ok = true;
            }
        }
    }
    if (sd == 0) {
        s_conf[la] = conf;
        s_idx[la] = cand_ok ? i : 0x7fffffff;
    }
    __syncthreads();
    for (int o = 32; o > 0; o >>= 1) {
        if (threadIdx.x < o) {
            const float c2 = s_conf[threadIdx.x + o];
            const int i2 = s_idx[threadIdx.x + o];
            if (c2 > s_conf[threadIdx.x] || (c2 == s_conf[threadIdx.x] && i2 < s_idx[threadIdx.x])) {
                s_conf[threadIdx.x] = c2;
                s_idx[threadIdx.x] = i2;
            }
        }
        __syncthreads();
    }
    float* mine = cand + ((long long)b * nblk + blockIdx.x) * 8;
    if (s_conf[0] < 0.f) {
        if (threadIdx.x == 0) {
            __hip_atomic_store(mine + 4, -1.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store((int*)mine + 5, 0x7fffffff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    } else if (sd == 0 && cand_ok && i == s_idx[0]) {
#pragma unroll
        for (int k = 0; k < 4; ++k) __hip_atomic_store(mine + k, r[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(mine + 4, conf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store((int*)mine + 5, i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the candidate is written through before the arrival is counted
    __syncthreads();
    if (threadIdx.x == 0) s_last = (atomicAdd(counter + b, 1) == nblk - 1) ? 1 : 0;
    __syncthreads();
    if (!s_last) return;
    if (threadIdx.x < 64) {
        float bc = -1.f;
        int bi = 0x7fffffff, bk = -1;
        for (int k = threadIdx.x; k < nblk; k += 64) {
            const float c2 = __hip_atomic_load(cand + ((long long)b * nblk + k) * 8 + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int i2 = __hip_atomic_load((const int*)(cand + ((long long)b * nblk + k) * 8) + 5, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (c2 > bc || (c2 == bc && i2 < bi)) {
                bc = c2;
                bi = i2;
                bk = k;
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float c2 = __shfl_down(bc, o, 64);
            const int i2 = __shfl_down(bi, o, 64), k2 = __shfl_down(bk, o, 64);
            if (c2 > bc || (c2 == bc && i2 < bi)) {
                bc = c2;
                bi = i2;
                bk = k2;
            }
        }
        if (threadIdx.x == 0) {
            float* o = a.best + (long long)b * 5;
            if (bc < 0.f) {
                o[0] = o[1] = o[2] = o[3] = 0.f;
                o[4] = -1.f;
            } else {
                const float* w = cand + ((long long)b * nblk + bk) * 8;
#pragma unroll
                for (int k = 0; k < 4; ++k) o[k] = __hip_atomic_load(w + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                o[4] = bc;
            }
            counter[b] = 0;   // ready for the next launch on this stream
        }
    }
}

// =======================================================================================
// YOLO-Crop+UNet geometry on the device (scripts/eval_girafe.py:127-159, utils.py:97-186):
// crop the box, letterbox it NEAREST into a size x size tile (zero padded), and the inverse
// (unletterbox NEAREST + paste into a zero frame).  Index rule = OpenCV INTER_NEAREST as restated in
// openglottal_amd/geometry.py: src = min(floor(dst * src_len / dst_len), src_len - 1), in float64.
// geom[b] = {pad_top, pad_left, content_h, content_w} is computed on the host (Python's round()).
// A box with x2 <= x1 or y2 <= y1 (or x1 < 0) yields an all-zero tile / frame.
// =======================================================================================
__device__ __forceinline__ int og_nearest(int d, int src_len, int dst_len) {
    const int s = (int)floor((double)d * ((double)src_len / (double)dst_len));
    return s < src_len - 1 ? s : src_len - 1;
}

// A box / geometry record a kernel may index with: box inside the frame, content rectangle inside the tile.  Anything
// else (a raw TemporalDetector box passed without clamping, a stale geom) yields zeros instead of an out-of-bounds access.
__device__ __forceinline__ bool og_crop_ok(int x1, int y1, int x2, int y2, int top, int left, int ch, int cw, int H, int W, int size) {
    return x1 >= 0 && y1 >= 0 && x2 > x1 && y2 > y1 && x2 <= W && y2 <= H && top >= 0 && left >= 0 && ch > 0 && cw > 0 &&
           top + ch <= size && left + cw <= size;
}

__global__ __launch_bounds__(256) void k_crop_letterbox(const uint8_t* __restrict__ gray, int H, int W, const int32_t* __restrict__ boxes,
                                                        const int32_t* __restrict__ geom, int size, uint8_t* __restrict__ tiles) {
    const int b = blockIdx.y;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= size * size) return;
    const int ty = t / size, tx = t - ty * size;
    const int x1 = boxes[b * 4], y1 = boxes[b * 4 + 1], x2 = boxes[b * 4 + 2], y2 = boxes[b * 4 + 3];
    const int top = geom[b * 4], left = geom[b * 4 + 1], ch = geom[b * 4 + 2], cw = geom[b * 4 + 3];
    uint8_t v = 0;
    const int cy = ty - top, cx = tx - left;
    if (og_crop_ok(x1, y1, x2, y2, top, left, ch, cw, H, W, size) && cy >= 0 && cy < ch && cx >= 0 && cx < cw) {
        const int sy = og_nearest(cy, y2 - y1, ch), sx = og_nearest(cx, x2 - x1, cw);
        v = gray[((long long)b * H + y1 + sy) * W + x1 + sx];
    }
    tiles[(long long)b * size * size + t] = v;
}

__global__ __launch_bounds__(256) void k_unletterbox_paste(const uint8_t* __restrict__ tiles, int size, const int32_t* __restrict__ boxes,
                                                           const int32_t* __restrict__ geom, int H, int W, uint8_t* __restrict__ out) {
    const int b = blockIdx.y;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= H * W) return;
    const int y = p / W, x = p - y * W;
    const int x1 = boxes[b * 4], y1 = boxes[b * 4 + 1], x2 = boxes[b * 4 + 2], y2 = boxes[b * 4 + 3];
    const int top = geom[b * 4], left = geom[b * 4 + 1], ch = geom[b * 4 + 2], cw = geom[b * 4 + 3];
    uint8_t v = 0;
    if (og_crop_ok(x1, y1, x2, y2, top, left, ch, cw, H, W, size) && x >= x1 && x < x2 && y >= y1 && y < y2) {
        const int hh = y2 - y1, ww = x2 - x1;
        const int sy = (ch == hh) ? (y - y1) : og_nearest(y - y1, ch, hh);
        const int sx = (cw == ww) ? (x - x1) : og_nearest(x - x1, cw, ww);
        v = tiles[(long long)b * size * size + (long long)(top + sy) * size + left + sx];
    }
    out[(long long)b * H * W + p] = v;
}

// =======================================================================================
// BAGLS front end on the device (scripts/eval_bagls.py:46-70,153-155): frames and GT masks of mixed sizes are scaled so
// that the longest side equals `size` and padded symmetrically (letterbox).  2-D arrays (gray frames, masks) are
// resampled INTER_NEAREST, BGR frames INTER_LINEAR, as the reference does (eval_bagls.py:57).  Index / coefficient rules
// = OpenCV's as restated in openglottal_amd/geometry.py (resize_nearest / resize_linear, u8 path: half-pixel centres,
// 11-bit coefficients, ((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2): the two are compared pixel for pixel.
// Frames are packed back to back in one buffer; frame b starts at offsets[b] and is shapes[b] = {H, W}.
// geom[b] = {pad_top, pad_left, content_h, content_w} (host: Python round()).
// =======================================================================================
__device__ __forceinline__ void og_linear_tap(int d, int src_len, int dst_len, int& i0, int& i1, int& a0, int& a1) {
    const double f = ((double)d + 0.5) * ((double)src_len / (double)dst_len) - 0.5;
    int k = (int)floor(f);
    float frac = (float)(f - (double)k);
    if (k < 0) { k = 0; frac = 0.f; }
    if (k >= src_len - 1) { k = src_len - 1; frac = 0.f; }
    i0 = k;
    i1 = (k + 1 < src_len) ? k + 1 : src_len - 1;
    a1 = (int)rintf(frac * 2048.0f);
    a0 = 2048 - a1;
}

template <int C>
__global__ __launch_bounds__(256) void k_canvas_letterbox(const uint8_t* __restrict__ packed, const long long* __restrict__ offsets,
                                                          const int32_t* __restrict__ shapes, const int32_t* __restrict__ geom, int size,
                                                          int value, uint8_t* __restrict__ out) {
    const int b = blockIdx.y;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= size * size) return;
    const int ty = t / size, tx = t - ty * size;
    const int H = shapes[b * 2], W = shapes[b * 2 + 1];
    const int top = geom[b * 4], left = geom[b * 4 + 1], ch = geom[b * 4 + 2], cw = geom[b * 4 + 3];
    const uint8_t* src = packed + offsets[b];
    uint8_t* o = out + ((long long)b * size * size + t) * C;
    const int cy = ty - top, cx = tx - left;
    const bool ok = H > 0 && W > 0 && ch > 0 && cw > 0 && top >= 0 && left >= 0 && top + ch <= size && left + cw <= size;
    if (!ok || cy < 0 || cy >= ch || cx < 0 || cx >= cw) {
#pragma unroll
        for (int c = 0; c < C; ++c) o[c] = (uint8_t)value;
        return;
    }
    if (C == 1) {
        const int sy = og_nearest(cy, H, ch), sx = og_nearest(cx, W, cw);
        o[0] = src[(long long)sy * W + sx];
    } else {
        int x0, x1, ax0, ax1, y0, y1, ay0, ay1;
        og_linear_tap(cx, W, cw, x0, x1, ax0, ax1);
        og_linear_tap(cy, H, ch, y0, y1, ay0, ay1);
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const int r0 = (int)src[((long long)y0 * W + x0) * C + c] * ax0 + (int)src[((long long)y0 * W + x1) * C + c] * ax1;
            const int r1 = (int)src[((long long)y1 * W + x0) * C + c] * ax0 + (int)src[((long long)y1 * W + x1) * C + c] * ax1;
            int v = (((ay0 * (r0 >> 4)) >> 16) + ((ay1 * (r1 >> 4)) >> 16) + 2) >> 2;
            o[c] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
        }
    }
}

// Per-frame confusion counts of a predicted mask against a ground-truth mask (scripts/eval_bagls.py:75-87 `frame_metrics`
// needs tp, fp, fn): stats[b] = {tp, n_pred, n_gt}.  With boxes, the prediction counts only inside the box -- the
// "yolo+unet" row zeroes the mask outside it (eval_bagls.py:201-207); x1 < 0 = no detection = empty prediction.
__global__ __launch_bounds__(256) void k_mask_stats(const uint8_t* __restrict__ pred, const uint8_t* __restrict__ gt, int HW, int W,
                                                    const int32_t* __restrict__ boxes, int32_t* __restrict__ stats, int blocks_per_frame) {
    const int b = blockIdx.x / blocks_per_frame;
    const int blk = blockIdx.x - b * blocks_per_frame;
    int bx1 = 0, by1 = 0, bx2 = 1 << 30, by2 = 1 << 30;
    if (boxes != nullptr) {
        bx1 = boxes[b * 4 + 0];
        by1 = boxes[b * 4 + 1];
        bx2 = boxes[b * 4 + 2];
        by2 = boxes[b * 4 + 3];
        if (bx1 < 0) { bx2 = 0; by2 = 0; bx1 = 0; by1 = 0; }
    }
    int tp = 0, np_ = 0, ng = 0;
    for (int p = blk * 4096 + threadIdx.x; p < min(HW, (blk + 1) * 4096); p += 256) {
        const int y = p / W, x = p - y * W;
        const bool pr = pred[(long long)b * HW + p] > 0 && x >= bx1 && x < bx2 && y >= by1 && y < by2;
        const bool g = gt[(long long)b * HW + p] > 0;
        tp += (pr && g) ? 1 : 0;
        np_ += pr ? 1 : 0;
        ng += g ? 1 : 0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        tp += __shfl_down(tp, o);
        np_ += __shfl_down(np_, o);
        ng += __shfl_down(ng, o);
    }
    if ((threadIdx.x & 63) == 0) {
        if (tp) atomicAdd(&stats[b * 3 + 0], tp);
        if (np_) atomicAdd(&stats[b * 3 + 1], np_);
        if (ng) atomicAdd(&stats[b * 3 + 2], ng);
    }
}
