// Questions behind the opt-in split-precision mode (f16 hi/lo, 3 MFMAs per f32 product, f32 accumulate):
//  (1) does v_mfma_f32_32x32x16_f16 keep f16 SUBNORMAL inputs (the `lo` halves of small values) or flush them?
//  (2) what does a bare 3-MFMA-per-step loop deliver on random data (clock give-back included)?
//  (3) how exact is hi*hi + hi*lo + lo*hi against an fmaf chain on random data?
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

__global__ void k_denorm(float* out) {
    h8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)9.5367431640625e-07f; b[j] = (_Float16)1024.0f; }   // 2^-20 (subnormal) x 2^10
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = acc[0];       // 16 k-steps x 2^-10 = 2^-6 = 0.015625 when subnormals are kept
    h8 c;
    for (int j = 0; j < 8; ++j) c[j] = (_Float16)6.103515625e-05f;   // smallest normal
    f32x16 acc2;
    for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
    acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(c, c, acc2, 0, 0, 0);   // products 2^-28: f32 result, 16 x 2^-28 = 2^-24
    if (threadIdx.x == 0) out[1] = acc2[0];
}

template <int NACC>
__global__ __launch_bounds__(256) void k_rate(float* out, int iters, const h8* src) {
    f32x16 acc[NACC], cor[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) { acc[i][r] = 0.f; cor[i][r] = 0.f; }
    const h8 ah = src[threadIdx.x], al = src[256 + threadIdx.x], bh = src[512 + threadIdx.x], bl = src[768 + threadIdx.x];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[i], 0, 0, 0);
            cor[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, cor[i], 0, 0, 0);
            cor[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, cor[i], 0, 0, 0);
        }
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r] + cor[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// (2b): the conv kernel's mix: per 24 MFMAs (4 sub-tiles x 6), 4 B-fragment + 16 A-fragment ds_read_b128 of random data
__global__ __launch_bounds__(256) void k_rate_lds(float* out, int iters, const h8* src) {
    __shared__ h8 lds[4096];   // 64 KB
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = src[i & 1023];
    __syncthreads();
    f32x16 acc[4], cor[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) { acc[i][r] = 0.f; cor[i][r] = 0.f; }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int off = w * 1024 + lane;
    for (int it = 0; it < iters; ++it) {
        h8 bv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) bv[j] = lds[(off + 64 * j) & 4095];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            h8 av[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) av[j] = lds[(off + 256 + 64 * (4 * m + j)) & 4095];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[t], bv[t], acc[m], 0, 0, 0);
                cor[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[t], bv[2 + t], cor[m], 0, 0, 0);
                cor[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[2 + t], bv[t], cor[m], 0, 0, 0);
            }
        }
        off = (off + 1344) & 4095;
    }
    float s = 0;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r] + cor[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// (3): one 32x32 tile, K = 4608: split product vs double reference
__global__ void k_exact(const float* A, const float* B, int K, float* D_split, float* D_single) {   // A [32][K], B [K][32]
    const int l = threadIdx.x, r = l & 31, h = l >> 5;
    f32x16 acc, cor;
    for (int i = 0; i < 16; ++i) { acc[i] = 0.f; cor[i] = 0.f; }
    for (int k0 = 0; k0 < K; k0 += 16) {
        h8 ah, al, bh, bl;
        for (int j = 0; j < 8; ++j) {
            const float a = A[r * K + k0 + 8 * h + j], b = B[(k0 + 8 * h + j) * 32 + r];
            ah[j] = (_Float16)a; al[j] = (_Float16)((a - (float)ah[j]) * 2048.0f);
            bh[j] = (_Float16)b; bl[j] = (_Float16)((b - (float)bh[j]) * 2048.0f);
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
        cor = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, cor, 0, 0, 0);
        cor = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, cor, 0, 0, 0);
    }
    for (int i = 0; i < 16; ++i) {
        const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
        D_split[row * 32 + r] = fmaf(cor[i], 1.0f / 2048.0f, acc[i]);
        D_single[row * 32 + r] = acc[i];
    }
}

int main() {
    float* out; hipMalloc(&out, 1 << 22);
    hipLaunchKernelGGL(k_denorm, dim3(1), dim3(64), 0, 0, out);
    float h[2]; hipMemcpy(h, out, 8, hipMemcpyDeviceToHost);
    printf("(1) subnormal f16 inputs: 16 x (2^-20 x 2^10) = %.9g (0.015625 = kept, 0 = flushed); 16 x (2^-14)^2 = %.9g (5.96e-08 expected)\n", h[0], h[1]);
    // (2)
    std::vector<_Float16> hs(1024 * 8);
    srand(1);
    for (auto& v : hs) v = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 2.f);
    h8* src; hipMalloc(&src, hs.size() * 2); hipMemcpy(src, hs.data(), hs.size() * 2, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wg = 1; wg <= 2; ++wg) {
        const int iters = 20000, grid = 256 * wg;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k_rate<2>, dim3(grid), dim3(256), 0, 0, out, iters, src);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double mfma = (double)grid * 4 * iters * 2 * 3;
        printf("(2) 3 MFMA(32x32x16 f16)/step, %d wave(s)/SIMD, random data: %.1f TFLOP/s raw f16 = %.1f TFLOP/s of f32-equivalent products (x%.2f of 157.3)\n",
               wg, mfma * 32768 / ms / 1e9, mfma / 3 * 32768 / ms / 1e9, mfma / 3 * 32768 / ms / 1e9 / 157.3);
    }
    for (int wg = 1; wg <= 2; ++wg) {
        const int iters = 2000, grid = 256 * wg;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k_rate_lds, dim3(grid), dim3(256), 0, 0, out, iters, src);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double mfma = (double)grid * 4 * iters * 24;
        printf("(2b) same with the conv kernel's LDS traffic (20 ds_read_b128 per 24 MFMAs), %d wave(s)/SIMD: %.1f TFLOP/s raw f16 = %.1f f32-equivalent\n",
               wg, mfma * 32768 / ms / 1e9, mfma / 3 * 32768 / ms / 1e9);
    }
    // (3)
    const int K = 4608;
    std::vector<float> A(32 * K), B(K * 32);
    for (auto& v : A) v = fmaxf(0.f, (rand() / (float)RAND_MAX - 0.3f) * 3.f);     // ReLU-like activations
    for (auto& v : B) v = (rand() / (float)RAND_MAX - 0.5f) * 0.072f;              // He-uniform weights at K = 4608
    float *dA, *dB, *dS, *d1;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dS, 4096); hipMalloc(&d1, 4096);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_exact, dim3(1), dim3(64), 0, 0, dA, dB, K, dS, d1);
    std::vector<float> S(1024), S1(1024); hipMemcpy(S.data(), dS, 4096, hipMemcpyDeviceToHost); hipMemcpy(S1.data(), d1, 4096, hipMemcpyDeviceToHost);
    double e_split = 0, e_single = 0, e_f32 = 0, mag = 0;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
        double ref = 0; float f = 0.f;
        for (int k = 0; k < K; ++k) { ref += (double)A[i * K + k] * B[k * 32 + j]; f = fmaf(A[i * K + k], B[k * 32 + j], f); }
        e_split = fmax(e_split, fabs(S[i * 32 + j] - ref)); e_single = fmax(e_single, fabs(S1[i * 32 + j] - ref));
        e_f32 = fmax(e_f32, fabs(f - ref)); mag = fmax(mag, fabs(ref));
    }
    printf("(3) K=%d dot products, |max| %.3g: max abs error  f32 fmaf chain %.3g | f16 hi/lo split (3 MFMA) %.3g | f16 single MFMA %.3g\n", K, mag, e_f32, e_split, e_single);
    return 0;
}
