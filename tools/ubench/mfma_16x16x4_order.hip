// Is v_mfma_f32_16x16x4_f32 a k-ordered fmaf chain (k = 0, 1, 2, 3) like v_mfma_f32_32x32x2_f32 (k = 0, 1)?
// A 16-window Winograd variant can only reproduce k_conv_wino's bits if it is: the canonical channel order inside a group of 8
// is (0,4,1,5,2,6,3,7) = two 32x32x2 k pairs per fragment element, and a 16x16x4 step would have to run (0,4,1,5) then (2,6,3,7).
// Prints how many of 256 x 64 outputs match each candidate association, on random operands with cancellation.
// build: hipcc --offload-arch=gfx950 -O2 -o mfma_16x16x4_order mfma_16x16x4_order.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k16(const float* A, const float* B, const float* C, float* D, int reps) {
    // A[16][4 reps] row-major per rep: A[(r*16 + i)*4 + k]; B[(r*4 + k)*16 + j]; C/D[16][16]: row = 4*(lane>>4) + reg, col = lane & 15
    const int lane = threadIdx.x & 63, i = lane & 15, kq = lane >> 4;
    f32x4 acc;
    for (int g = 0; g < 4; ++g) acc[g] = C[(4 * kq + g) * 16 + i];
    for (int r = 0; r < reps; ++r)
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[(r * 16 + i) * 4 + kq], B[(r * 4 + kq) * 16 + i], acc, 0, 0, 0);
    for (int g = 0; g < 4; ++g) D[(4 * kq + g) * 16 + i] = acc[g];
}
__global__ void k32(const float* A, const float* B, const float* C, float* D, int reps) {
    // 32x32x2 on the same data laid out as 16 rows (upper 16 rows zero): A[(r*2 + kk)][...]: reps x 2 steps of k pairs (k0,k1),(k2,k3)
    const int lane = threadIdx.x & 63, i = lane & 31, h = lane >> 5;
    f32x16 acc;
    for (int g = 0; g < 16; ++g) {
        const int row = (g & 3) + 8 * (g >> 2) + 4 * h;
        acc[g] = (row < 16 && i < 16) ? C[row * 16 + i] : 0.f;
    }
    for (int r = 0; r < reps; ++r)
        for (int kk = 0; kk < 2; ++kk) {
            const float a = (i < 16) ? A[(r * 16 + i) * 4 + 2 * kk + h] : 0.f, b = (i < 16) ? B[(r * 4 + 2 * kk + h) * 16 + i] : 0.f;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
    for (int g = 0; g < 16; ++g) {
        const int row = (g & 3) + 8 * (g >> 2) + 4 * h;
        if (row < 16 && i < 16) D[row * 16 + i] = acc[g];
    }
}
int main() {
    const int reps = 64, trials = 256;
    std::vector<float> A(reps * 64), B(reps * 64), C(256), D(256), E(256);
    float *dA, *dB, *dC, *dD, *dE;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, 1024); hipMalloc(&dD, 1024); hipMalloc(&dE, 1024);
    long n = 0, seq = 0, pair = 0, rev = 0, same32 = 0;
    srand(1);
    for (int t = 0; t < trials; ++t) {
        for (auto& v : A) v = (float)rand() / RAND_MAX * 2.f - 1.f;
        for (auto& v : B) v = ((float)rand() / RAND_MAX * 2.f - 1.f) * ((rand() & 7) ? 1.f : 1000.f);
        for (auto& v : C) v = (float)rand() / RAND_MAX - 0.5f;
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dC, C.data(), 1024, hipMemcpyHostToDevice);
        k16<<<1, 64>>>(dA, dB, dC, dD, reps); k32<<<1, 64>>>(dA, dB, dC, dE, reps);
        hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost); hipMemcpy(E.data(), dE, 1024, hipMemcpyDeviceToHost);
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j) {
                float s = C[i * 16 + j], p = C[i * 16 + j], q = C[i * 16 + j];
                for (int r = 0; r < reps; ++r) {
                    const float* a = &A[(r * 16 + i) * 4];
                    auto b = [&](int k) { return B[(r * 4 + k) * 16 + j]; };
                    for (int k = 0; k < 4; ++k) s = fmaf(a[k], b(k), s);                       // sequential chain
                    p = p + (fmaf(a[1], b(1), a[0] * b(0)) + fmaf(a[3], b(3), a[2] * b(2)));   // one pairwise tree
                    for (int k = 3; k >= 0; --k) q = fmaf(a[k], b(k), q);                       // reversed chain
                }
                ++n;
                seq += (s == D[i * 16 + j]); pair += (p == D[i * 16 + j]); rev += (q == D[i * 16 + j]); same32 += (D[i * 16 + j] == E[i * 16 + j]);
            }
    }
    printf("v_mfma_f32_16x16x4_f32 over %ld outputs (K = %d): == fmaf chain k=0..3: %ld | == pairwise tree: %ld | == reversed chain: %ld | == v_mfma_f32_32x32x2_f32 over the same k order: %ld\n",
           n, 4 * reps, seq, pair, rev, same32);
    return (seq == n && same32 == n) ? 0 : 1;
}
