// v_pk_fma_f32 with a normal pair as src0 and low-half broadcasts as src1/src2 (op_sel_hi:[1,0,0]) vs fmaf.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
__global__ void k(const f32x2* xs, const float* sc, const float* sh, f32x2* out_pk, f32x2* out_sc) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const f32x2 x = xs[i];
    const float s = sc[i & 31], h = sh[i & 31];
    const f32x2 s2 = {s, s}, h2 = {h, h};
    out_pk[i] = __builtin_elementwise_fma(x, s2, h2);
    asm volatile("" ::: "memory");
    out_sc[i] = f32x2{fmaf(x.x, s, h), fmaf(x.y, s, h)};
}
int main() {
    const int n = 8192;
    std::vector<float> hx(n * 2), hs(32), hh(32);
    unsigned s = 1;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xFFFF) / 32768.0f - 1.0f; };
    for (auto& v : hx) v = rnd();
    for (auto& v : hs) v = rnd();
    for (auto& v : hh) v = rnd();
    float *dx, *ds, *dh, *dp, *dq;
    hipMalloc(&dx, n * 8); hipMalloc(&ds, 128); hipMalloc(&dh, 128); hipMalloc(&dp, n * 8); hipMalloc(&dq, n * 8);
    hipMemcpy(dx, hx.data(), n * 8, hipMemcpyHostToDevice);
    hipMemcpy(ds, hs.data(), 128, hipMemcpyHostToDevice);
    hipMemcpy(dh, hh.data(), 128, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, (const f32x2*)dx, ds, dh, (f32x2*)dp, (f32x2*)dq);
    std::vector<float> p(n * 2), q(n * 2);
    hipMemcpy(p.data(), dp, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(q.data(), dq, n * 8, hipMemcpyDeviceToHost);
    int bad = 0, badlo = 0, badhi = 0;
    for (int i = 0; i < n * 2; ++i) if (p[i] != q[i]) { ++bad; (i & 1) ? ++badhi : ++badlo; }
    printf("op_sel_hi:[1,0,0] pattern: %d of %d differ (low halves %d, high halves %d)\n", bad, n * 2, badlo, badhi);
    return 0;
}
