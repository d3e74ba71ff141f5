// Does v_pk_fma_f32 with op_sel broadcasts (as hipcc emits them for {x,x} splats fed from a 64-bit load) agree with
// scalar fmaf on gfx950?   hipcc --offload-arch=gfx950 -O3 tools/ubench/pk_fma_check.hip -o tools/ubench/pk_fma_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* xs, const f32x4* ws, f32x4* out_pk, f32x4* out_sc, int taps) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const f32x2* xp = (const f32x2*)(xs + (size_t)i * 10);   // 64-bit loads: both halves get broadcast (op_sel / op_sel_hi)
    const f32x2 p0 = xp[0], p1 = xp[1], p2 = xp[2], p3 = xp[3];
    const float xv[8] = {p0.x, p0.y, p1.x, p1.y, p2.x, p2.y, p3.x, p3.y};
    f32x2 s01 = {0.f, 0.f}, s23 = {0.f, 0.f};     // first fma takes the inline constant 0 as its addend
    float a = 0.f, b = 0.f, c = 0.f, d = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const f32x4 wv = ws[t];
        const f32x2 x2 = {xv[t], xv[t]};
        s01 = __builtin_elementwise_fma(x2, f32x2{wv.x, wv.y}, s01);
        s23 = __builtin_elementwise_fma(x2, f32x2{wv.z, wv.w}, s23);
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const f32x4 wv = ws[t];
        asm volatile("" ::: "memory");
        a = fmaf(xv[t], wv.x, a); b = fmaf(xv[t], wv.y, b); c = fmaf(xv[t], wv.z, c); d = fmaf(xv[t], wv.w, d);
    }
    out_pk[i] = f32x4{s01.x, s01.y, s23.x, s23.y};
    out_sc[i] = f32x4{a, b, c, d};
}
int main() {
    const int n = 4096, taps = 9;
    std::vector<float> hx(n * 10), hw(taps * 4);
    unsigned s = 1;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xFFFF) / 32768.0f - 1.0f; };
    for (auto& v : hx) v = rnd();
    for (auto& v : hw) v = rnd();
    float *dx, *dw, *dp, *ds;
    hipMalloc(&dx, hx.size() * 4); hipMalloc(&dw, hw.size() * 4); hipMalloc(&dp, n * 16); hipMalloc(&ds, n * 16);
    hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, (const f32x4*)dw, (f32x4*)dp, (f32x4*)ds, taps);
    std::vector<float> p(n * 4), q(n * 4);
    hipMemcpy(p.data(), dp, n * 16, hipMemcpyDeviceToHost);
    hipMemcpy(q.data(), ds, n * 16, hipMemcpyDeviceToHost);
    int bad = 0; double worst = 0;
    for (int i = 0; i < n * 4; ++i) if (p[i] != q[i]) { ++bad; worst = std::fmax(worst, std::fabs(p[i] - q[i])); }
    printf("v_pk_fma_f32 vs fmaf: %d of %d differ, max |diff| %.3g\n", bad, n * 4, worst);
    return bad != 0;
}
