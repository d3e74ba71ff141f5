// Per-workgroup timeline of the occupancy conv kernel on a synthetic layer: entry / prologue done / main loop
// done / stores done (s_memtime) + HW_ID, dumped raw for tools/ubench/occ_timeline.py.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I openglottal_amd/csrc tools/ubench/occ_timeline.hip -o tools/ubench/occ_timeline
//   occ_timeline <variant: n1 | n2t8 | n2t16> <B> <HW> <n_chunks> <out.bin>
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "og_kernels.hpp"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
    if (argc < 6) { printf("usage: occ_timeline n1|n2t8|n2t16 B HW n_chunks out.bin [prio_mode]\n"); return 2; }
    const std::string var = argv[1];
    const int B = atoi(argv[2]), HW = atoi(argv[3]), nch = atoi(argv[4]);
    const int NT = (var == "n1") ? 1 : 2, TH = (var == "n2t16") ? 16 : 8;
    const int Cin = 32 * nch, Cout = 32 * NT;  // one column tile: the timeline of a workgroup does not depend on how many there are
    if (B < 1 || B > 256 || HW % 16 || HW < 16 || HW > 256 || nch < 1 || nch > 16) { printf("bad shape\n"); return 2; }
    const size_t n_in = (size_t)B * HW * HW * Cin, n_out = (size_t)B * HW * HW * Cout, n_w = (size_t)nch * 9 * 32 * NT * 32;
    std::vector<float> hin(n_in), hw(n_w);
    uint32_t s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xFFFF) / 65536.0f - 0.5f; };
    for (auto& v : hin) v = rnd();
    for (auto& v : hw) v = rnd() * 0.1f;
    float *din, *dout, *dw, *dsc, *dsh, *dz;
    unsigned long long* dst;
    CK(hipMalloc(&din, n_in * 4)); CK(hipMalloc(&dout, n_out * 4)); CK(hipMalloc(&dw, n_w * 4));
    CK(hipMalloc(&dsc, 256)); CK(hipMalloc(&dsh, 256)); CK(hipMalloc(&dz, 256));
    CK(hipMemcpy(din, hin.data(), n_in * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dw, hw.data(), n_w * 4, hipMemcpyHostToDevice));
    std::vector<float> one(64, 1.f);
    CK(hipMemcpy(dsc, one.data(), 256, hipMemcpyHostToDevice));
    CK(hipMemset(dsh, 0, 256)); CK(hipMemset(dz, 0, 256));
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.in = din; a.in_frame_stride = (long long)HW * HW * Cin; a.in_pix_stride = Cin; a.n_chunks = nch;
    a.H = HW; a.W = HW; a.tiles_x = HW / 16; a.tiles_y = HW / TH; a.n_spatial = B * a.tiles_x * a.tiles_y;
    a.wpk = dw; a.scale = dsc; a.shift = dsh; a.aff_mod = Cout;
    a.out = dout; a.out_frame_stride = (long long)HW * HW * Cout; a.out_pix_stride = Cout;
    a.zero_page = dz; a.act = 1; a.ksplit = 1; a.zdiv = 1; a.zrcp = 1.0f; a.zgroup_shift = 0; a.frames = B;
    a.prio_mode = (argc > 6) ? atoi(argv[6]) : 0;
    const int grid = a.n_spatial;
    CK(hipMalloc(&dst, (size_t)grid * 64));
    CK(hipMemset(dst, 0, (size_t)grid * 64));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms_plain = 0, ms_st = 0;
    for (int rep = 0; rep < 4; ++rep) {   // reps 0-2 without stamps (timing reference), rep 3 with
        a.stamps = (rep == 3) ? dst : nullptr;
        CK(hipEventRecord(e0));
        if (var == "n1") {
            const int lds = 18 * 10 * 128 + 3 * 32 * 128;
            CK(hipFuncSetAttribute((const void*)k_conv_mfma_o<1, 0, 8, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            hipLaunchKernelGGL((k_conv_mfma_o<1, 0, 8, 3>), dim3(a.tiles_x, a.tiles_y, B), dim3(256), lds, 0, a);
        } else if (var == "n2t8") {
            const int lds = 18 * 10 * 128 + 3 * 64 * 128;
            CK(hipFuncSetAttribute((const void*)k_conv_mfma_o<2, 0, 8, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            hipLaunchKernelGGL((k_conv_mfma_o<2, 0, 8, 3>), dim3(a.tiles_x, a.tiles_y, B), dim3(256), lds, 0, a);
        } else {
            const int lds = 18 * 18 * 128 + 3 * 64 * 128;
            CK(hipFuncSetAttribute((const void*)k_conv_mfma_o<2, 0, 16, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            hipLaunchKernelGGL((k_conv_mfma_o<2, 0, 16, 2>), dim3(a.tiles_x, a.tiles_y, B), dim3(256), lds, 0, a);
        }
        CK(hipGetLastError());
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep == 2) ms_plain = ms;
        if (rep == 3) ms_st = ms;
    }
    std::vector<unsigned long long> st((size_t)grid * 8);
    CK(hipMemcpy(st.data(), dst, st.size() * 8, hipMemcpyDeviceToHost));
    FILE* f = fopen(argv[5], "wb");
    if (!f) { printf("cannot open %s\n", argv[5]); return 1; }
    fwrite(st.data(), 8, st.size(), f);
    fclose(f);
    const double fl = 2.0 * B * HW * HW * 9.0 * Cin * Cout;
    printf("{\"variant\": \"%s\", \"B\": %d, \"HW\": %d, \"n_chunks\": %d, \"grid\": %d, \"ms\": %.4f, \"ms_with_stamps\": %.4f, \"tflops\": %.1f}\n",
           var.c_str(), B, HW, nch, grid, ms_plain, ms_st, fl / (ms_plain * 1e-3) / 1e12);
    return 0;
}
