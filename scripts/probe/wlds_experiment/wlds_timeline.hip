// In-kernel phase stamps of conv_wlds.hip on ResNet50's res4 3x3 layer (8 x 14 x 14, 256 -> 256, split-K 4: 208 workgroups x 18 stages):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -DSABER_TIMELINE -mllvm -amdgpu-mfma-vgpr-form -I anakin_amd/csrc \
//         scripts/probe/wlds_experiment/wlds_timeline.hip -o /tmp/wlds_timeline.bin
// Stamps (100 MHz wall clock, wave 0 of every workgroup): 0 entry, 1 stage 0's operands in registers, 2 / 3 before stages 6 / 12, 4 loop done,
// 5 (last arrival only) past the split-K counter, 6 partials summed, 7 done. Random operands: timing only.
#include "conv_wlds.hip"      // (this directory; the kernel includes the product headers: -I anakin_amd/csrc)
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>
__device__ unsigned long long* saber_tl_buf = nullptr;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
using namespace saber_mi355x;
static void* dalloc(size_t n, int fill) {
    void* p = nullptr;
    if (hipMalloc(&p, n) != hipSuccess) return nullptr;
    if (fill >= 0) (void)hipMemset(p, fill, n);
    else {
        std::vector<unsigned> h(n / 4 + 1);
        unsigned s = 12345u;
        for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (s >> 9) | 0x3f000000u; }      // floats in [0.5, 1)
        (void)hipMemcpy(p, h.data(), n, hipMemcpyHostToDevice);
    }
    return p;
}
int main(int argc, char** argv) {
    const int variant = argc > 1 ? atoi(argv[1]) : 1, sh = argc > 2 ? atoi(argv[2]) : 2;
    const int N = 8, H = 14, W = 14, C = 256, K = 256, M = N * H * W;
    ConvKArgs a;
    memset(&a, 0, sizeof a);
    a.x = dalloc((size_t)M * C * 4, -1);
    a.w = dalloc((size_t)(K / 16) * (C / 32) * 9 * 3 * 1024, 0x3c);      // bf16 0x3c3c: small positive values
    a.zero = dalloc(256, 0);
    a.y = dalloc((size_t)M * K * 4, 0);
    a.bias = nullptr;
    a.M = M; a.OH = H; a.OW = W; a.H = H; a.W = W; a.C = C; a.K = K; a.N = N;
    a.inv_ohw = 1.0f / (H * W); a.inv_ow = 1.0f / W;
    a.kh = a.kw = 3; a.stride_h = a.stride_w = 1; a.pad_h = a.pad_w = 1; a.dil_h = a.dil_w = 1;
    a.relu = 1; a.res_mode = RES_NONE; a.ksplit_sh = sh;
    a.part = (float*)dalloc((size_t)(M + 255) * (K + 127) * 8 * 4, 0);
    a.part_ctr = (unsigned*)dalloc(65536, 0);
    a.part_err = nullptr;
    unsigned long long* tl = (unsigned long long*)dalloc((size_t)4096 * 128, 0);
    CK(hipMemcpyToSymbol(HIP_SYMBOL(saber_tl_buf), &tl, sizeof(tl)));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    void* big = dalloc((size_t)192 << 20, 1);
    for (int it = 0; it < 4; ++it) {
        CK(hipMemsetAsync(big, it, (size_t)192 << 20, st));      // push the operands out of the L2s (the Infinity Cache keeps them)
        CK(hipEventRecord(e0, st));
        CK(launch_conv_wlds(variant, a, st));
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
    }
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    int p = 0, dx = 0, mb = 0;
    (void)conv_wlds_variant(variant, &p, &dx, &mb);
    const int tiles = ((M + 64 * p - 1) / (64 * p)) * (K / 64), grid = (8 * ((tiles + 7) / 8)) << sh;
    std::vector<unsigned long long> h((size_t)grid * 16);
    CK(hipMemcpy(h.data(), tl, h.size() * 8, hipMemcpyDeviceToHost));
    unsigned long long t0 = ~0ull;
    for (int b = 0; b < grid; ++b) if (h[b * 16]) t0 = std::min(t0, h[b * 16]);
    printf("conv_wlds variant %d (P %d, DX %d), 8 x 14 x 14, 256 -> 256, 3x3, split %d: %d workgroups, %.2f us by events (cold L2)\n", variant, p, dx, 1 << sh, grid, ms * 1e3);
    const char* nm[8] = {"entry (after the first workgroup's)", "stage 0 in registers", "before stage 6", "before stage 12", "loop done", "last arrival: past the counter",
                         "partials summed", "done"};
    for (int s = 0; s < 8; ++s) {
        std::vector<double> v;
        for (int b = 0; b < grid; ++b) if (h[b * 16] && h[b * 16 + s]) v.push_back((double)(h[b * 16 + s] - t0) / 100.0);
        if (v.empty()) continue;
        std::sort(v.begin(), v.end());
        printf("  %-38s n %4zu  min %6.2f  median %6.2f  max %6.2f us\n", nm[s], v.size(), v.front(), v[v.size() / 2], v.back());
    }
    return 0;
}
