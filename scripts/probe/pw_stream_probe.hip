// scripts/probe/pw_stream_probe.hip - can a PERSISTENT pointwise kernel keep the memory pipe full?  
//
// ResNet50 FP32's res2 `branch2c + sum` (64 -> 256 channels on 25 088 pixels, f32 NHWC, in-place sum + relu) moves 57.8 MB in 24 us through
// the implicit-GEMM kernels because the workgroups of a round run load / MFMA / epilogue in lockstep (profiles/r04/timeline_f32_pointwise.txt).
// This probe keeps the layer's MEMORY behaviour and drops its arithmetic: a workgroup owns 64 / 32 / 16-pixel tiles (x: PX x C f32 in, y: PX x K f32 read,
// added, relu'd, written back), `grid` workgroups loop over the tiles. Variants:
//   0  one tile per workgroup visit, residual requested after the x tile has been consumed (what the conv kernels do)
//   1  persistent, the NEXT tile's x and residual in flight while the current tile is combined and stored (register double buffer)
// for grid = 1, 2, 3, 4 workgroups per CU. Prints us per pass and TB/s of (x + 2 y) bytes. No product code involved.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/probe/pw_stream_probe.hip -o scripts/probe/_bin/pw_stream_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int C, int K, int PX, bool PIPE>
__global__ __launch_bounds__(256) void pw_stream(const float4* __restrict__ x, float4* __restrict__ y, int tiles, int M) {
    constexpr int XC = (PX * C / 4 + 255) / 256;      // float4 per thread of an x tile (PX pixels)
    constexpr int YC = PX * K / 4 / 256;      // float4 per thread of a y tile
    const int t = threadIdx.x;
    auto load_tile = [&](int tile, float4 (&xv)[XC], float4 (&rv)[YC], bool with_res) {
        const size_t xb = (size_t)tile * (PX * C / 4), yb = (size_t)tile * (PX * K / 4);
#pragma unroll
        for (int j = 0; j < XC; ++j) xv[j] = x[xb + (t + 256 * j) % (PX * C / 4)];
        if (with_res) {
#pragma unroll
            for (int j = 0; j < YC; ++j) rv[j] = y[yb + t + 256 * j];
        }
    };
    auto finish = [&](int tile, const float4 (&xv)[XC], float4 (&rv)[YC], bool res_loaded) {
        const size_t yb = (size_t)tile * (PX * K / 4);
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < XC; ++j) s += xv[j].x + xv[j].y + xv[j].z + xv[j].w;      // stands in for the conv: the x tile is consumed
        if (!res_loaded) {
#pragma unroll
            for (int j = 0; j < YC; ++j) rv[j] = y[yb + t + 256 * j];
        }
#pragma unroll
        for (int j = 0; j < YC; ++j) {
            float4 o = rv[j];
            o.x = fmaxf(o.x + s, 0.f); o.y = fmaxf(o.y + s, 0.f); o.z = fmaxf(o.z + s, 0.f); o.w = fmaxf(o.w + s, 0.f);
            y[yb + t + 256 * j] = o;
        }
    };
    (void)M;
    if constexpr (!PIPE) {
        for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
            float4 xv[XC], rv[YC];
            load_tile(tile, xv, rv, false);
            finish(tile, xv, rv, false);
        }
    } else {
        float4 xa[XC], ra[YC], xb_[XC], rb[YC];
        int tile = blockIdx.x;
        if (tile >= tiles) return;
        load_tile(tile, xa, ra, true);
        for (;;) {
            const int n1 = tile + gridDim.x;
            if (n1 < tiles) load_tile(n1, xb_, rb, true);
            finish(tile, xa, ra, true);
            if (n1 >= tiles) break;
            const int n2 = n1 + gridDim.x;
            if (n2 < tiles) load_tile(n2, xa, ra, true);
            finish(n1, xb_, rb, true);
            if (n2 >= tiles) break;
            tile = n2;
        }
    }
}

template <int C, int K, int PX>
static void run_case(const char* name, int M, hipStream_t st) {
    const int tiles = M / PX;
    float4 *x, *y, *big;
    CK(hipMalloc(&x, (size_t)M * C * 4));
    CK(hipMalloc(&y, (size_t)M * K * 4));
    CK(hipMalloc(&big, (size_t)384 << 20));
    CK(hipMemset(x, 0, (size_t)M * C * 4));
    CK(hipMemset(y, 0, (size_t)M * K * 4));
    CK(hipMemset(big, 0, (size_t)384 << 20));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const double mb = ((double)M * C * 4 + 2.0 * M * K * 4) / 1e6;
    for (int pipe = 0; pipe < 2; ++pipe)
        for (int per_cu : {1, 2, 3, 4, 0}) {
            const int grid = per_cu ? 256 * per_cu : tiles;      // 0: one workgroup per tile (the conv kernels' launch shape)
            auto launch = [&] {
                if (pipe) hipLaunchKernelGGL((pw_stream<C, K, PX, true>), dim3(grid), dim3(256), 0, st, x, y, tiles, M);
                else hipLaunchKernelGGL((pw_stream<C, K, PX, false>), dim3(grid), dim3(256), 0, st, x, y, tiles, M);
            };
            launch();
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < 30; ++i) launch();
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double warm = ms * 1e3 / 30;
            double cold = 0;
            for (int i = 0; i < 6; ++i) {      // after a 384 MB sweep: operands from HBM
                CK(hipMemsetAsync(big, i, (size_t)384 << 20, st));
                CK(hipEventRecord(e0, st));
                launch();
                CK(hipEventRecord(e1, st));
                CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (i) cold += ms * 1e3 / 5;
            }
            printf("%-28s %-26s grid %5d: back to back %6.2f us = %4.1f TB/s | after a 384 MB sweep %6.2f us (one event pair)\n", name,
                   pipe ? "persistent, next tile ahead" : "residual after the x tile", grid, warm, mb / warm, cold);
        }
    CK(hipFree(x)); CK(hipFree(y)); CK(hipFree(big));
}

int main() {
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    run_case<64, 256, 64>("res2 64->256 + sum, b8", 8 * 56 * 56, st);
    run_case<128, 512, 32>("res3 128->512 + sum, b8", 8 * 28 * 28, st);
    run_case<256, 1024, 16>("res4 256->1024 + sum, b8", 8 * 14 * 14, st);
    return 0;
}
