// anakin_amd/csrc/conv1x1_pw.hip - FP32 pointwise (1x1 / stride 1) convolution with few input channels (C = 64 / 128) on the bf16 matrix
// cores WITHOUT LDS and WITHOUT barriers: persistent, fully independent waves that keep their output channels' weight planes in registers.
//
// Role: SaberConv2D<AK_FLOAT> / SaberConvEltwise<AK_FLOAT> on ResNet's wide, shallow layers - res2's `branch2c` (64 -> 256) + the in-place
// sum + relu, res3's (128 -> 512), res2a's `branch2a` - the role of the reference's SaberConv1X1 with beta = 1 (saber/funcs/impl/x86/
// saber_conv_eltwise.cpp:40-151) and of NV's conv_gemm_k1s1p0 with beta (third-party/sass/include/sass_funcs.h:699-781).
//
// Why (round-4 verdict item 3): these layers move 58 MB for 0.8 GFLOP; the implicit-GEMM kernels run them at 2.4 TB/s whichever
// tile is chosen, although the same access pattern without arithmetic streams at 7.1 TB/s (profiles/r04/pw_stream_probe.txt). The in-kernel
// timeline showed workgroups of a round running load -> LDS -> barrier -> MFMA -> epilogue in lockstep. Here there is nothing to wait
// for but the wave's own loads: a wave owns TILES x 16 output channels for the whole launch, holds their three weight planes in MFMA
// A-operand registers (C = 64: 4 tiles x 2 slabs x 3 planes = 96 VGPRs; C = 128: 2 tiles x 4 x 3), and walks over 16-pixel groups with the
// next D groups' activations AND residual values already requested; a lane splits its pixel's f32 activations into the three bf16 planes
// in registers (x = h + m + l exactly), six plane products per 32-deep slab in mma_step3's order, issued term-major over the wave's
// accumulators. Prototyped and measured in round 4 (scripts/probe/pw_direct_probe.hip -> profiles/r04/pw_direct_probe.txt: res2's
// branch2c + sum 24.2 -> 13.0 - 14.2 us back to back, res3's 21.2 -> 16.1); this is that kernel with the product's epilogue.
//
// Weights (api_conv.hip: set_weights, saber_hip_conv::d_wpw): [64-channel block][16-row tile i][32-deep slab s][plane][lane] x 8 bf16; row
// r = lane & 15 of tile i is output channel  block * 64 + 16 i + r  (natural order: a store instruction covers 64 contiguous bytes of a
// pixel), element j of lane (r, kg = lane >> 4) is input channel  32 s + (j < 4 ? 4 kg + j : 16 + 4 kg + j - 4)  (a load instruction covers
// 64 contiguous bytes of a pixel: two 16-byte loads per slab). Same k order on the activation side, so the products pair up.
// Epilogue = epilogue_f32 of conv_igemm_impl.h: d = acc; [d += y_old]; d += bias; relu / leaky. The accumulation order over input channels
// differs from the implicit-GEMM kernels' (and from MKL's): inside the 1e-4 FP32 tolerance, like every FP32 kernel here.
#include "conv_igemm_impl.h"

namespace saber_mi355x {

template <int C, int TILES, int D, bool SUM>
__global__ __launch_bounds__(256) void conv1x1_pw_kernel(const ConvKArgs a) {
    constexpr int NS = C / 32, NB = 4 / TILES;          // NB: waves (channel groups) per 64 output channels
    constexpr int R = D + 1;                            // register buffers: D groups in flight behind the one being combined
    const int M = a.M, K = a.K;
    const float* __restrict__ x = (const float*)a.x;
    float* y = (float*)a.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int frow = lane & 15, fq = lane >> 4;
    const int gw = blockIdx.x * 4 + wave, nw = gridDim.x * 4;
    const int cblocks = (K >> 6) * NB;
    const int cb = gw % cblocks, slot = gw / cblocks, nslots = nw / cblocks;      // (launcher: nw % cblocks == 0)
    const int kb64 = cb / NB, i0 = (cb % NB) * TILES;
    const int kb = kb64 * 64 + i0 * 16 + fq * 4;          // tile i of this wave: channels kb + 16 i .. + 3
    v4i wreg[TILES][NS][3];
    {
        const v4i* wf = (const v4i*)a.w + (size_t)kb64 * (4 * NS * 3 * 64) + lane;
#pragma unroll
        for (int i = 0; i < TILES; ++i)
#pragma unroll
            for (int s = 0; s < NS; ++s)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) wreg[i][s][pl] = wf[(((i0 + i) * NS + s) * 3 + pl) * 64];
    }
    float4 bs[TILES];
#pragma unroll
    for (int i = 0; i < TILES; ++i) bs[i] = a.bias ? *(const float4*)(a.bias + kb + 16 * i) : make_float4(0.f, 0.f, 0.f, 0.f);
    const bool relu = a.relu != 0;
    const float slope = a.neg_slope;
    const int groups = (M + 15) >> 4;
    // requests are UNCONDITIONAL (a group past the end re-reads the last one and is never combined): with a branch around them the
    // compiler's s_waitcnt insertion waits for everything at the first use and the groups in flight overlap nothing
    auto request = [&](int g, float4 (&xv)[NS][2], float4 (&rs)[TILES]) {
        g = g < groups ? g : groups - 1;
        const int p = g * 16 + frow;
        const int pc = p < M ? p : M - 1;
        const float* xr = x + (size_t)pc * C + fq * 4;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            xv[s][0] = *(const float4*)(xr + s * 32);
            xv[s][1] = *(const float4*)(xr + s * 32 + 16);
        }
        if (SUM) {
            const float* yr = y + (size_t)pc * K + kb;
#pragma unroll
            for (int i = 0; i < TILES; ++i) rs[i] = *(const float4*)(yr + 16 * i);
        }
    };
    auto finish = [&](int g, const float4 (&xv)[NS][2], const float4 (&rs)[TILES]) {
        v4i bp[NS][3];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            unsigned h[4], m[4], l[4];
            split3_pair(xv[s][0].x, xv[s][0].y, h[0], m[0], l[0]);
            split3_pair(xv[s][0].z, xv[s][0].w, h[1], m[1], l[1]);
            split3_pair(xv[s][1].x, xv[s][1].y, h[2], m[2], l[2]);
            split3_pair(xv[s][1].z, xv[s][1].w, h[3], m[3], l[3]);
            bp[s][0] = v4i{(int)h[0], (int)h[1], (int)h[2], (int)h[3]};
            bp[s][1] = v4i{(int)m[0], (int)m[1], (int)m[2], (int)m[3]};
            bp[s][2] = v4i{(int)l[0], (int)l[1], (int)l[2], (int)l[3]};
        }
        const int p = g * 16 + frow;
        float* yr = y + (size_t)(p < M ? p : M - 1) * K + kb;
        // term-major: the six plane products of a slab run over the TILES accumulators before the next product, so two consecutive MFMAs
        // never touch the same accumulator (accumulator-major, each waits out its predecessor: conv_igemm_impl.h)
        v4f accs[TILES];
#pragma unroll
        for (int i = 0; i < TILES; ++i) accs[i] = v4f{0.f, 0.f, 0.f, 0.f};
        constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};      // (weights, activations): small terms first
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int i = 0; i < TILES; ++i)
                    accs[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, wreg[i][s][PA[t]]),
                                                                      __builtin_bit_cast(v8bf, bp[s][PB[t]]), accs[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < TILES; ++i) {
            const v4f acc = accs[i];
            float o[4] = {acc[0], acc[1], acc[2], acc[3]};
            const float r4[4] = {rs[i].x, rs[i].y, rs[i].z, rs[i].w};
            const float b4[4] = {bs[i].x, bs[i].y, bs[i].z, bs[i].w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float d = o[r];
                if (SUM) d = __fadd_rn(d, r4[r]);
                d = __fadd_rn(d, b4[r]);
                if (relu) d = d > 0.f ? d : (slope == 0.f ? 0.f : __fmul_rn(d, slope));
                o[r] = d;
            }
            if (p < M) *(float4*)(yr + 16 * i) = make_float4(o[0], o[1], o[2], o[3]);
        }
    };
    float4 xv[R][NS][2], rs[R][TILES];
    if (slot >= groups) return;
#pragma unroll
    for (int j = 0; j < D; ++j) request(slot + j * nslots, xv[j], rs[j]);
    for (int base = 0;; base += R) {
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const int g = slot + (base + j) * nslots;
            if (g >= groups) return;
            request(g + D * nslots, xv[(j + D) % R], rs[(j + D) % R]);
            finish(g, xv[j], rs[j]);
        }
    }
}

// C in {64, 128}, K % 64 == 0, 1x1 / stride 1 / pad 0, NHWC f32 in and out, no pair / pooling epilogue; a.w: the fragment-ordered planes.
bool conv1x1_pw_ok(int c, int k) { return (c == 64 || c == 128) && k >= 64 && k % 64 == 0 && (1024 % ((k >> 6) * (c == 64 ? 1 : 2))) == 0; }
hipError_t launch_conv1x1_pw(const ConvKArgs& a, hipStream_t s) {
    if (!conv1x1_pw_ok(a.C, a.K) || a.kh != 1 || a.kw != 1 || a.stride_h != 1 || a.stride_w != 1 || a.pad_h || a.pad_w || a.out_nchw || a.K2 ||
        a.pool_ow || (a.res_mode != RES_NONE && a.res_mode != RES_SUM_INPLACE))
        return hipErrorInvalidValue;
    // 256 workgroups = 1024 waves: one workgroup per CU (the weight registers leave room for one wave per SIMD at C = 64 / 128 alike)
    const dim3 grid(256), block(256);
    const bool sum = a.res_mode == RES_SUM_INPLACE;
    if (a.C == 64) {
        if (sum) hipLaunchKernelGGL((conv1x1_pw_kernel<64, 4, 1, true>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((conv1x1_pw_kernel<64, 4, 1, false>), grid, block, 0, s, a);
    } else {
        if (sum) hipLaunchKernelGGL((conv1x1_pw_kernel<128, 2, 2, true>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((conv1x1_pw_kernel<128, 2, 2, false>), grid, block, 0, s, a);
    }
    return hipGetLastError();
}

}  // namespace saber_mi355x
