// anakin_amd/csrc/fc_small.hip — INT8 fully-connected layer for SMALL batches (m <= 16 rows), gfx950.
//
// Role: SaberFc / VenderFc<X86,AK_INT8> at inference batch sizes (reference: PackedMKLInt8Gemm::dispatch,
// mkl_packed_int8_gemm.cpp:46-97; VenderFc u8 path, vender_fc.cpp:284-335). Through the implicit-GEMM conv kernel a
// [8 x 2048] x [2048 x 1000] fc gets 16 workgroups that each walk a 4..8-stage reduction (7.8 us at batch 8): the
// layer is a 2 MB weight stream, so it wants every CU pulling a slice of the weights at once. Here
//   workgroup = 16 output channels, its 4 waves split the reduction 4 ways;
//   a wave loads its weight slice [16 rows][k/4] and the matching activation slice STRAIGHT into MFMA operand
//   registers (A: row = lane & 15 = output channel, B: row = lane & 15 = batch row, k-group = lane >> 4; all loads of
//   the wave in flight together: one exposed memory latency), runs v_mfma_i32_16x16x64_i8 over them and the four
//   partial accumulators are summed through LDS (integer sums: exact, order independent).
// Same epilogue arithmetic as the conv kernel's EPI_I8_FC_S8 / EPI_I8_FC_U8 branches (conv_igemm_impl.h).
#include "conv_igemm_impl.h"

namespace saber_mi355x {

// KSW: 64-byte k-steps per wave (reduction = 4 waves x KSW x 64, zero padded weights beyond Kg)
template <int KSW>
__global__ __launch_bounds__(256) void fc_i8_small_kernel(const ConvKArgs a) {
    __shared__ v4i red[3][64];
    SABER_TL_DECL;
    SABER_TL(0);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fq = lane >> 4;
    const int n0 = blockIdx.x * 16;
    const int m = frow < a.M ? frow : a.M - 1;            // rows beyond the batch re-read the last one (results dropped)
    const char* wp = (const char*)a.w + (size_t)(n0 + frow) * a.Kg_pad + (size_t)wave * (KSW * 64) + fq * 16;
    const char* xp = (const char*)a.x + (size_t)m * a.C + (size_t)wave * (KSW * 64) + fq * 16;
    v4i wf[KSW], xf[KSW];
#pragma unroll
    for (int s = 0; s < KSW; ++s) wf[s] = *(const v4i*)(wp + s * 64);
#pragma unroll
    for (int s = 0; s < KSW; ++s) {
        // the activation row is only C bytes long: k-steps beyond it multiply zero weights, fetch the zero page
        const bool in = (wave * KSW + s) * 64 + fq * 16 < a.C;
        xf[s] = *(const v4i*)(in ? xp + s * 64 : (const char*)a.zero);
    }
    const int kb = n0 + fq * 4;
    ChanParams<4> cp;
    load_chan_params<4>(a, kb, cp);
    const int xmask = a.in_u8 ? (int)0x80808080u : 0;
    SABER_TL(1);
    v4i acc = {0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s < KSW; ++s) {
        v4i b = xf[s];
        b.x ^= xmask; b.y ^= xmask; b.z ^= xmask; b.w ^= xmask;
        acc = mma_step(wf[s], b, acc);
    }
    SABER_TL(2);
    if (wave > 0) red[wave - 1][lane] = acc;
    __syncthreads();
    SABER_TL(3);
    if (wave > 0) return;
    acc += red[0][lane];
    acc += red[1][lane];
    acc += red[2][lane];
    // lane: output channels kb..kb+3 of batch row frow
    if (frow >= a.M || kb >= a.K) return;
    float out[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int v = acc[r] + cp.comp[r];
        const float d = (float)v;
        if (a.epi == EPI_I8_FC_S8) out[r] = __fadd_rn(__fmul_rn(d, cp.scale[r]), cp.bias[r]);
        else out[r] = (cp.scale[r] == 1.f) ? d : __fmul_rn(cp.scale[r], d);     // EPI_I8_FC_U8
    }
    float* y = (float*)a.y + (size_t)frow * a.K + kb;
    if (kb + 4 <= a.K && (a.K & 3) == 0) {
        *(float4*)y = make_float4(out[0], out[1], out[2], out[3]);
    } else {
        for (int r = 0; r < 4; ++r)
            if (kb + r < a.K) y[r] = out[r];
    }
    SABER_TL(4);
    SABER_TL_FLUSH();
}

// FP32 fc (VenderFc<X86,AK_FLOAT>, vender_fc.cpp:154-212: out = in W^T + bias) at <= 16 batch rows: the same shape of
// kernel on v_mfma_f32_16x16x4_f32 - 16 outputs per workgroup, the 4 waves split the reduction, weight and activation chunks
// (16 floats per row per step) go straight into MFMA operand registers sixteen steps at a time, partial sums meet in LDS.
// ResNet50's fc (8 x 2048 -> 1000, an 8 MB weight stream): 15.4 us through the implicit-GEMM kernel's 32 workgroups.
__global__ __launch_bounds__(256) void fc_f32_small_kernel(const ConvKArgs a, int ksw) {
    __shared__ v4f redf[3][64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fq = lane >> 4;
    const int n0 = blockIdx.x * 16;
    const int m = frow < a.M ? frow : a.M - 1;
    const int k0 = wave * ksw * 16 + fq * 4;                 // this lane's first reduction index
    const float* wp = (const float*)a.w + (size_t)(n0 + frow) * a.Kg_pad + k0;
    const float* xp = (const float*)a.x + (size_t)m * a.C + k0;
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    for (int s0 = 0; s0 < ksw; s0 += 16) {
        v4i wf[16], xf[16];
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const bool in = s0 + s < ksw && k0 + (s0 + s) * 16 < a.C;      // the activation row is C floats; weights are zero padded
            wf[s] = in ? *(const v4i*)(wp + (s0 + s) * 16) : v4i{0, 0, 0, 0};
            xf[s] = in ? *(const v4i*)(xp + (s0 + s) * 16) : v4i{0, 0, 0, 0};
        }
#pragma unroll
        for (int s = 0; s < 16; ++s) acc = mma_step(wf[s], xf[s], acc);
    }
    if (wave > 0) redf[wave - 1][lane] = acc;
    __syncthreads();
    if (wave > 0) return;
    acc += redf[0][lane];
    acc += redf[1][lane];
    acc += redf[2][lane];
    const int kb = n0 + fq * 4;
    if (frow >= a.M || kb >= a.K) return;
    float out[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float d = acc[r];
        if (a.bias && kb + r < a.K) d = __fadd_rn(d, a.bias[kb + r]);
        if (a.relu) d = d > 0.f ? d : (a.neg_slope == 0.f ? 0.f : __fmul_rn(d, a.neg_slope));
        out[r] = d;
    }
    float* y = (float*)a.y + (size_t)frow * a.K + kb;
    if (kb + 4 <= a.K && (a.K & 3) == 0) {
        *(float4*)y = make_float4(out[0], out[1], out[2], out[3]);
    } else {
        for (int r = 0; r < 4; ++r)
            if (kb + r < a.K) y[r] = out[r];
    }
}
bool fc_f32_small_ok(int m, int c, int kg_pad) { return m >= 1 && m <= 16 && c % 4 == 0 && c <= 65536 && kg_pad >= (c + 63) / 64 * 64; }
hipError_t launch_fc_f32_small(const ConvKArgs& a, hipStream_t s) {
    if (!fc_f32_small_ok(a.M, a.C, a.Kg_pad)) return hipErrorInvalidValue;
    const int ksw = (a.C + 63) / 64;       // 16-float k-steps per wave
    hipLaunchKernelGGL(fc_f32_small_kernel, dim3((a.K + 15) / 16), dim3(256), 0, s, a, ksw);
    return hipGetLastError();
}

// a.M = batch rows (<= 16), a.C = reduction length (multiple of 16), a.K = outputs, a.Kg_pad = weight row pitch
bool fc_i8_small_ok(int m, int c, int kg_pad) { return m >= 1 && m <= 16 && c % 16 == 0 && c <= 4 * 16 * 64 && kg_pad >= ((c + 255) / 256) * 256; }

hipError_t launch_fc_i8_small(const ConvKArgs& a, hipStream_t s) {
    if (!fc_i8_small_ok(a.M, a.C, a.Kg_pad)) return hipErrorInvalidValue;
    const int ksw = (a.C + 255) / 256;     // k-steps per wave
    dim3 grid((a.K + 15) / 16), block(256);
#define SABER_FC_CASE(n) case n: hipLaunchKernelGGL((fc_i8_small_kernel<n>), grid, block, 0, s, a); break;
    switch (ksw) {
        SABER_FC_CASE(1) SABER_FC_CASE(2) SABER_FC_CASE(3) SABER_FC_CASE(4) SABER_FC_CASE(5) SABER_FC_CASE(6)
        SABER_FC_CASE(7) SABER_FC_CASE(8) SABER_FC_CASE(9) SABER_FC_CASE(10) SABER_FC_CASE(11) SABER_FC_CASE(12)
        SABER_FC_CASE(13) SABER_FC_CASE(14) SABER_FC_CASE(15) SABER_FC_CASE(16)
    default: return hipErrorInvalidValue;
    }
#undef SABER_FC_CASE
    return hipGetLastError();
}

}  // namespace saber_mi355x
