// anakin_amd/csrc/epilogue_pack.h - the INT8 requantisation epilogues on 4 consecutive channels of one pixel -> 4 packed
// bytes, shared by the kernels that keep a lane's accumulator rows in registers (conv1x1_chain.hip, stage_xcd.hip).
// Same float operation sequence as epilogue_i8_fast / epilogue_i8_pair (conv_igemm_impl.h): the bits of the separate ops.
#pragma once
#include "conv_igemm_impl.h"

namespace saber_mi355x {

// the fused-eltwise epilogue of epilogue_i8_fast<NV, EK_ELT> (conv_igemm_impl.h) on 4 channels -> 4 packed s8
template <class ARGS>   // ARGS: coeff_conv, scale_conv, coeff_res, scale_res
__device__ __forceinline__ unsigned chain_elt_pack(const v4i acc, const v4i comp, const v4f bias, const v4f scale,
                                                   unsigned rs, float lo_s8, float res_lo, const ARGS& a) {
    float dq[4];
#pragma unroll
    for (int r = 0; r < 4; r += 2) {
        v2f d2 = {(float)(acc[r] + comp[r]), (float)(acc[r + 1] + comp[r + 1])};
        d2 = d2 + v2f{bias[r], bias[r + 1]};
        d2 = d2 * v2f{scale[r], scale[r + 1]};
        dq[r] = d2.x;
        dq[r + 1] = d2.y;
    }
    // the eltwise on PAIRS of values with packed f32 multiplies / adds (IEEE per component, no contraction: the bits of the
    // scalar sequence, ~3 VALU instructions fewer per output)
    unsigned w = 0;
    const v2f cc = {a.coeff_conv, a.coeff_conv}, sc2 = {a.scale_conv, a.scale_conv};
    const v2f cr = {a.coeff_res, a.coeff_res}, sr2 = {a.scale_res, a.scale_res};
#pragma unroll
    for (int t = 0; t < 4; t += 2) {
        v2f q = {__builtin_amdgcn_fmed3f(rintf(dq[t]), lo_s8, 127.f), __builtin_amdgcn_fmed3f(rintf(dq[t + 1]), lo_s8, 127.f)};
        v2f rv = {(float)(int)(int8_t)(rs >> (8 * t)), (float)(int)(int8_t)(rs >> (8 * t + 8))};
        v2f e = (cc * q) * sc2;
        e = e + (cr * rv) * sr2;
        e.x = fmaxf(e.x, res_lo);
        e.y = fmaxf(e.y, res_lo);
        // round_half_away(e) + 128: trunc(e + copysign(0.49999997, e)) + 128
        v2f h = {copysignf(0x1.fffffep-2f, e.x), copysignf(0x1.fffffep-2f, e.y)};
        v2f r = e + h;
        r.x = truncf(r.x);
        r.y = truncf(r.y);
        r = r + v2f{128.f, 128.f};
        w = __builtin_amdgcn_cvt_pk_u8_f32(r.x, t, w);
        w = __builtin_amdgcn_cvt_pk_u8_f32(r.y, t + 1, w);
    }
    return w ^ 0x80808080u;
}

// the s8 / u8 epilogue of epilogue_i8_pair (run-time output type) on 4 channels
__device__ __forceinline__ unsigned chain_out_pack(const v4i acc, const v4i comp, const v4f bias, const v4f scale,
                                                   float lo, float off, unsigned xm) {
    float dq[4];
#pragma unroll
    for (int r = 0; r < 4; r += 2) {
        v2f d2 = {(float)(acc[r] + comp[r]), (float)(acc[r + 1] + comp[r + 1])};
        d2 = d2 + v2f{bias[r], bias[r + 1]};
        d2 = d2 * v2f{scale[r], scale[r + 1]};
        dq[r] = d2.x;
        dq[r + 1] = d2.y;
    }
    unsigned w = 0;
    if (xm == 0u) {   // u8: the saturating convert clamps at 0 itself (the relu, if any, is implied) - no max, no offset
#pragma unroll
        for (int t = 0; t < 4; ++t) w = __builtin_amdgcn_cvt_pk_u8_f32(rintf(dq[t]), t, w);
        return w;
    }
#pragma unroll
    for (int t = 0; t < 4; t += 2) {
        v2f q = {fmaxf(rintf(dq[t]), lo), fmaxf(rintf(dq[t + 1]), lo)};
        q = q + v2f{off, off};
        w = __builtin_amdgcn_cvt_pk_u8_f32(q.x, t, w);
        w = __builtin_amdgcn_cvt_pk_u8_f32(q.y, t + 1, w);
    }
    return w ^ xm;
}

__device__ __forceinline__ void lds_dma16(const void* src, void* dst_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)dst_wave_base, 16, 0, 0);
}


}  // namespace saber_mi355x
