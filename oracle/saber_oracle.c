/* oracle/saber_oracle.c — TEST INFRASTRUCTURE ONLY. NOT PART OF THE PRODUCT PATH.
 *
 * A plain-C CPU restatement of the arithmetic of Anakin's x86 Saber fused-operator hot path
 * (the path SURVEY.md §8 scopes). Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library, and only as the CHECKER for the HIP kernels.
 *
 * Parity status: PINNED for every function marked [pinned] below — each is checked bit-for-bit
 * against the reference's own unmodified sources compiled into oracle/_ref/libanakin_x86_ref.so
 * (tests/test_oracle_vs_ref.py) and against the committed golden vectors in tests/golden/
 * (generated FROM that compiled reference by tests/golden/make_golden.py). Functions marked
 * [unpinned] restate reference code that cannot be compiled here (xbyak JIT, MKL-packed paths);
 * they follow the cited lines but have no reference-produced vectors behind them.
 *
 * All float arithmetic is IEEE binary32, round-to-nearest-even, NO fused multiply-add unless the
 * reference uses one (compile with -ffp-contract=off; see oracle/Makefile).
 *
 * dtype codes (same as include/saber_hip.h): 0 = f32, 1 = s8, 2 = u8.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_F32 0
#define ORC_S8 1
#define ORC_U8 2

static inline float u_factor(int dtype) { return dtype == ORC_U8 ? (127.f / 255.f) : 1.f; }

static inline int conv_out_dim(int in, int pad, int k, int dil, int stride) {
    /* saber/funcs/funcs_utils.h:29-53 (conv_compute_shape): floor division */
    return (in + 2 * pad - (dil * (k - 1) + 1)) / stride + 1;
}

static inline int8_t sat_s8_from_float(float v) {
    /* saber/funcs/saber_util.h:513-527 saturate<int8_t>(float): clamp in the float domain, then cast */
    if (v < -128.f) v = -128.f;
    if (v > 127.f) v = 127.f;
    return (int8_t)v;
}
static inline uint8_t sat_u8_from_float(float v) {
    if (v < 0.f) v = 0.f;
    if (v > 255.f) v = 255.f;
    return (uint8_t)v;
}

/* ---- weight quantisation -------------------------------------------------------------------- */

/* [pinned] ScaleUtils::get_tensor_scale(axis=0): scale[oc] = max|w[oc,:]| / 127
 * (saber/funcs/impl/x86/x86_utils.h:141-166). inner = C*kh*kw. */
void orc_weight_scales(const float* w, int K, int inner, float* scale) {
    for (int k = 0; k < K; ++k) {
        float max_val = -1e20f;
        for (int i = 0; i < inner; ++i) {
            float a = fabsf(w[(size_t)k * inner + i]);
            max_val = a > max_val ? a : max_val;
        }
        scale[k] = max_val / 127.f;
    }
}

/* [pinned] ScaleUtils::scale_conv_weights_to_nchw_host / scale_fc_weights_to_nchw_host:
 * q = static_cast<char>(w / scale[oc]) — C cast, i.e. TRUNCATION toward zero
 * (x86_utils.h:293-322, :188-208). */
void orc_quant_weights(const float* w, int K, int inner, const float* scale, int8_t* q) {
    for (int k = 0; k < K; ++k) {
        for (int i = 0; i < inner; ++i) {
            q[(size_t)k * inner + i] = (int8_t)(w[(size_t)k * inner + i] / scale[k]);
        }
    }
}

/* ---- INT8 convolution ----------------------------------------------------------------------- */

/* [pinned] GemmX8S8S32XConv::create: per-out-channel pre-scaled bias and requantisation scale.
 * bias_p[oc] = bias[oc] * (1.f / (w_scale[oc] * in_scale * u_in))      gemm_x8s8s32x_conv.cpp:90-105
 *                                                                       + x86_utils.h:1053-1070
 * scale[oc]  = (w_scale[oc] * in_scale * u_in) / (out_scale * u_out)    gemm_x8s8s32x_conv.cpp:145-182
 * (u = 127/255 on a u8 side, absent on an s8 side; no division when the output is f32; the
 * u8->u8 case is written (a*u)/(b*u) in the reference and is kept in that form). */
void orc_conv_i8_prepare(int K, const float* w_scale, const float* bias, float in_scale,
                         float out_scale, int in_dtype, int out_dtype, float* bias_p, float* scale) {
    for (int k = 0; k < K; ++k) {
        float s_in;
        if (in_dtype == ORC_U8) {
            s_in = w_scale[k] * in_scale * (127.f / 255.f);
        } else {
            s_in = w_scale[k] * in_scale;
        }
        if (bias_p) {
            bias_p[k] = bias ? bias[k] * (1.f / s_in) : 0.f;
        }
        if (out_dtype == ORC_F32) {
            scale[k] = s_in;
        } else if (out_dtype == ORC_U8) {
            scale[k] = s_in / (out_scale * (127.f / 255.f));
        } else {
            scale[k] = s_in / out_scale;
        }
    }
}

/* Residual / fused-sum modes of the INT8 conv epilogue. */
#define ORC_RES_NONE 0
/* [unpinned] JIT `with_sum` post-op (jit_avx512_core_x8s8s32x_conv_kernel.cpp:19-39,156-177):
 * relu BEFORE the sum is skipped; d = (sum_scale == 1) ? d + prev : fmaf(prev, sum_scale, d);
 * relu after the sum when with_relu or the output is u8. prev is read from `out` (in place). */
#define ORC_RES_JIT_SUM 1
/* Exact two-op equivalent of  conv(->s8, no relu)  followed by  SaberEltwise<X86,AK_INT8> sum(+relu)
 * (the unfused INT8 graph, framework/graph/graph.cpp:423-436): q = sat8(rne(d)) as the conv would
 * have stored, then  t = c0*(float)q*s0 ; t += c1*(float)res*s1 ; relu ; sat8(roundf(t))
 * (saber_eltwise.cpp:98-111). Output is s8. */
#define ORC_RES_ELTWISE 2

typedef struct {
    int mode;          /* ORC_RES_* */
    int with_relu;     /* relu of the eltwise (mode 2) */
    float sum_scale;   /* mode 1 */
    int res_dtype;     /* mode 1: dtype of prev; mode 2: must be s8 */
    float coeff_conv;  /* mode 2: c0 */
    float coeff_res;   /* mode 2: c1 */
    float scale_conv;  /* mode 2: s0 = the conv output tensor's scale */
    float scale_res;   /* mode 2: s1 */
} orc_residual_t;

/* [pinned for mode 0] GemmX8S8S32XConv::sub_dispatch (gemm_x8s8s32x_conv.cpp:187-288).
 *   acc = sum_{kh,kw,ic} x*w in int32, zero padding. (For s8 inputs the reference shifts the
 *   activations by +128, pads with 128 and adds the -128*sum(w) offset inside the integer GEMM,
 *   :124-133,:459-485,:488-570 — the shifts cancel exactly in int32, so acc is the plain sum.)
 *   d = (float)acc; d += bias_p[oc]; d *= scale[oc]; if (relu && d < 0) d = 0;
 *   out = f32 ? d : saturate(nearbyintf(d))
 * The reference's GEMM path casts `(OutputDtype)nearbyintf(d)` without saturating (:278, undefined
 * for out-of-range d); the JIT path saturates (vpmovsdb / vpmovusdb, ..._conv_kernel.cpp:206-212).
 * This oracle pins SATURATE; the two agree whenever d is in range, which the parity tests ensure
 * when comparing with oracle/_ref.
 * x: NHWC [N,H,W,C] (s8 or u8). wq: OIHW [K, C/group, kh, kw] s8. out: NHWC [N,OH,OW,K].
 * res: optional residual tensor NHWC [N,OH,OW,K] (mode 2), or NULL. */
int orc_conv_i8(int N, int H, int W, int C, int K, int kh, int kw, int pad_h, int pad_w,
                int stride_h, int stride_w, int dil_h, int dil_w, int group, int in_dtype,
                int out_dtype, int with_relu, const void* x, const int8_t* wq, const float* bias_p,
                const float* scale, const orc_residual_t* rp, const void* res, void* out) {
    const int OH = conv_out_dim(H, pad_h, kh, dil_h, stride_h);
    const int OW = conv_out_dim(W, pad_w, kw, dil_w, stride_w);
    const int Cg = C / group, Kg = K / group;
    if (OH <= 0 || OW <= 0 || C % group || K % group) {
        return -1;
    }
    /* repack weights to [K][kh][kw][Cg] so the inner reduction is contiguous in both operands */
    int8_t* wr = (int8_t*)malloc((size_t)K * kh * kw * Cg);
    if (!wr) {
        return -2;
    }
    for (int k = 0; k < K; ++k)
        for (int c = 0; c < Cg; ++c)
            for (int i = 0; i < kh; ++i)
                for (int j = 0; j < kw; ++j)
                    wr[(((size_t)k * kh + i) * kw + j) * Cg + c] =
                        wq[(((size_t)k * Cg + c) * kh + i) * kw + j];
    const int8_t* xs = (const int8_t*)x;
    const uint8_t* xu = (const uint8_t*)x;
    const int mode = rp ? rp->mode : ORC_RES_NONE;
#pragma omp parallel for collapse(3) schedule(static)
    for (int n = 0; n < N; ++n) {
        for (int oh = 0; oh < OH; ++oh) {
            for (int ow = 0; ow < OW; ++ow) {
                for (int k = 0; k < K; ++k) {
                    const int g = k / Kg;
                    int32_t acc = 0;
                    for (int i = 0; i < kh; ++i) {
                        const int ih = oh * stride_h - pad_h + i * dil_h;
                        if (ih < 0 || ih >= H) continue;
                        for (int j = 0; j < kw; ++j) {
                            const int iw = ow * stride_w - pad_w + j * dil_w;
                            if (iw < 0 || iw >= W) continue;
                            const size_t xo = (((size_t)n * H + ih) * W + iw) * C + (size_t)g * Cg;
                            const int8_t* wp = wr + (((size_t)k * kh + i) * kw + j) * Cg;
                            int32_t s = 0;
                            if (in_dtype == ORC_U8) {
                                for (int c = 0; c < Cg; ++c) s += (int32_t)xu[xo + c] * (int32_t)wp[c];
                            } else {
                                for (int c = 0; c < Cg; ++c) s += (int32_t)xs[xo + c] * (int32_t)wp[c];
                            }
                            acc += s;
                        }
                    }
                    const size_t oo = (((size_t)n * OH + oh) * OW + ow) * K + k;
                    float d = (float)acc;
                    if (bias_p) {
                        d += bias_p[k];
                    }
                    d *= scale[k];
                    if (mode == ORC_RES_JIT_SUM) {
                        float prev;
                        if (rp->res_dtype == ORC_F32) prev = ((const float*)out)[oo];
                        else if (rp->res_dtype == ORC_U8) prev = (float)((const uint8_t*)out)[oo];
                        else prev = (float)((const int8_t*)out)[oo];
                        if (rp->sum_scale == 1.f) d = d + prev;
                        else d = fmaf(prev, rp->sum_scale, d);
                        if (with_relu || out_dtype == ORC_U8) d = d > 0.f ? d : 0.f;
                    } else if (with_relu && d < 0) {
                        d = 0;
                    }
                    if (mode == ORC_RES_ELTWISE) {
                        const int8_t q = sat_s8_from_float(nearbyintf(d));
                        float t = rp->coeff_conv * (float)q * rp->scale_conv;
                        t += rp->coeff_res * (float)((const int8_t*)res)[oo] * rp->scale_res;
                        if (rp->with_relu) t = t > 0 ? t : 0;
                        ((int8_t*)out)[oo] = sat_s8_from_float(roundf(t));
                    } else if (out_dtype == ORC_F32) {
                        ((float*)out)[oo] = d;
                    } else if (out_dtype == ORC_U8) {
                        ((uint8_t*)out)[oo] = sat_u8_from_float(nearbyintf(d));
                    } else {
                        ((int8_t*)out)[oo] = sat_s8_from_float(nearbyintf(d));
                    }
                }
            }
        }
    }
    free(wr);
    return 0;
}

/* Raw int32 accumulators of the same convolution (no epilogue); used by property tests. */
int orc_conv_i8_acc(int N, int H, int W, int C, int K, int kh, int kw, int pad_h, int pad_w,
                    int stride_h, int stride_w, int dil_h, int dil_w, int group, int in_dtype,
                    const void* x, const int8_t* wq, int32_t* acc_out) {
    const int OH = conv_out_dim(H, pad_h, kh, dil_h, stride_h);
    const int OW = conv_out_dim(W, pad_w, kw, dil_w, stride_w);
    const int Cg = C / group, Kg = K / group;
    const int8_t* xs = (const int8_t*)x;
    const uint8_t* xu = (const uint8_t*)x;
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; ++n)
        for (int oh = 0; oh < OH; ++oh)
            for (int ow = 0; ow < OW; ++ow)
                for (int k = 0; k < K; ++k) {
                    const int g = k / Kg;
                    int32_t acc = 0;
                    for (int c = 0; c < Cg; ++c)
                        for (int i = 0; i < kh; ++i) {
                            const int ih = oh * stride_h - pad_h + i * dil_h;
                            if (ih < 0 || ih >= H) continue;
                            for (int j = 0; j < kw; ++j) {
                                const int iw = ow * stride_w - pad_w + j * dil_w;
                                if (iw < 0 || iw >= W) continue;
                                const size_t xo = (((size_t)n * H + ih) * W + iw) * C + (size_t)g * Cg + c;
                                const int32_t xv = in_dtype == ORC_U8 ? (int32_t)xu[xo] : (int32_t)xs[xo];
                                acc += xv * (int32_t)wq[(((size_t)k * Cg + c) * kh + i) * kw + j];
                            }
                        }
                    acc_out[(((size_t)n * OH + oh) * OW + ow) * K + k] = acc;
                }
    return 0;
}

/* ---- FP32 convolution ----------------------------------------------------------------------- */

/* [pinned] conv_basic_check<float> (test/saber/conv_func_helper.h:196-264), the reference's own
 * naive FP32 oracle: NCHW, dst = dst*beta, accumulate in (ic,kh,kw) order in float, then *alpha,
 * +bias, relu. The x86 production paths (MKL sgemm / JIT) differ from this only by summation
 * order, hence FP32 parity is tolerance-based (1e-4 relative, BASELINE.json). */
int orc_conv_f32_nchw(int N, int C, int H, int W, int K, int kh, int kw, int pad_h, int pad_w,
                      int stride_h, int stride_w, int dil_h, int dil_w, int group, const float* x,
                      const float* w, const float* bias, int with_relu, float alpha, float beta,
                      float* out) {
    const int OH = conv_out_dim(H, pad_h, kh, dil_h, stride_h);
    const int OW = conv_out_dim(W, pad_w, kw, dil_w, stride_w);
    const int Cg = C / group, Kg = K / group;
    if (OH <= 0 || OW <= 0) return -1;
#pragma omp parallel for collapse(3) schedule(static)
    for (int n = 0; n < N; ++n)
        for (int k = 0; k < K; ++k)
            for (int oh = 0; oh < OH; ++oh)
                for (int ow = 0; ow < OW; ++ow) {
                    const int g = k / Kg;
                    const size_t oo = (((size_t)n * K + k) * OH + oh) * OW + ow;
                    float d = out[oo] * beta;
                    for (int c = 0; c < Cg; ++c)
                        for (int i = 0; i < kh; ++i)
                            for (int j = 0; j < kw; ++j) {
                                const int iw = ow * stride_w - pad_w + j * dil_w;
                                const int ih = oh * stride_h - pad_h + i * dil_h;
                                if (iw < 0 || iw >= W) continue;
                                if (ih < 0 || ih >= H) continue;
                                d += x[(((size_t)n * C + (size_t)g * Cg + c) * H + ih) * W + iw] *
                                     w[(((size_t)k * Cg + c) * kh + i) * kw + j];
                            }
                    d *= alpha;
                    d += bias ? bias[k] : 0.f;
                    if (with_relu) d = d > 0.f ? d : 0.f;
                    out[oo] = d;
                }
    return 0;
}

/* ---- quantise / dequantise + layout --------------------------------------------------------- */

/* [pinned] reorder_nhwc_nchw, NCHW f32 -> NHWC s8/u8 (saber/funcs/saber_util.h:759-797):
 *   s8: saturate<int8_t>(roundf(x * (1.f/scale)))          (round half AWAY from zero)
 *   u8: saturate<uint8_t>(roundf(x * (1.f/(scale*127/255)))) */
void orc_quant_nchw_to_nhwc(int N, int C, int H, int W, int out_dtype, float scale, const float* x,
                            void* out) {
    const float inv = out_dtype == ORC_U8 ? 1.f / (scale * (127.f / 255.f)) : 1.f / scale;
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; ++n)
        for (int h = 0; h < H; ++h)
            for (int w = 0; w < W; ++w)
                for (int c = 0; c < C; ++c) {
                    const float v = roundf(x[(((size_t)n * C + c) * H + h) * W + w] * inv);
                    const size_t oo = (((size_t)n * H + h) * W + w) * C + c;
                    if (out_dtype == ORC_U8) ((uint8_t*)out)[oo] = sat_u8_from_float(v);
                    else ((int8_t*)out)[oo] = sat_s8_from_float(v);
                }
}

/* [pinned] reorder_nhwc_nchw, NHWC s8/u8 -> NCHW f32 (saber_util.h:646-683):
 *   s8: q * scale ;  u8: (float)q * (scale * (127/255)) */
void orc_dequant_nhwc_to_nchw(int N, int C, int H, int W, int in_dtype, float scale, const void* x,
                              float* out) {
    const float s = in_dtype == ORC_U8 ? scale * (127.f / 255.f) : scale;
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; ++n)
        for (int c = 0; c < C; ++c)
            for (int h = 0; h < H; ++h)
                for (int w = 0; w < W; ++w) {
                    const size_t io = (((size_t)n * H + h) * W + w) * C + c;
                    const float q = in_dtype == ORC_U8 ? (float)((const uint8_t*)x)[io]
                                                       : (float)((const int8_t*)x)[io];
                    out[(((size_t)n * C + c) * H + h) * W + w] = q * s;
                }
}

/* [pinned through the INT8 fc: tests/test_oracle_vs_ref.py::test_fc_i8_matches_reference_packed_gemm]
 * ScaleUtils::scale_fp32_int8 (x86_utils.h:325-346; same layout, flat):
 * secur_cast2char(x * (1.f/scale)) = clamp((int)roundf(v), -128, 127). Used by the INT8 FC when
 * handed an f32 input (mkl_packed_int8_gemm.cpp:52-57). */
void orc_quant_flat_s8(size_t n, float scale, const float* x, int8_t* out) {
    const float inv = 1.f / scale;
    for (size_t i = 0; i < n; ++i) {
        float t = roundf(x[i] * inv);
        int ti = (int)t;
        ti = ti > 127 ? 127 : ti;
        ti = ti < -128 ? -128 : ti;
        out[i] = (int8_t)ti;
    }
}

/* ---- eltwise -------------------------------------------------------------------------------- */

/* [pinned] SaberEltwise<X86, AK_INT8>::simple_sum (saber_eltwise.cpp:71-113), two inputs:
 *   t = c0*(float)a*sa;  t += c1*(float)b*sb;  relu;  saturate<int8_t>(roundf(t))
 * The output tensor's scale is NOT applied (declared and unused at :85). */
void orc_eltwise_i8(size_t n, const int8_t* a, const int8_t* b, float sa, float sb, float c0,
                    float c1, int with_relu, int8_t* out) {
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) {
        float t = c0 * (float)a[i] * sa;
        t += c1 * (float)b[i] * sb;
        if (with_relu) t = t > 0 ? t : 0;
        out[i] = sat_s8_from_float(roundf(t));
    }
}

/* [pinned] SaberEltwise<X86, AK_FLOAT>::simple_sum (saber_eltwise.cpp:40-70). */
void orc_eltwise_f32(size_t n, const float* a, const float* b, float c0, float c1, int with_relu,
                     float* out) {
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) {
        float t = c0 * a[i];
        t += c1 * b[i];
        out[i] = with_relu ? (t > 0 ? t : 0) : t;
    }
}

/* ---- pooling -------------------------------------------------------------------------------- */

/* Pooling<>::compute_output_shape (saber/funcs/pooling.h:69-130): ceil unless
 * cmp_out_shape_floor_as_conv; with padding, drop a window that starts beyond the padded input. */
int orc_pool_out_dim(int in, int pad, int win, int stride, int floor_mode) {
    int o;
    if (floor_mode) {
        o = (int)((float)(in + 2 * pad - win) / stride) + 1;
        if (o <= 0) o = 1;
    } else {
        o = (int)ceilf((float)(in + 2 * pad - win) / stride) + 1;
    }
    if (pad > 0 && (o - 1) * stride >= in + pad) --o;
    return o;
}

/* [unpinned: JIT] SaberPooling<X86, AK_INT8> (saber_pooling.cpp:589-654 +
 * kernel/jit_avx512_core_8bit_pooling_kernel.cpp:166-287), NHWC s8/u8 in:
 *   max: element-wise max over the in-bounds window (signed or unsigned compare by src dtype);
 *   avg: s = int32 sum over the in-bounds window; f = (float)s * idivider, idivider =
 *        1.0f / (exclude_padding ? valid_h*valid_w : kh*kw); out = f32 ? f : sat(rne(f)).
 * type: 0 max, 1 avg include padding, 2 avg exclude padding. The output keeps the INPUT scale
 * (saber_pooling.cpp:583-584). Agrees with pool_basic_check_int8 (conv_func_helper.h:29-100)
 * except where sum/count and sum*(1/count) round differently (tests/test_oracle_vs_ref.py). */
int orc_pool_i8_nhwc(int N, int H, int W, int C, int OH, int OW, int kh, int kw, int stride_h,
                     int stride_w, int pad_h, int pad_w, int type, int in_dtype, int out_dtype,
                     const void* x, void* out) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; ++n)
        for (int oh = 0; oh < OH; ++oh)
            for (int ow = 0; ow < OW; ++ow) {
                int hs = oh * stride_h - pad_h, ws = ow * stride_w - pad_w;
                int he = hs + kh, we = ws + kw;
                if (hs < 0) hs = 0;
                if (ws < 0) ws = 0;
                if (he > H) he = H;
                if (we > W) we = W;
                const float idiv = 1.0f / (float)(type == 2 ? (he - hs) * (we - ws) : kh * kw);
                for (int c = 0; c < C; ++c) {
                    int32_t s = 0, m = in_dtype == ORC_U8 ? 0 : -128;
                    for (int ih = hs; ih < he; ++ih)
                        for (int iw = ws; iw < we; ++iw) {
                            const size_t io = (((size_t)n * H + ih) * W + iw) * C + c;
                            const int32_t v = in_dtype == ORC_U8 ? (int32_t)((const uint8_t*)x)[io]
                                                                 : (int32_t)((const int8_t*)x)[io];
                            s += v;
                            m = v > m ? v : m;
                        }
                    const size_t oo = (((size_t)n * OH + oh) * OW + ow) * C + c;
                    if (type == 0) {
                        if (out_dtype == ORC_U8) ((uint8_t*)out)[oo] = (uint8_t)m;
                        else ((int8_t*)out)[oo] = (int8_t)m;
                    } else {
                        const float f = (float)s * idiv;
                        if (out_dtype == ORC_F32) ((float*)out)[oo] = f;
                        else if (out_dtype == ORC_U8) ((uint8_t*)out)[oo] = sat_u8_from_float(nearbyintf(f));
                        else ((int8_t*)out)[oo] = sat_s8_from_float(nearbyintf(f));
                    }
                }
            }
    return 0;
}

/* [unpinned] SaberPooling<X86, AK_FLOAT> generic NCHW path (saber_pooling.cpp:385-500):
 * max over in-bounds window; avg = sequential sum / window area (include; clipped at in+pad on the
 * far edge) or / valid count (exclude). */
int orc_pool_f32_nchw(int N, int C, int H, int W, int OH, int OW, int kh, int kw, int stride_h,
                      int stride_w, int pad_h, int pad_w, int type, const float* x, float* out) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; ++n)
        for (int c = 0; c < C; ++c)
            for (int oh = 0; oh < OH; ++oh)
                for (int ow = 0; ow < OW; ++ow) {
                    int hs = oh * stride_h - pad_h, ws = ow * stride_w - pad_w;
                    int he = hs + kh, we = ws + kw;
                    if (hs < 0) hs = 0;
                    if (ws < 0) ws = 0;
                    if (he > H) he = H;
                    if (we > W) we = W;
                    const float* xp = x + ((size_t)n * C + c) * H * W;
                    /* saber_pooling.cpp:437-462: result = 0, the FIRST element of the window is assigned, the others are maxed / added - an EMPTY
                     * window (ceil-mode output whose window starts past the image: stride > window) therefore yields 0 for max pooling and 0 / 0 or
                     * 0 / negative for the averages. (Through round 5 the maximum started from xp[hs * W + ws] before the loop: the same for every
                     * non-empty window, the next row's first element - or a read past the tensor - for an empty one; found by the round-6 random
                     * sweep, where the GPU kernel answered 0 like the reference.) */
                    float r = 0.f;
                    int first = 1;
                    for (int ih = hs; ih < he; ++ih)
                        for (int iw = ws; iw < we; ++iw) {
                            const float v = xp[ih * W + iw];
                            if (first) { r = v; first = 0; }
                            else if (type == 0) r = r >= v ? r : v;
                            else r += v;
                        }
                    if (type == 1) { /* divisor clipped at in+pad on the far edge, saber_pooling.cpp:466-480 */
                        int bh = kh, bw = kw;
                        if (we == W) bw = (ws + kw >= W + pad_w ? W + pad_w : ws + kw) - ws;
                        if (he == H) bh = (hs + kh >= H + pad_h ? H + pad_h : hs + kh) - hs;
                        r /= (float)(bh * bw);
                    }
                    if (type == 2) r /= (float)((he - hs) * (we - ws));
                    out[(((size_t)n * C + c) * OH + oh) * OW + ow] = r;
                }
    return 0;
}

/* ---- GEMM / FC ------------------------------------------------------------------------------ */

/* [tolerance] Gemm<X86,...>: row-major C = alpha*op(A)*op(B) + beta*C (saber/funcs/gemm.h:27-66);
 * k-ordered float accumulation. */
void orc_gemm_f32(int trans_a, int trans_b, int M, int N, int Kd, float alpha, const float* A,
                  const float* B, float beta, float* Cm) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int m = 0; m < M; ++m)
        for (int n = 0; n < N; ++n) {
            float acc = 0.f;
            for (int k = 0; k < Kd; ++k) {
                const float a = trans_a ? A[(size_t)k * M + m] : A[(size_t)m * Kd + k];
                const float b = trans_b ? B[(size_t)n * Kd + k] : B[(size_t)k * N + n];
                acc += a * b;
            }
            const size_t o = (size_t)m * N + n;
            Cm[o] = beta == 0.f ? alpha * acc : alpha * acc + beta * Cm[o];
        }
}

/* [tolerance] VenderFc<X86, AK_FLOAT>::dispatch (vender_fc.cpp:154-212):
 * out[m,n] = sum_k in[m,k]*W[n,k] (+ bias[n] added afterwards by saxpy). W is [n,k] unless
 * is_transpose_weights. */
void orc_fc_f32(int M, int N, int Kd, const float* in, const float* Wt, int w_is_kn,
                const float* bias, float* out) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int m = 0; m < M; ++m)
        for (int n = 0; n < N; ++n) {
            float acc = 0.f;
            for (int k = 0; k < Kd; ++k)
                acc += in[(size_t)m * Kd + k] * (w_is_kn ? Wt[(size_t)k * N + n] : Wt[(size_t)n * Kd + k]);
            out[(size_t)m * N + n] = bias ? acc + bias[n] : acc;
        }
}

/* [pinned: bit-exact against the compiled PackedMKLInt8Gemm, tests/test_oracle_vs_ref.py + tests/golden/fc_i8_f32in.npz]
 * VenderFc<X86, AK_INT8> with s8 (or f32, quantised first by
 * orc_quant_flat_s8) input and f32 output = PackedMKLInt8Gemm::dispatch
 * (mkl_packed_int8_gemm.cpp:46-90; init :22-45):
 *   scale[n] = w_scale[n] * in_scale ;  out = (float)acc32 * scale[n] + bias[n]
 * (mul then add, two roundings). wq is [n,k] s8. */
void orc_fc_i8_s8in(int M, int N, int Kd, const int8_t* in, const int8_t* wq, const float* w_scale,
                    float in_scale, const float* bias, float* out) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int m = 0; m < M; ++m)
        for (int n = 0; n < N; ++n) {
            int32_t acc = 0;
            for (int k = 0; k < Kd; ++k)
                acc += (int32_t)in[(size_t)m * Kd + k] * (int32_t)wq[(size_t)n * Kd + k];
            const float sc = w_scale[n] * in_scale;
            float v = (float)acc * sc;
            if (bias) v = v + bias[n];
            out[(size_t)m * N + n] = v;
        }
}

/* [pinned: bit-exact against the compiled VenderFc<X86,AK_INT8>, tests/test_oracle_vs_ref.py]
 * VenderFc<X86, AK_INT8> with u8 input, f32 output (vender_fc.cpp:253-300,318-422):
 *   scale[n]  = (in_scale * w_scale[n]) / out_scale           (no 127/255 factor — reference quirk)
 *   bias_i[n] = (int)(bias[n] / scale[n])                     (truncation, x86_utils.h:276-291)
 *   out       = scale[n]==1 ? (float)(acc+bias_i) : scale[n] * (float)(acc + bias_i) */
void orc_fc_i8_u8in(int M, int N, int Kd, const uint8_t* in, const int8_t* wq, const float* w_scale,
                    float in_scale, float out_scale, const float* bias, float* out) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int m = 0; m < M; ++m)
        for (int n = 0; n < N; ++n) {
            int32_t acc = 0;
            for (int k = 0; k < Kd; ++k)
                acc += (int32_t)in[(size_t)m * Kd + k] * (int32_t)wq[(size_t)n * Kd + k];
            const float sc = (in_scale * w_scale[n]) / out_scale;
            if (bias) acc += (int32_t)(bias[n] / sc);
            out[(size_t)m * N + n] = sc == 1.f ? (float)acc : sc * (float)acc;
        }
}

/* ---- BN + Scale folding (cold path; defines the weights the hot path sees) ------------------- */

/* [pinned: bit-exact against the compiled framework/utils/parameter_fusion.cpp, tests/test_oracle_vs_ref.py]
 * WeightsFusion<float,T>::update_weights
 * (framework/utils/parameter_fusion.cpp:88-131), all in f32, in this order:
 *   s = bn_scale==0 ? 1 : 1/bn_scale; alpha = 1/sqrtf(var*s + eps); beta = -(mean*s)*alpha;
 *   alpha = scale_w*alpha; beta = beta*scale_w (+ scale_b);
 *   w[oc,:] *= alpha; bias[oc] = bias[oc]*alpha + beta */
void orc_bn_fold(int K, int inner, float* w, float* bias, int has_bias, float bn_scale, float eps,
                 const float* mean, const float* var, const float* scale_w, const float* scale_b) {
    const float s = bn_scale == 0.f ? 1.f : 1.f / bn_scale;
    for (int k = 0; k < K; ++k) {
        float alpha = 1.f / sqrtf(var[k] * s + eps);
        float beta = -1.f * (mean[k] * s) * alpha;
        alpha = scale_w[k] * alpha;
        beta = beta * scale_w[k];
        if (scale_b) beta = beta + scale_b[k];
        for (int i = 0; i < inner; ++i) w[(size_t)k * inner + i] *= alpha;
        const float b0 = has_bias ? bias[k] : 0.f;
        bias[k] = b0 * alpha + beta;
    }
}

/* ---- softmax -------------------------------------------------------------------------------- */

/* [tolerance] SaberSoftmax<X86, AK_FLOAT> along the channel axis of [outer, C, inner]
 * (saber/funcs/impl/x86/saber_softmax.cpp): max-subtract, expf, normalise. */
void orc_softmax_f32(int outer, int Cn, int inner, const float* x, float* out) {
    for (int o = 0; o < outer; ++o)
        for (int i = 0; i < inner; ++i) {
            const float* xp = x + (size_t)o * Cn * inner + i;
            float* op = out + (size_t)o * Cn * inner + i;
            float m = xp[0];
            for (int c = 1; c < Cn; ++c) m = xp[(size_t)c * inner] > m ? xp[(size_t)c * inner] : m;
            float s = 0.f;
            for (int c = 0; c < Cn; ++c) {
                op[(size_t)c * inner] = expf(xp[(size_t)c * inner] - m);
                s += op[(size_t)c * inner];
            }
            for (int c = 0; c < Cn; ++c) op[(size_t)c * inner] /= s;
        }
}

/* ---- identities the HIP kernels rely on ----------------------------------------------------- */

/* roundf(x) == truncf(x + copysignf(0x1.fffffep-2f, x)) for every float with |x| < 2^23; returns the
 * number of mismatches over ALL such floats (both signs). Used to justify the 3-instruction roundf of the
 * device epilogues. */
long orc_check_round_identity(void) {
    long bad = 0;
    const float h = 0x1.fffffep-2f;
#pragma omp parallel for reduction(+ : bad) schedule(static)
    for (uint32_t b = 0; b < 0x4B000000u; ++b) {
        float x;
        memcpy(&x, &b, 4);
        if (truncf(x + copysignf(h, x)) != roundf(x)) ++bad;
        x = -x;
        if (truncf(x + copysignf(h, x)) != roundf(x)) ++bad;
    }
    return bad;
}
