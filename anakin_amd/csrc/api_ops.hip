// anakin_amd/csrc/api_ops.hip - fully connected, INT8 / FP32 GEMM and the streaming operators of include/saber_hip.h.
#include "api_internal.h"

// ================================================================================================
// fully connected: a 1x1 convolution over a [m,1,1,k] tensor with the FC epilogues
// ================================================================================================
int saber_hip_fc_create(const saber_hip_fc_desc* desc, saber_hip_fc_t** out) {
    if (!desc || !out) return fail(SABER_HIP_INVALID_VALUE, "null argument");
    saber_hip_conv_desc c;
    std::memset(&c, 0, sizeof c);
    c.n = desc->m; c.h = 1; c.w = 1; c.c = desc->k; c.k = desc->n; c.kh = c.kw = 1;
    c.stride_h = c.stride_w = c.dil_h = c.dil_w = c.group = 1;
    c.in_layout = c.out_layout = SABER_HIP_NHWC;
    c.out_dtype = SABER_HIP_F32;
    c.int8_weights = desc->int8_weights;
    auto* fc = new saber_hip_fc();
    fc->d = *desc;
    if (desc->int8_weights) {
        if (desc->in_dtype == SABER_HIP_F32) {
            fc->pre_quant = true;  // PackedMKLInt8Gemm::dispatch: scale_fp32_int8 (mkl_packed_int8_gemm.cpp:52-57)
            c.in_dtype = SABER_HIP_S8;
        } else {
            c.in_dtype = desc->in_dtype;
        }
    } else {
        c.in_dtype = SABER_HIP_F32;
    }
    int rc = saber_hip_conv2d_create(&c, &fc->conv);
    if (rc) {
        delete fc;
        return rc;
    }
    if (desc->int8_weights) {
        if (fc->conv->algo != ALGO_IGEMM_I8) {
            saber_hip_conv2d_destroy(fc->conv);
            delete fc;
            return fail(SABER_HIP_UNIMPL, "INT8 fc needs k % 16 == 0");
        }
        fc->conv->epi = c.in_dtype == SABER_HIP_U8 ? EPI_I8_FC_U8 : EPI_I8_FC_S8;
        if (fc_small_ok(fc->conv)) {   // STATIC choice for inference batches (<= 16 rows): the weight-streaming kernel
            fc->conv->fc_small = 1;
            name_algo(fc->conv);
        }
    } else if (fc_small_ok(fc->conv)) {   // FP32: likewise
        fc->conv->fc_small = 1;
        name_algo(fc->conv);
    }
    *out = fc;
    return SABER_HIP_OK;
}

int saber_hip_fc_set_weights(saber_hip_fc_t* fc, const void* w, int w_dtype, const float* w_scale,
                             const float* bias, float in_scale, float out_scale) {
    const int N = fc->d.n, K = fc->d.k;
    fc->in_scale = in_scale;
    // bring the weights to [n,k]
    std::vector<uint8_t> wt;
    const void* wnk = w;
    const size_t es = w_dtype == SABER_HIP_F32 ? 4 : 1;
    if (fc->d.w_is_kn) {
        wt.resize((size_t)N * K * es);
        for (int n = 0; n < N; ++n)
            for (int k = 0; k < K; ++k)
                std::memcpy(&wt[((size_t)n * K + k) * es], (const uint8_t*)w + ((size_t)k * N + n) * es, es);
        wnk = wt.data();
    }
    saber_hip_conv* op = fc->conv;
    if (!fc->d.int8_weights) return saber_hip_conv2d_set_weights(op, wnk, w_dtype, w_scale, bias, 1.f, 1.f);
    // INT8: the conv-style set_weights gives us quantised weights + comp; then override scale/bias
    int rc = saber_hip_conv2d_set_weights(op, wnk, w_dtype, w_scale, nullptr, in_scale, out_scale);
    if (rc) return rc;
    const int K_pad = round_up(N, 128);
    std::vector<float> scale(K_pad, 0.f), b(K_pad, 0.f);
    if (op->epi == EPI_I8_FC_S8) {
        // _scale[n] = w_scale[n] * scale_a; out = acc*scale + bias   (mkl_packed_int8_gemm.cpp:36-38,78-81)
        for (int n = 0; n < N; ++n) {
            scale[n] = op->w_scale[n] * in_scale;
            if (bias) b[n] = bias[n];
        }
        op->has_bias = bias != nullptr;
        HIP_TRY(op->d_bias.upload(b));
        HIP_TRY(op->d_scale.upload(scale));
    } else {
        // u8 input (vender_fc.cpp:284-300): scale = (in_scale*w_scale)/out_scale; bias_i = (int)(bias/scale)
        std::vector<int> comp(K_pad, 0);
        const int8_t* q = op->wq_oihw.data();
        for (int n = 0; n < N; ++n) {
            scale[n] = (in_scale * op->w_scale[n]) / out_scale;
            int s = 0;
            for (int k = 0; k < K; ++k) s += (int)q[(size_t)n * K + k];
            comp[n] = 128 * s + (bias ? (int)(bias[n] / scale[n]) : 0);
        }
        op->has_bias = false;
        op->has_comp = true;
        HIP_TRY(op->d_comp.upload(comp));
        HIP_TRY(op->d_scale.upload(scale));
    }
    return SABER_HIP_OK;
}

size_t saber_hip_fc_workspace_bytes(const saber_hip_fc_t* fc) {
    return fc->pre_quant ? (size_t)fc->d.m * fc->d.k : 0;
}

int saber_hip_fc_run(saber_hip_fc_t* fc, const void* x, float* y, void* workspace, saber_hip_stream_t stream) {
    if (g_capture) return capture_fc(fc, x, y, false);
    const void* xin = x;
    if (fc->pre_quant) {
        if (!workspace) return fail(SABER_HIP_INVALID_VALUE, "workspace required");
        HIP_TRY(launch_quantize_flat_s8((size_t)fc->d.m * fc->d.k, fc->in_scale, (const float*)x, (int8_t*)workspace,
                                        (hipStream_t)stream));
        xin = workspace;
    }
    return saber_hip_conv2d_run(fc->conv, xin, y, nullptr, nullptr, stream);
}

int saber_hip_fc_run_q(saber_hip_fc_t* fc, const int8_t* xq, float* y, saber_hip_stream_t stream) {
    if (!fc || !xq || !y) return fail(SABER_HIP_INVALID_VALUE, "null argument");
    if (!fc->d.int8_weights || fc->conv->x_dtype != DT_S8) return fail(SABER_HIP_INVALID_VALUE, "fc_run_q: INT8 fc with s8 operand only");
    if (g_capture) return capture_fc(fc, xq, y, true);
    return saber_hip_conv2d_run(fc->conv, xq, y, nullptr, nullptr, stream);
}

// fc + Softmax in ONE launch (fc_small.hip: the last-arriving workgroup normalises the rows). Eligible: an INT8 fc whose small-batch
// weight-streaming kernel is selected, reading an 8-bit operand as it lies (no quantise-on-entry pre-pass), <= 1024 outputs, a
// reduction of 512 / 1024 / 2048 / 4096. Anything else - also a later change of the fc's kernel selection - runs the two launches:
// the entry point is always correct, the fusion is an optimisation.
bool fc_softmax_ok(const saber_hip_fc* fc, bool quantised_input) {
    if (!fc || !fc->conv || (fc->pre_quant && !quantised_input)) return false;
    const saber_hip_conv* c = fc->conv;
    if (c->fc_small && c->algo == ALGO_IGEMM_F32)      // FP32: the split-K kernel's last tile normalises the rows (fc_f32_splitk.hip)
        return c->d_fcpart.p && !c->d_wfc.p && fc_f32_splitk_ok(c->d.n, c->c_eff, c->Kg_pad, c->d.k, true);
    const int ksw = (c->c_eff + 255) / 256;
    return c->fc_small && c->algo == ALGO_IGEMM_I8 && (ksw == 2 || ksw == 4 || ksw == 8 || ksw == 16) &&
           fc_i8_small_softmax_ok(c->d.n, c->c_eff, c->Kg_pad, c->d.k);
}
int fc_softmax_prepare(saber_hip_fc* fc) {
    if (!fc || !fc->conv) return fail(SABER_HIP_INVALID_VALUE, "null argument");
    if (fc->conv->algo == ALGO_IGEMM_F32) return SABER_HIP_OK;      // (the FP32 kernel's counters come with its partial-sum scratch: set_weights)
    if (!fc->conv->d_sm_ctr.p) HIP_TRY(fc->conv->d_sm_ctr.alloc_zero(32));
    return SABER_HIP_OK;
}
// quantised_input: x is the s8 operand an f32-input INT8 fc would have computed on entry (saber_hip_fc_run_q's contract)
int fc_run_softmax(saber_hip_fc* fc, const void* x, float* y, float* prob, void* workspace, hipStream_t s, bool quantised_input) {
    if (!fc || !x || !y || !prob) return fail(SABER_HIP_INVALID_VALUE, "null argument");
    if (!g_capture && fc_softmax_ok(fc, quantised_input) && !fc->conv->d_sm_ctr.p) {
        // first use outside a net: the counter (nothing may be allocated under stream capture)
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &cs) == hipSuccess && cs == hipStreamCaptureStatusNone) (void)fc_softmax_prepare(fc);
    }
    if (!g_capture && fc->conv->algo == ALGO_IGEMM_F32 && fc_softmax_ok(fc, quantised_input)) {
        saber_hip_conv* op = fc->conv;
        if (!op->weights_set) return fail(SABER_HIP_INVALID_VALUE, "set_weights not called");
        ConvKArgs a;
        conv_fill_args(op, a, x, y, nullptr);
        HIP_TRY(launch_fc_f32_splitk(a, op->d_fcpart.p, op->d_fcctr.p, prob, s));
        return SABER_HIP_OK;
    }
    if (!g_capture && fc_softmax_ok(fc, quantised_input) && fc->conv->d_sm_ctr.p) {
        saber_hip_conv* op = fc->conv;
        if (!op->weights_set) return fail(SABER_HIP_INVALID_VALUE, "set_weights not called");
        ConvKArgs a;
        conv_fill_args(op, a, x, y, nullptr);
        HIP_TRY(launch_fc_i8_small_softmax(a, prob, op->d_sm_ctr.p, s));
        return SABER_HIP_OK;
    }
    // the two operators one after the other (under an op-list capture: recorded as the two operators they are)
    const int rc = quantised_input ? saber_hip_fc_run_q(fc, (const int8_t*)x, y, (saber_hip_stream_t)s)
                                   : saber_hip_fc_run(fc, x, y, workspace, (saber_hip_stream_t)s);
    return rc ? rc : saber_hip_softmax_f32(fc->d.m, fc->d.n, y, prob, (saber_hip_stream_t)s);
}
int saber_hip_fc_run_softmax(saber_hip_fc_t* fc, const void* x, float* y, float* prob, void* workspace, saber_hip_stream_t stream) {
    return fc_run_softmax(fc, x, y, prob, workspace, (hipStream_t)stream, false);
}

const char* saber_hip_fc_algo(const saber_hip_fc_t* fc) { return fc ? fc->conv->algo_name.c_str() : ""; }
int saber_hip_fc_set_tile(saber_hip_fc_t* fc, int tile) {
    if (!fc) return fail(SABER_HIP_INVALID_VALUE, "null argument");
    return saber_hip_conv2d_set_tile(fc->conv, tile);
}
void saber_hip_fc_destroy(saber_hip_fc_t* fc) {
    if (fc) saber_hip_conv2d_destroy(fc->conv);
    delete fc;
}

// ================================================================================================
// INT8 GEMM: C[m,n] (int32) = op(A)[m,k] (s8|u8) x op(B)[k,n] (s8), exact; B packed at create time
// (MklDnnGemm<int8_t|uint8_t, int8_t, int> in PACKED_MKLGEMM mode, saber/funcs/impl/x86/mkl_gemm.cpp:138-256).
// Runs on the implicit-GEMM kernel as a 1x1 convolution over [m,1,1,k] with the raw-accumulator epilogue.
// ================================================================================================
struct saber_hip_gemm_i8 {
    int trans_a = 0, m = 0, n = 0, k = 0, k_pad = 0;
    saber_hip_conv* conv = nullptr;
};

int saber_hip_gemm_i8_create(int trans_a, int trans_b, int m, int n, int k, int a_dtype, const int8_t* b_host,
                             saber_hip_gemm_i8_t** out) {
    if (!out || !b_host) return fail(SABER_HIP_INVALID_VALUE, "null argument");
    if (m <= 0 || n <= 0 || k <= 0) return fail(SABER_HIP_INVALID_VALUE, "bad gemm shape");
    if (a_dtype != SABER_HIP_S8 && a_dtype != SABER_HIP_U8) return fail(SABER_HIP_INVALID_VALUE, "A must be s8 or u8");
    auto* g = new saber_hip_gemm_i8();
    g->trans_a = trans_a ? 1 : 0; g->m = m; g->n = n; g->k = k; g->k_pad = round_up(k, 16);
    saber_hip_conv_desc c;
    std::memset(&c, 0, sizeof c);
    c.n = m; c.h = 1; c.w = 1; c.c = g->k_pad; c.k = n; c.kh = c.kw = 1;
    c.stride_h = c.stride_w = c.dil_h = c.dil_w = c.group = 1;
    c.in_layout = c.out_layout = SABER_HIP_NHWC;
    c.in_dtype = a_dtype;
    c.out_dtype = SABER_HIP_F32;       // 4-byte outputs: the raw epilogue stores int32 bit patterns
    c.int8_weights = 1;
    int rc = saber_hip_conv2d_create(&c, &g->conv);
    if (rc) { delete g; return rc; }
    // op(B)[k,n] -> weight rows [n][k_pad]
    std::vector<int8_t> w((size_t)n * g->k_pad, 0);
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < k; ++i) w[(size_t)j * g->k_pad + i] = trans_b ? b_host[(size_t)j * k + i] : b_host[(size_t)i * n + j];
    std::vector<float> ones(n, 1.f);
    rc = saber_hip_conv2d_set_weights(g->conv, w.data(), SABER_HIP_S8, ones.data(), nullptr, 1.f, 1.f);
    if (rc) { saber_hip_conv2d_destroy(g->conv); delete g; return rc; }
    g->conv->epi = EPI_I8_RAW_S32;
    *out = g;
    return SABER_HIP_OK;
}
size_t saber_hip_gemm_i8_workspace_bytes(const saber_hip_gemm_i8_t* g) {
    return (g->trans_a || g->k_pad != g->k) ? (size_t)g->m * g->k_pad : 0;
}
int saber_hip_gemm_i8_run(saber_hip_gemm_i8_t* g, const void* a, int32_t* c, void* workspace, saber_hip_stream_t stream) {
    if (g_capture) return capture_unsupported("saber_hip_gemm_i8_run");
    if (!g || !a || !c) return fail(SABER_HIP_INVALID_VALUE, "null argument");
    const void* ain = a;
    if (saber_hip_gemm_i8_workspace_bytes(g)) {
        if (!workspace) return fail(SABER_HIP_INVALID_VALUE, "workspace required");
        if (g->trans_a) HIP_TRY(launch_transpose_bytes(g->k, g->m, g->k_pad, a, workspace, (hipStream_t)stream));   // A is [k][m]
        else HIP_TRY(launch_pad_channels_i8((size_t)g->m, g->k, g->k_pad, a, workspace, (hipStream_t)stream));
        ain = workspace;
    }
    return saber_hip_conv2d_run(g->conv, ain, c, nullptr, nullptr, stream);
}
void saber_hip_gemm_i8_destroy(saber_hip_gemm_i8_t* g) {
    if (g) saber_hip_conv2d_destroy(g->conv);
    delete g;
}

// ================================================================================================
// thin wrappers
// ================================================================================================
// (saber_hip_gemm_f32: api_gemm.hip)
int saber_hip_quantize_nchw_to_nhwc(int n, int c, int h, int w, int c_pad, int out_dtype, float scale,
                                    const float* x, void* y, saber_hip_stream_t s) {
    if (c_pad < c || (out_dtype != SABER_HIP_S8 && out_dtype != SABER_HIP_U8))
        return fail(SABER_HIP_INVALID_VALUE, "bad quantize arguments");
    if ((size_t)n * c * h * w == 0) return SABER_HIP_OK;
    if (g_capture) {
        const int p[6] = {n, c, h, w, c_pad, out_dtype};
        return capture_stream_op(OP_QUANT, "quantize_nchw_to_nhwc", p, 6, &scale, 1, 0, x, (size_t)n * c * h * w * 4, nullptr, 0, y,
                                 (size_t)n * h * w * c_pad, nullptr, 0);
    }
    HIP_TRY(launch_quantize_nchw_to_nhwc(n, c, h, w, c_pad, out_dtype, scale, x, y, (hipStream_t)s));
    return SABER_HIP_OK;
}
int saber_hip_dequantize_nhwc_to_nchw(int n, int c, int h, int w, int in_dtype, float scale, const void* x,
                                      float* y, saber_hip_stream_t s) {
    if (in_dtype != SABER_HIP_S8 && in_dtype != SABER_HIP_U8) return fail(SABER_HIP_INVALID_VALUE, "bad dtype");
    if ((size_t)n * c * h * w == 0) return SABER_HIP_OK;
    if (g_capture) {
        const int p[5] = {n, c, h, w, in_dtype};
        return capture_stream_op(OP_DEQUANT, "dequantize_nhwc_to_nchw", p, 5, &scale, 1, 0, x, (size_t)n * c * h * w, nullptr, 0, y,
                                 (size_t)n * c * h * w * 4, nullptr, 0);
    }
    HIP_TRY(launch_dequantize_nhwc_to_nchw(n, c, h, w, in_dtype, scale, x, y, (hipStream_t)s));
    return SABER_HIP_OK;
}
int saber_hip_transpose_nchw_to_nhwc_f32(int n, int c, int h, int w, int c_pad, const float* x, float* y,
                                         saber_hip_stream_t s) {
    if (g_capture) {
        const int p[5] = {n, c, h, w, c_pad};
        return capture_stream_op(OP_TRANSPOSE_IN, "transpose_nchw_to_nhwc_f32", p, 5, nullptr, 0, 0, x, (size_t)n * c * h * w * 4, nullptr, 0,
                                 y, (size_t)n * h * w * c_pad * 4, nullptr, 0);
    }
    HIP_TRY(launch_transpose_nchw_to_nhwc_f32(n, c, h, w, c_pad, x, y, (hipStream_t)s));
    return SABER_HIP_OK;
}
int saber_hip_transpose_nhwc_to_nchw_f32(int n, int c, int h, int w, int c_pad, const float* x, float* y,
                                         saber_hip_stream_t s) {
    if (g_capture) return capture_unsupported("saber_hip_transpose_nhwc_to_nchw_f32");
    HIP_TRY(launch_transpose_nhwc_to_nchw_f32(n, c, h, w, c_pad, x, y, (hipStream_t)s));
    return SABER_HIP_OK;
}
int saber_hip_quantize_flat_s8(size_t count, float scale, const float* x, int8_t* y, saber_hip_stream_t s) {
    if (!count) return SABER_HIP_OK;
    if (g_capture) return capture_unsupported("saber_hip_quantize_flat_s8");
    HIP_TRY(launch_quantize_flat_s8(count, scale, x, y, (hipStream_t)s));
    return SABER_HIP_OK;
}
int saber_hip_eltwise_sum_i8(size_t count, const int8_t* a, const int8_t* b, float sa, float sb, float c0, float c1,
                             int relu, int8_t* y, saber_hip_stream_t s) {
    if (!count) return SABER_HIP_OK;
    if (g_capture) {
        const float f[4] = {sa, sb, c0, c1};
        return capture_stream_op(OP_ELT_I8, "eltwise_sum_i8", &relu, 1, f, 4, count, a, count, b, count, y, count, nullptr, 0);
    }
    HIP_TRY(launch_eltwise_sum_i8(count, a, b, sa, sb, c0, c1, relu, y, (hipStream_t)s));
    return SABER_HIP_OK;
}
int saber_hip_eltwise_sum_f32(size_t count, const float* a, const float* b, float c0, float c1, int relu, float* y,
                              saber_hip_stream_t s) {
    if (!count) return SABER_HIP_OK;
    if (g_capture) {
        const float f[2] = {c0, c1};
        return capture_stream_op(OP_ELT_F32, "eltwise_sum_f32", &relu, 1, f, 2, count, a, count * 4, b, count * 4, y, count * 4, nullptr, 0);
    }
    HIP_TRY(launch_eltwise_sum_f32(count, a, b, c0, c1, relu, y, (hipStream_t)s));
    return SABER_HIP_OK;
}
int saber_hip_relu_f32(size_t count, const float* x, float* y, saber_hip_stream_t s) {
    if (!count) return SABER_HIP_OK;
    if (g_capture) return capture_stream_op(OP_RELU_F32, "relu_f32", nullptr, 0, nullptr, 0, count, x, count * 4, nullptr, 0, y, count * 4, nullptr, 0);
    HIP_TRY(launch_relu_f32(count, x, y, (hipStream_t)s));
    return SABER_HIP_OK;
}
static bool activation_kind_ok(int active) { return active == 1 || active == 3 || active == 4 || active == 5 || active == 9 || active == 11 || active == 12; }
int saber_hip_activation_f32(int active, size_t count, float negative_slope, float coef, const float* x, float* y, saber_hip_stream_t s) {
    if (active == 2) {      // Active_relu: the standalone operator ignores negative_slope (saber_activation.cpp:136-154)
        return saber_hip_relu_f32(count, x, y, s);
    }
    if (!activation_kind_ok(active)) return fail(SABER_HIP_UNIMPL, "activation: sigmoid 1, relu 2, tanh 3, clipped relu 4, elu 5, stanh 9, gelu 11, swish 12 (prelu: saber_hip_prelu_f32)");
    if (!count) return SABER_HIP_OK;
    if (!x || !y) return fail(SABER_HIP_INVALID_VALUE, "null argument");
    if (g_capture) {
        const float f[2] = {negative_slope, coef};
        return capture_stream_op(OP_ACT_F32, "activation_f32", &active, 1, f, 2, count, x, count * 4, nullptr, 0, y, count * 4, nullptr, 0);
    }
    HIP_TRY(launch_activation_f32(active, count, negative_slope, coef, x, y, (hipStream_t)s));
    return SABER_HIP_OK;
}
int saber_hip_prelu_f32(size_t count, int channels, int inner, int channel_shared, const float* slope, const float* x, float* y,
                        saber_hip_stream_t s) {
    if (g_capture) return capture_unsupported("saber_hip_prelu_f32 (its slope tensor has no op-list form)");
    if (!count) return SABER_HIP_OK;
    if (!slope || !x || !y || channels <= 0 || inner <= 0) return fail(SABER_HIP_INVALID_VALUE, "prelu: bad argument");
    HIP_TRY(launch_prelu_f32(count, channels, inner, channel_shared, slope, x, y, (hipStream_t)s));
    return SABER_HIP_OK;
}
int saber_hip_pool_out_dim2(int in, int pad, int window, int stride, int floor_mode, int any_pad) {
    int o;  // Pooling<>::compute_output_shape, saber/funcs/pooling.h:92-121
    if (floor_mode) {
        o = (int)((float)(in + 2 * pad - window) / stride) + 1;
        if (o <= 0) o = 1;
    } else {
        o = (int)ceilf((float)(in + 2 * pad - window) / stride) + 1;
    }
    // the reference applies the clip to BOTH dimensions whenever pooling_padded(), i.e. pad_h || pad_w
    // (pooling.h:113-120, saber_funcs_param.h:2141), not per dimension
    if (any_pad && (o - 1) * stride >= in + pad) --o;
    return o;
}
int saber_hip_pool_out_dim(int in, int pad, int window, int stride, int floor_mode) {
    return saber_hip_pool_out_dim2(in, pad, window, stride, floor_mode, pad > 0);
}
int saber_hip_pool2d_i8_nhwc(int n, int h, int w, int c, int oh, int ow, int kh, int kw, int sh, int sw, int ph,
                             int pw, int type, int in_dtype, int out_dtype, const void* x, void* y,
                             saber_hip_stream_t s) {
    if (type == SABER_HIP_POOL_MAX && out_dtype == SABER_HIP_F32)
        return fail(SABER_HIP_UNIMPL, "dst format (AK_FLOAT) and pooling type (Pooling_max): NOT supported");
    if (g_capture) {
        const int p[15] = {n, h, w, c, oh, ow, kh, kw, sh, sw, ph, pw, type, in_dtype, out_dtype};
        return capture_stream_op(OP_POOL_I8, "pool2d_i8_nhwc", p, 15, nullptr, 0, 0, x, (size_t)n * h * w * c, nullptr, 0, y,
                                 (size_t)n * oh * ow * c * (out_dtype == SABER_HIP_F32 ? 4 : 1), nullptr, 0);
    }
    HIP_TRY(launch_pool2d_i8_nhwc(n, h, w, c, oh, ow, kh, kw, sh, sw, ph, pw, type, in_dtype, out_dtype, x, y,
                                  (hipStream_t)s));
    return SABER_HIP_OK;
}
int saber_hip_pool2d_f32(int n, int h, int w, int c, int oh, int ow, int kh, int kw, int sh, int sw, int ph, int pw,
                         int type, int layout, const float* x, float* y, saber_hip_stream_t s) {
    if (g_capture) {
        const int p[14] = {n, h, w, c, oh, ow, kh, kw, sh, sw, ph, pw, type, layout};
        return capture_stream_op(OP_POOL_F32, "pool2d_f32", p, 14, nullptr, 0, 0, x, (size_t)n * h * w * c * 4, nullptr, 0, y,
                                 (size_t)n * oh * ow * c * 4, nullptr, 0);
    }
    HIP_TRY(launch_pool2d_f32(n, h, w, c, oh, ow, kh, kw, sh, sw, ph, pw, type, layout == SABER_HIP_NCHW, x, y,
                              (hipStream_t)s));
    return SABER_HIP_OK;
}
int saber_hip_pool2d_f32_from_i8_q(int n, int h, int w, int c, int oh, int ow, int kh, int kw, int sh, int sw, int ph,
                                   int pw, int type, int in_dtype, float scale, const void* x, float* y, float q_scale,
                                   int8_t* yq, saber_hip_stream_t s) {
    if (in_dtype != SABER_HIP_S8 && in_dtype != SABER_HIP_U8) return fail(SABER_HIP_INVALID_VALUE, "bad dtype");
    if (yq && !(q_scale > 0.f)) return fail(SABER_HIP_INVALID_VALUE, "bad quantisation scale");
    if (g_capture) {
        const int p[14] = {n, h, w, c, oh, ow, kh, kw, sh, sw, ph, pw, type, in_dtype};
        const float f[2] = {scale, q_scale};
        return capture_stream_op(OP_POOL_F32_I8, yq ? "pool2d_f32_from_i8+quantize" : "pool2d_f32_from_i8", p, 14, f, 2, 0, x,
                                 (size_t)n * h * w * c, nullptr, 0, y, (size_t)n * oh * ow * c * 4, yq, (size_t)n * oh * ow * c);
    }
    HIP_TRY(launch_pool2d_f32_from_i8(n, h, w, c, oh, ow, kh, kw, sh, sw, ph, pw, type, in_dtype, scale, x, y, q_scale,
                                      yq, (hipStream_t)s));
    return SABER_HIP_OK;
}
int saber_hip_pool2d_f32_from_i8(int n, int h, int w, int c, int oh, int ow, int kh, int kw, int sh, int sw, int ph,
                                 int pw, int type, int in_dtype, float scale, const void* x, float* y,
                                 saber_hip_stream_t s) {
    return saber_hip_pool2d_f32_from_i8_q(n, h, w, c, oh, ow, kh, kw, sh, sw, ph, pw, type, in_dtype, scale, x, y, 1.f,
                                          nullptr, s);
}
int saber_hip_softmax_f32(int rows, int cols, const float* x, float* y, saber_hip_stream_t s) {
    if (rows <= 0 || cols <= 0) return fail(SABER_HIP_INVALID_VALUE, "bad softmax shape");
    if (g_capture) {
        const int p[2] = {rows, cols};
        return capture_stream_op(OP_SOFTMAX, "softmax_f32", p, 2, nullptr, 0, 0, x, (size_t)rows * cols * 4, nullptr, 0, y, (size_t)rows * cols * 4,
                                 nullptr, 0);
    }
    HIP_TRY(launch_softmax_f32(rows, cols, x, y, (hipStream_t)s));
    return SABER_HIP_OK;
}

