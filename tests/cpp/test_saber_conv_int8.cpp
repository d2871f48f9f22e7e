// tests/cpp/test_saber_conv_int8.cpp — C++ parity test in the style of the reference's
// test/saber/test_saber_conv_int8.cpp:39-234 and test_saber_conv_eltwise_int8.cpp:36-215 (sweep over
// kernel / pad / stride / dilation / bias / relu / batch / channels, init -> dispatch through the Saber
// interface), but with an INTEGER oracle and a BIT-EXACT pass criterion instead of the reference's
// rel-L2 < 0.15 against FP32: every output byte must equal oracle/saber_oracle.c (the restatement pinned to
// the compiled reference). Inputs are seeded (the reference's are not).
// Built by __graft_entry__.build(); run on the GPU by tests/test_gpu_cpp.py.
#include <cstdint>
#include <cstdio>
#include <random>
#include <vector>

#include "saber_mi355x.hpp"

using namespace anakin::saber;

extern "C" {   // oracle/libsaber_oracle.so (test infrastructure)
void orc_weight_scales(const float* w, int K, int inner, float* scale);
void orc_quant_weights(const float* w, int K, int inner, const float* scale, int8_t* q);
void orc_conv_i8_prepare(int K, const float* w_scale, const float* bias, float in_scale, float out_scale, int in_dtype,
                         int out_dtype, float* bias_p, float* scale);
typedef struct { int mode, with_relu; float sum_scale; int res_dtype; float coeff_conv, coeff_res, scale_conv, scale_res; } orc_residual_t;
int orc_conv_i8(int N, int H, int W, int C, int K, int kh, int kw, int pad_h, int pad_w, int stride_h, int stride_w,
                int dil_h, int dil_w, int group, int in_dtype, int out_dtype, int with_relu, const void* x,
                const int8_t* wq, const float* bias_p, const float* scale, const orc_residual_t* rp, const void* res,
                void* out);
void orc_eltwise_i8(size_t n, const int8_t* a, const int8_t* b, float sa, float sb, float c0, float c1, int with_relu,
                    int8_t* out);
int orc_pool_out_dim(int in, int pad, int window, int stride, int floor_mode);
int orc_pool_i8_nhwc(int N, int H, int W, int C, int OH, int OW, int kh, int kw, int sh, int sw, int ph, int pw, int type,
                      int in_dtype, int out_dtype, const void* x, void* out);
void orc_quant_nchw_to_nhwc(int N, int C, int H, int W, int out_dtype, float scale, const float* x, void* y);
}

static int g_fail = 0, g_run = 0;

static DataType ak(int code) { return code == 0 ? AK_FLOAT : (code == 1 ? AK_INT8 : AK_UINT8); }

// one case: conv (+ optional fused eltwise with a residual tensor)
static void test_conv_int8(int N, int C, int H, int W, int K, int k, int pad, int stride, int dil, bool bias_term,
                           bool relu, int in_dt /*1 s8, 2 u8*/, int out_dt /*0 f32, 1 s8, 2 u8*/, bool fuse_eltwise,
                           Context<MI355X>& ctx) {
    std::mt19937 rng(1234 + N * 7 + C * 13 + K * 17 + k * 19 + pad + stride * 3 + in_dt * 5 + out_dt * 11 + relu);
    const int OH = (H + 2 * pad - (dil * (k - 1) + 1)) / stride + 1, OW = (W + 2 * pad - (dil * (k - 1) + 1)) / stride + 1;
    std::vector<uint8_t> x((size_t)N * H * W * C);
    for (auto& v : x) v = in_dt == 2 ? (uint8_t)(rng() % 256) : (uint8_t)(int8_t)((int)(rng() % 256) - 128);
    std::vector<float> w((size_t)K * C * k * k), b(K);
    std::normal_distribution<float> nd(0.f, 0.3f);
    for (auto& v : w) v = nd(rng);
    for (auto& v : b) v = nd(rng);
    const float in_scale = 0.021f, out_scale = 0.37f, res_scale = 0.043f, elt_scale = 0.06f;

    // ---- oracle ------------------------------------------------------------------------------
    std::vector<float> ws(K), bp(K), sc(K);
    std::vector<int8_t> wq(w.size());
    orc_weight_scales(w.data(), K, C * k * k, ws.data());
    orc_quant_weights(w.data(), K, C * k * k, ws.data(), wq.data());
    orc_conv_i8_prepare(K, ws.data(), bias_term ? b.data() : nullptr, in_scale, out_scale, in_dt, out_dt, bp.data(), sc.data());
    const size_t on = (size_t)N * OH * OW * K;
    std::vector<uint8_t> want(on * (out_dt == 0 ? 4 : 1)), got(want.size());
    std::vector<int8_t> res(on);
    for (auto& v : res) v = (int8_t)((int)(rng() % 256) - 128);
    if (fuse_eltwise) {   // conv -> s8, then SaberEltwise<AK_INT8> sum + relu: two oracle ops
        std::vector<int8_t> mid(on);
        orc_conv_i8(N, H, W, C, K, k, k, pad, pad, stride, stride, dil, dil, 1, in_dt, 1, relu, x.data(), wq.data(),
                    bias_term ? bp.data() : nullptr, sc.data(), nullptr, nullptr, mid.data());
        orc_eltwise_i8(on, mid.data(), res.data(), out_scale, res_scale, 1.f / elt_scale, 1.f / elt_scale, 1, (int8_t*)want.data());
    } else {
        orc_conv_i8(N, H, W, C, K, k, k, pad, pad, stride, stride, dil, dil, 1, in_dt, out_dt, relu, x.data(), wq.data(),
                    bias_term ? bp.data() : nullptr, sc.data(), nullptr, nullptr, want.data());
    }

    // ---- device, through the Saber interface ----------------------------------------------------
    Tensor<MI355X> tin(Shape({N, H, W, C}, Layout_NHWC), ak(in_dt));
    Tensor<MI355X> tout(Shape({N, OH, OW, K}, Layout_NHWC), ak(out_dt));
    Tensor<MI355X> tres(Shape({N, OH, OW, K}, Layout_NHWC), AK_INT8);
    tin.set_scale({in_scale});
    tout.set_scale({out_scale});
    tres.set_scale({res_scale});
    tin.copy_from_host(x.data());
    tres.copy_from_host(res.data());
    HostBlob hw(Shape({K, C, k, k}), AK_FLOAT, w.data());
    HostBlob hb(Shape({1, K, 1, 1}), AK_FLOAT, b.data());
    ActivationParam<MI355X> act = relu ? ActivationParam<MI355X>(Active_relu) : ActivationParam<MI355X>();
    ConvParam<MI355X> cp(1, pad, pad, stride, stride, dil, dil, &hw, bias_term ? &hb : nullptr, act);
    std::vector<Tensor<MI355X>*> ins{&tin}, outs{&tout};
    SaberStatus st;
    const char* algo = "";
    if (fuse_eltwise) {
        EltwiseParam<MI355X> ep(Eltwise_sum, {1.f / elt_scale, 1.f / elt_scale}, ActivationParam<MI355X>(Active_relu));
        ConvEltwiseParam<MI355X> cep(cp, ep);
        ins.push_back(&tres);
        SaberConvEltwise<MI355X, AK_INT8> op;
        st = op.init(ins, outs, cep, ctx);
        if (st == SaberSuccess) st = op.dispatch(ins, outs, cep);
        (void)hipStreamSynchronize(ctx.get_compute_stream());
        algo = op.algo();
        tout.copy_to_host(got.data());
    } else {
        SaberConv2D<MI355X, AK_INT8> op;
        st = op.init(ins, outs, cp, ctx);
        if (st == SaberSuccess) st = op.dispatch(ins, outs, cp);
        (void)hipStreamSynchronize(ctx.get_compute_stream());
        algo = op.algo();
        tout.copy_to_host(got.data());
    }
    ++g_run;
    size_t diff = 0;
    if (st != SaberSuccess) diff = want.size();
    else for (size_t i = 0; i < want.size(); ++i) diff += want[i] != got[i];
    if (diff) {
        ++g_fail;
        printf("FAIL N=%d C=%d HxW=%dx%d K=%d k=%d pad=%d stride=%d dil=%d bias=%d relu=%d in=%d out=%d fused=%d status=%d "
               "mismatching bytes=%zu/%zu [%s]\n", N, C, H, W, K, k, pad, stride, dil, bias_term, relu, in_dt, out_dt,
               fuse_eltwise, (int)st, diff, want.size(), algo);
    }
}

// SaberConv2DPooling: conv + pooling through the Saber interface == oracle conv followed by oracle pooling.
// stem = true: 7x7/2 conv over a 3-channel f32 NCHW image + 3x3/2 max pooling (fused kernel); else a 3x3 conv + 2x2/2
// max pooling (two launches behind the same interface).
static void test_conv_pooling(bool stem, Context<MI355X>& ctx) {
    std::mt19937 rng(stem ? 77 : 78);
    const int N = 2, C = stem ? 3 : 32, H = stem ? 75 : 14, W = stem ? 62 : 14, K = 64;
    const int k = stem ? 7 : 3, pad = stem ? 3 : 1, stride = stem ? 2 : 1;
    const int pk = stem ? 3 : 2, ps = 2;
    const int OH = (H + 2 * pad - k) / stride + 1, OW = (W + 2 * pad - k) / stride + 1;
    const int PH = orc_pool_out_dim(OH, 0, pk, ps, 0), PW = orc_pool_out_dim(OW, 0, pk, ps, 0);
    const float in_scale = stem ? 1.f / 127.f : 0.02f, out_scale = 0.05f;
    std::vector<float> w((size_t)K * C * k * k), b(K), xf((size_t)N * C * H * W);
    std::normal_distribution<float> nd(0.f, 0.1f);
    std::uniform_real_distribution<float> ud(-1.f, 1.f);
    for (auto& v : w) v = nd(rng);
    for (auto& v : b) v = nd(rng);
    for (auto& v : xf) v = ud(rng);
    std::vector<uint8_t> xq((size_t)N * H * W * C);
    const int in_dt = stem ? 1 : 2;
    if (stem) orc_quant_nchw_to_nhwc(N, C, H, W, 1, in_scale, xf.data(), xq.data());
    else for (auto& v : xq) v = (uint8_t)(rng() % 256);
    std::vector<float> ws(K), bp(K), sc(K);
    std::vector<int8_t> wq(w.size());
    orc_weight_scales(w.data(), K, C * k * k, ws.data());
    orc_quant_weights(w.data(), K, C * k * k, ws.data(), wq.data());
    orc_conv_i8_prepare(K, ws.data(), b.data(), in_scale, out_scale, in_dt, 2, bp.data(), sc.data());
    std::vector<uint8_t> conv_out((size_t)N * OH * OW * K), want((size_t)N * PH * PW * K), got(want.size());
    orc_conv_i8(N, H, W, C, K, k, k, pad, pad, stride, stride, 1, 1, 1, in_dt, 2, 1, xq.data(), wq.data(), bp.data(),
                sc.data(), nullptr, nullptr, conv_out.data());
    orc_pool_i8_nhwc(N, OH, OW, K, PH, PW, pk, pk, ps, ps, 0, 0, 0, 2, 2, conv_out.data(), want.data());

    Tensor<MI355X> tin(stem ? Shape({N, C, H, W}, Layout_NCHW) : Shape({N, H, W, C}, Layout_NHWC), stem ? AK_FLOAT : AK_UINT8);
    Tensor<MI355X> tout(Shape({N, PH, PW, K}, Layout_NHWC), AK_UINT8);
    tin.set_scale({in_scale});
    tout.set_scale({out_scale});
    tin.copy_from_host(stem ? (const void*)xf.data() : (const void*)xq.data());
    HostBlob hw(Shape({K, C, k, k}), AK_FLOAT, w.data());
    HostBlob hb(Shape({1, K, 1, 1}), AK_FLOAT, b.data());
    ConvParam<MI355X> cp(1, pad, pad, stride, stride, 1, 1, &hw, &hb, ActivationParam<MI355X>(Active_relu));
    PoolingParam<MI355X> pp(pk, pk, 0, 0, ps, ps, Pooling_max);
    ConvPoolingParam<MI355X> cpp(cp, pp);
    std::vector<Tensor<MI355X>*> ins{&tin}, outs{&tout};
    SaberConv2DPooling<MI355X, AK_INT8> op;
    SaberStatus st = op.init(ins, outs, cpp, ctx);
    if (st == SaberSuccess) st = op.dispatch(ins, outs, cpp);
    (void)hipStreamSynchronize(ctx.get_compute_stream());
    tout.copy_to_host(got.data());
    ++g_run;
    size_t diff = 0;
    if (st != SaberSuccess || op.fused() != stem) diff = want.size();
    else for (size_t i = 0; i < want.size(); ++i) diff += want[i] != got[i];
    if (diff) {
        ++g_fail;
        printf("FAIL conv+pooling stem=%d fused=%d status=%d mismatching bytes=%zu/%zu [%s]\n", (int)stem, (int)op.fused(),
               (int)st, diff, want.size(), op.algo());
    }
}

int main() {
    if (!saber_hip_device_ok()) {
        printf("no gfx950 device: the MI355X Saber target has no fallback\n");
        return 2;
    }
    Context<MI355X> ctx(0, 0, 0);
    // the reference sweep (test_saber_conv_int8.cpp:204-234): kernel {1,3}, pad {0,1}, stride {1,2}, bias, relu,
    // batch {1,3}, plus dilation and ResNet-like channel counts
    for (int k : {1, 3})
        for (int pad : {0, 1})
            for (int stride : {1, 2})
                for (int bias : {0, 1})
                    for (int relu : {0, 1})
                        for (int N : {1, 3})
                            for (int C : {16, 64})
                                for (int in_dt : {1, 2}) {
                                    const int out_dt = relu ? 2 : 1;   // x86 rule: conv+relu -> u8, else s8
                                    test_conv_int8(N, C, 13, 11, 32, k, pad, stride, 1, bias, relu, in_dt, out_dt, false, ctx);
                                }
    for (int dil : {1, 2}) test_conv_int8(2, 32, 12, 12, 48, 3, 2, 1, dil, true, true, 2, 2, false, ctx);
    test_conv_int8(1, 256, 14, 14, 64, 1, 0, 1, 1, true, true, 1, 2, false, ctx);      // res3 2a shape
    test_conv_int8(2, 128, 28, 28, 128, 3, 1, 1, 1, true, true, 2, 2, false, ctx);     // res3 2b shape
    test_conv_int8(2, 512, 7, 7, 512, 3, 1, 1, 1, true, true, 2, 0, false, ctx);       // f32 output
    test_conv_int8(2, 64, 56, 56, 256, 1, 0, 1, 1, true, false, 2, 1, true, ctx);      // res2 2c + fused eltwise
    test_conv_int8(1, 512, 7, 7, 2048, 1, 0, 1, 1, true, false, 2, 1, true, ctx);      // res5 2c + fused eltwise
    test_conv_pooling(true, ctx);     // SaberConv2DPooling: fused stem + max pooling
    test_conv_pooling(false, ctx);    // SaberConv2DPooling: two launches behind the same interface
    printf("%d/%d cases bit-exact\n", g_run - g_fail, g_run);
    return g_fail ? 1 : 0;
}
