// integration/test_saber_conv_mi355x.cpp — the MI355X target EXECUTED inside the reference's own operator stack.
//
// Compiled against the patched copy of the reference's Saber library (integration/apply_mi355x_target.py) — its
// unmodified Tensor / Shape / Env / Context / BaseFunc / Conv<> facade code — this drives, in the style of
// test/saber/test_saber_base.h:501-562 (TestSaberBase: fill host tensors, copy to the device target, init the op,
// run, copy back, compare):
//     Env<MI355X>::env_init -> Context<MI355X> -> Tensor<MI355X> (TargetWrapper<MI355X> on HIP)
//     Conv<MI355X, AK_INT8>::init(SPECIFY, SABER_IMPL)   -> BaseFunc::init -> SaberConv2D<MI355X,AK_INT8>::init
//     Conv::operator()                                   -> BaseFunc::operator() (saber/funcs/base.h:138-162)
//         -> SaberConv2D::dispatch -> integration/saber_mi355x_adaptor.h -> include/saber_hip.h -> HIP kernels
//     a second call with a DIFFERENT input shape          -> BaseFunc detects it and calls create() again (:151-161)
//     ConvEltwise-style INT8 conv + in-place sum (beta / beta_type -> sum_scale as the x86 impl derives it)
// and checks every output byte against oracle/saber_oracle.c (the restatement pinned to the compiled reference).
// Also exercises SaberTimer<MI355X> (hipEvents on the context's compute stream).
// Exit code 0 = all cases bit-exact. Run on the GPU by tests/test_gpu_cpp.py.
#include "anakin_config.h"
#include "saber/core/context.h"
#include "saber/core/tensor.h"
#include "saber/core/tensor_op.h"
#include "saber/funcs/conv.h"
#include "saber/funcs/conv_eltwise.h"
#include "saber/funcs/activation.h"
#include "saber/funcs/timer.h"

#include <cstdio>
#include <random>
#include <vector>

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
}

static int g_fail = 0, g_run = 0;
static int code(DataType t) { return t == AK_FLOAT ? 0 : (t == AK_INT8 ? 1 : 2); }

struct Layer {
    int C, K, k, pad, stride;
    bool relu;
    DataType in_dt, out_dt;
    std::vector<float> w, b, ws, bp, sc;
    std::vector<int8_t> wq;
    float in_scale, out_scale;
    Tensor<MI355X> dw, db;     // device copies (what a PBlock's d_tensor() is in the framework)
    Layer(int C_, int K_, int k_, int pad_, int stride_, bool relu_, DataType i, DataType o, unsigned seed)
        : C(C_), K(K_), k(k_), pad(pad_), stride(stride_), relu(relu_), in_dt(i), out_dt(o), in_scale(0.021f), out_scale(0.37f) {
        std::mt19937 rng(seed);
        std::normal_distribution<float> nd(0.f, 0.3f);
        w.resize((size_t)K * C * k * k); b.resize(K); ws.resize(K); bp.resize(K); sc.resize(K); wq.resize(w.size());
        for (auto& v : w) v = nd(rng);
        for (auto& v : b) v = nd(rng);
        orc_weight_scales(w.data(), K, C * k * k, ws.data());
        orc_quant_weights(w.data(), K, C * k * k, ws.data(), wq.data());
        orc_conv_i8_prepare(K, ws.data(), b.data(), in_scale, out_scale, code(in_dt), code(out_dt), bp.data(), sc.data());
        Tensor<X86> hw(Shape({K, C, k, k}, Layout_NCHW), AK_FLOAT), hb(Shape({1, K, 1, 1}, Layout_NCHW), AK_FLOAT);
        memcpy(hw.mutable_data(), w.data(), w.size() * 4);
        memcpy(hb.mutable_data(), b.data(), b.size() * 4);
        dw.re_alloc(hw.valid_shape(), AK_FLOAT);
        db.re_alloc(hb.valid_shape(), AK_FLOAT);
        dw.copy_from(hw);      // TargetWrapper<MI355X>::sync_memcpy(..., __HtoD)
        db.copy_from(hb);
    }
};

// one forward of `conv` on a batch of N HxW images; sum_prev: bytes already in the output for the in-place sum (or null)
static void run_and_check(Conv<MI355X, AK_INT8>& conv, ConvParam<MI355X>& param, Layer& L, int N, int H, int W, Context<MI355X>& ctx,
                          bool first, const char* what, DataType sum_dt = AK_INVALID, float sum_scale = 1.f) {
    std::mt19937 rng(77 + N * 3 + H * 5 + L.C);
    const int OH = (H + 2 * L.pad - L.k) / L.stride + 1, OW = (W + 2 * L.pad - L.k) / L.stride + 1;
    Tensor<X86> hx(Shape({N, H, W, L.C}, Layout_NHWC), L.in_dt);
    uint8_t* px = (uint8_t*)hx.mutable_data();
    for (size_t i = 0; i < (size_t)N * H * W * L.C; ++i) px[i] = (uint8_t)(rng() % 256);
    static Tensor<MI355X> dx, dy;          // the SAME tensor objects across calls: BaseFunc compares their shapes
    dx.re_alloc(hx.valid_shape(), L.in_dt);
    dx.set_scale({L.in_scale});
    dx.copy_from(hx);
    std::vector<Tensor<MI355X>*> ins{&dx}, outs{&dy};
    dy.set_dtype(L.out_dt);
    dy.set_layout(Layout_NHWC);
    conv.compute_output_shape(ins, outs, param);       // Conv<>::compute_output_shape (conv.h:73-79)
    dy.re_alloc(dy.valid_shape(), L.out_dt);
    dy.set_scale({L.out_scale});
    const size_t on = (size_t)N * OH * OW * L.K;
    std::vector<uint8_t> prev(on, 0);
    if (sum_dt != AK_INVALID) {
        Tensor<X86> hp(dy.valid_shape(), L.out_dt);
        for (auto& v : prev) v = (uint8_t)(rng() % 256);
        memcpy(hp.mutable_data(), prev.data(), on);
        dy.copy_from(hp);
    }
    SaberStatus st = SaberSuccess;
    if (first) st = conv.init(ins, outs, param, SPECIFY, SABER_IMPL, ctx);   // BaseFunc::init -> impl->init
    if (st == SaberSuccess) st = conv(ins, outs, param, ctx);                  // BaseFunc::operator(): shape change -> create
    outs[0]->record_event(ctx.get_compute_stream());
    outs[0]->sync();
    ++g_run;
    if (st != SaberSuccess) { printf("FAIL %s: status %d\n", what, (int)st); ++g_fail; return; }
    Tensor<X86> hy(dy.valid_shape(), L.out_dt);
    hy.copy_from(dy);                      // __DtoH
    std::vector<uint8_t> want(on);
    orc_residual_t rp = {0, 0, 1.f, 0, 1.f, 1.f, 1.f, 1.f};
    if (sum_dt != AK_INVALID) { rp.mode = 1; rp.with_relu = 1; rp.sum_scale = sum_scale; rp.res_dtype = code(sum_dt); memcpy(want.data(), prev.data(), on); }
    orc_conv_i8(N, H, W, L.C, L.K, L.k, L.k, L.pad, L.pad, L.stride, L.stride, 1, 1, 1, code(L.in_dt), code(L.out_dt), L.relu,
                hx.data(), L.wq.data(), L.bp.data(), L.sc.data(), sum_dt != AK_INVALID ? &rp : nullptr, nullptr, want.data());
    size_t bad = 0;
    const uint8_t* got = (const uint8_t*)hy.data();
    for (size_t i = 0; i < on; ++i) bad += got[i] != want[i];
    if (bad) { printf("FAIL %s N=%d %dx%d: %zu of %zu bytes differ\n", what, N, H, W, bad, on); ++g_fail; }
    else printf("ok   %s N=%d %dx%d -> %dx%dx%d bit-exact\n", what, N, H, W, OH, OW, L.K);
}

int main() {
    Env<MI355X>::env_init();               // Device<MI355X>: hipGetDeviceProperties + streams
    Context<MI355X> ctx(0, 0, 0);
    auto& dev = Env<MI355X>::cur_env()[0];
    printf("device %s (%s), %d CUs, %d MiB\n", dev._info._device_name.c_str(), dev._info._compute_ability.c_str(),
           dev._info._compute_core_num, dev._info._max_memory);
    struct Case { int C, K, k, pad, stride; bool relu; DataType i, o; } cases[] = {
        {64, 64, 3, 1, 1, true, AK_UINT8, AK_UINT8},      // res2 branch2b
        {256, 128, 1, 0, 2, true, AK_INT8, AK_UINT8},     // res3a branch2a (stride 2)
        {512, 256, 1, 0, 1, false, AK_UINT8, AK_INT8},    // branch2c (u8 -> s8)
        {32, 48, 3, 1, 2, false, AK_INT8, AK_INT8},
    };
    unsigned seed = 1;
    for (const Case& c : cases) {
        Layer L(c.C, c.K, c.k, c.pad, c.stride, c.relu, c.i, c.o, seed++);
        ActivationParam<MI355X> act = c.relu ? ActivationParam<MI355X>(Active_relu) : ActivationParam<MI355X>();
        ConvParam<MI355X> param(1, c.pad, c.pad, c.stride, c.stride, 1, 1, &L.dw, &L.db, act);
        Conv<MI355X, AK_INT8> conv;
        run_and_check(conv, param, L, 2, 14, 14, ctx, true, "Conv<MI355X,AK_INT8> init + operator()");
        run_and_check(conv, param, L, 3, 9, 11, ctx, false, "  same op, new input shape (BaseFunc re-creates)");
        run_and_check(conv, param, L, 3, 9, 11, ctx, false, "  same op, same shape (dispatch only)");
    }
    {   // ConvEltwise<MI355X,AK_INT8>: conv + relu'd in-place sum onto the output tensor, s8 bytes added into a u8 output.
        // The framework hands the ADDED tensor's scale as ConvParam::beta (+ beta_type); the impl derives
        // sum_scale = beta * (255/127) / out_scale as the x86 impl does (jit_avx512_core_x8s8s32x_conv.cpp:174-189).
        Layer L(128, 64, 1, 0, 1, true, AK_UINT8, AK_UINT8, 7);
        const float added_scale = 0.043f;
        ConvParam<MI355X> cp(1, 0, 0, 1, 1, 1, 1, &L.dw, &L.db, ActivationParam<MI355X>(Active_relu), 1.f, added_scale);
        cp.beta_type = AK_INT8;
        EltwiseParam<MI355X> ep(Eltwise_sum, {1.f, 1.f}, ActivationParam<MI355X>(Active_relu));
        ConvEltwiseParam<MI355X> param(cp, ep);
        const int N = 2, H = 14, W = 14;
        std::mt19937 rng(5);
        Tensor<X86> hx(Shape({N, H, W, L.C}, Layout_NHWC), AK_UINT8), hp(Shape({N, H, W, L.K}, Layout_NHWC), AK_UINT8);
        const size_t on = (size_t)N * H * W * L.K;
        for (size_t i = 0; i < (size_t)N * H * W * L.C; ++i) ((uint8_t*)hx.mutable_data())[i] = (uint8_t)(rng() % 256);
        std::vector<uint8_t> prev(on), want(on);
        for (auto& v : prev) v = (uint8_t)(rng() % 256);      // s8 bit patterns sitting in the (u8) output tensor
        memcpy(hp.mutable_data(), prev.data(), on);
        Tensor<MI355X> dx(hx.valid_shape(), AK_UINT8), dy(hp.valid_shape(), AK_UINT8);
        dx.set_scale({L.in_scale}); dy.set_scale({L.out_scale});
        dx.copy_from(hx); dy.copy_from(hp);
        std::vector<Tensor<MI355X>*> ins{&dx}, outs{&dy};
        ConvEltwise<MI355X, AK_INT8> op;
        SaberStatus st = op.init(ins, outs, param, SPECIFY, SABER_IMPL, ctx);
        if (st == SaberSuccess) st = op(ins, outs, param, ctx);
        outs[0]->record_event(ctx.get_compute_stream());
        outs[0]->sync();
        Tensor<X86> hy(dy.valid_shape(), AK_UINT8);
        hy.copy_from(dy);
        orc_residual_t rp = {1, 1, added_scale * (255.f / 127.f) / L.out_scale, 1 /* s8 */, 1.f, 1.f, 1.f, 1.f};
        memcpy(want.data(), prev.data(), on);
        orc_conv_i8(N, H, W, L.C, L.K, 1, 1, 0, 0, 1, 1, 1, 1, 1, 2, 2, 1, hx.data(), L.wq.data(), L.bp.data(), L.sc.data(), &rp,
                    nullptr, want.data());
        size_t bad = 0;
        for (size_t i = 0; i < on; ++i) bad += ((const uint8_t*)hy.data())[i] != want[i];
        ++g_run;
        if (st != SaberSuccess || bad) { printf("FAIL ConvEltwise<MI355X,AK_INT8> in-place sum: status %d, %zu bytes differ\n", (int)st, bad); ++g_fail; }
        else printf("ok   ConvEltwise<MI355X,AK_INT8> conv + in-place sum (s8 added into u8, beta -> sum_scale) bit-exact\n");
    }
    {   // SaberTimer<MI355X>: hipEvents around repeated dispatches on the compute stream
        Layer L(256, 256, 3, 1, 1, true, AK_UINT8, AK_UINT8, 99);
        ConvParam<MI355X> param(1, 1, 1, 1, 1, 1, 1, &L.dw, &L.db, ActivationParam<MI355X>(Active_relu));
        Conv<MI355X, AK_INT8> conv;
        run_and_check(conv, param, L, 8, 14, 14, ctx, true, "res4 branch2b, batch 8");
        SaberTimer<MI355X> t;
        Tensor<MI355X> dx(Shape({8, 14, 14, 256}, Layout_NHWC), AK_UINT8), dy(Shape({8, 14, 14, 256}, Layout_NHWC), AK_UINT8);
        dx.set_scale({L.in_scale}); dy.set_scale({L.out_scale});
        std::vector<Tensor<MI355X>*> ins{&dx}, outs{&dy};
        Conv<MI355X, AK_INT8> c2;
        if (c2.init(ins, outs, param, SPECIFY, SABER_IMPL, ctx) != SaberSuccess) { printf("FAIL timer init\n"); ++g_fail; }
        for (int i = 0; i < 20; ++i) { t.start(ctx); c2(ins, outs, param, ctx); t.end(ctx); }
        printf("SaberTimer<MI355X>: %.1f us average, %.1f us best per dispatch (event pair included)\n", t.get_average_ms() * 1e3f,
               t.get_best_ms() * 1e3f);
        if (!(t.get_average_ms() > 0.f)) { printf("FAIL timer\n"); ++g_fail; }
    }
    {   // Activation<MI355X, AK_FLOAT> beyond relu, under the reference's own BaseFunc (saber/funcs/activation.h): sigmoid, swish,
        // prelu with per-channel slopes - against the scalar formulas of test/saber/test_saber_activation.cpp:17-115
        const int n = 2, c = 5, h = 7, w = 9, count = n * c * h * w;
        Tensor<X86> hx(Shape({n, c, h, w}, Layout_NCHW), AK_FLOAT), hslope(Shape({1, c, 1, 1}, Layout_NCHW), AK_FLOAT);
        float* px = (float*)hx.mutable_data();
        for (int i = 0; i < count; ++i) px[i] = ((i * 37) % 101 - 50) * 0.07f;
        float* ps = (float*)hslope.mutable_data();
        for (int i = 0; i < c; ++i) ps[i] = 0.1f * (i + 1);
        Tensor<MI355X> dx(hx.valid_shape(), AK_FLOAT), dy(hx.valid_shape(), AK_FLOAT), dslope(hslope.valid_shape(), AK_FLOAT);
        dx.copy_from(hx);
        dslope.copy_from(hslope);
        std::vector<Tensor<MI355X>*> ins{&dx}, outs{&dy};
        struct { ActiveType a; const char* name; } kinds[] = {{Active_sigmoid, "sigmoid"}, {Active_swish, "swish"}, {Active_prelu, "prelu"}};
        for (auto& k : kinds) {
            PreluParam<MI355X> prelu(false, &dslope);
            ActivationParam<MI355X> param(k.a, 0.f, 1.3f, prelu);
            Activation<MI355X, AK_FLOAT> act;
            SaberStatus st = act.init(ins, outs, param, SPECIFY, SABER_IMPL, ctx);
            if (st == SaberSuccess) st = act(ins, outs, param, ctx);
            Tensor<X86> hy(hx.valid_shape(), AK_FLOAT);
            hy.copy_from(dy);
            const float* py = (const float*)hy.data();
            double worst = 0;
            for (int i = 0; i < count; ++i) {
                const float v = px[i];
                const int ch = (i / (h * w)) % c;
                const float want = k.a == Active_sigmoid ? 1.0f / (expf(-v) + 1.0f)
                                   : (k.a == Active_swish ? v / (1.0f + expf(-(v * 1.3f))) : (v > 0 ? v : v * ps[ch]));
                worst = std::max(worst, (double)fabsf(py[i] - want));
            }
            ++g_run;
            if (st != SaberSuccess || worst > 1e-5) { printf("FAIL Activation<MI355X,AK_FLOAT> %s: status %d, max error %.3g\n", k.name, (int)st, worst); ++g_fail; }
            else printf("ok   Activation<MI355X,AK_FLOAT> %s under BaseFunc, max error %.2g\n", k.name, worst);
        }
    }
    printf("%d cases, %d failed\n", g_run, g_fail);
    return g_fail ? 1 : 0;
}
