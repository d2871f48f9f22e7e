// include/saber_mi355x.hpp — standalone C++ mirror of the Saber operator interface for the MI355X target.
//
// Header-only, over the C ABI (saber_hip.h). Same names, argument meaning and error behaviour as the
// reference so that host code and tests read like the reference's own:
//   Shape / Tensor<T> / Context<T>                   saber/core/shape.h, tensor.h:28, context.h:29
//   SaberStatus (SaberSuccess == -1!), DataType,
//   LayoutType, ActiveType, EltwiseType              saber/saber_types.h:21-36,69-87,205-233
//   ActivationParam / ConvParam / EltwiseParam /
//   ConvEltwiseParam / FcParam                       saber/saber_funcs_param.h:48-110,470-581,586-615,1077-1140,1236-1279
//   SaberConv2D / SaberConvEltwise / SaberFc::init, create, dispatch   saber/funcs/impl/impl_base.h:33-69
//   Gemm::init / dispatch                            saber/funcs/gemm.h:27-66
// Inside the reference tree the real Tensor/Context types are used instead (integration/saber_mi355x_adaptor.h);
// this header exists so the target can be driven from C++ without the framework.
#pragma once
#include <hip/hip_runtime.h>

#include <cstring>
#include <vector>

#include "saber_hip.h"

namespace anakin {
namespace saber {

struct MI355X {};   // TargetType tag (saber_types.h:21-36 gains eMI355X)

enum SaberStatus { SaberSuccess = -1, SaberNotInitialized = 1, SaberInvalidValue = 2, SaberMemAllocFailed = 3,
                   SaberUnKownError = 4, SaberOutOfAuthority = 5, SaberOutOfMem = 6, SaberUnImplError = 7,
                   SaberWrongDevice = 8 };
enum DataType { AK_INVALID = 0, AK_HALF = 1, AK_FLOAT = 2, AK_DOUBLE = 3, AK_INT8 = 4, AK_INT16 = 5, AK_INT32 = 6,
                AK_INT64 = 7, AK_UINT8 = 8 };
enum LayoutType { Layout_invalid = 0, Layout_NCHW = 2, Layout_NHWC = 3 };
enum ActiveType { Active_unknow = 0, Active_relu = 2 };
enum EltwiseType { Eltwise_unknow = 0, Eltwise_prod = 1, Eltwise_sum = 2, Eltwise_max = 3 };
enum PoolingType { Pooling_unknow = 0, Pooling_max = 1, Pooling_average_include_padding = 2,
                   Pooling_average_exclude_padding = 3 };   // saber_types.h:305-311

inline SaberStatus to_status(int rc) {
    switch (rc) {
    case SABER_HIP_OK: return SaberSuccess;
    case SABER_HIP_INVALID_VALUE: return SaberInvalidValue;
    case SABER_HIP_UNIMPL: return SaberUnImplError;
    case SABER_HIP_OUT_OF_MEM: return SaberOutOfMem;
    default: return SaberUnKownError;
    }
}
inline int to_hip_dtype(DataType t) { return t == AK_FLOAT ? SABER_HIP_F32 : (t == AK_INT8 ? SABER_HIP_S8 : SABER_HIP_U8); }
inline size_t type_bytes(DataType t) { return t == AK_FLOAT || t == AK_INT32 ? 4 : 1; }

class Shape : public std::vector<int> {
public:
    Shape() : _layout(Layout_NCHW) {}
    Shape(std::initializer_list<int> d, LayoutType l = Layout_NCHW) : std::vector<int>(d), _layout(l) {}
    LayoutType get_layout() const { return _layout; }
    long long count() const {
        long long c = 1;
        for (int v : *this) c *= v;
        return c;
    }
    // logical N, C, H, W regardless of the storage layout
    int num() const { return (*this)[0]; }
    int channel() const { return _layout == Layout_NHWC ? (*this)[3] : (*this)[1]; }
    int height() const { return _layout == Layout_NHWC ? (*this)[1] : (*this)[2]; }
    int width() const { return _layout == Layout_NHWC ? (*this)[2] : (*this)[3]; }

private:
    LayoutType _layout;
};

template <typename TargetType>
class Context {
public:
    explicit Context(int device_id = 0, int = 0, int = 0) : _dev(device_id), _stream(nullptr) {
        (void)hipSetDevice(device_id);
        (void)hipStreamCreate(&_stream);
    }
    ~Context() {
        if (_stream) (void)hipStreamDestroy(_stream);
    }
    Context(const Context&) = delete;
    Context& operator=(const Context&) = delete;
    hipStream_t get_compute_stream() const { return _stream; }
    int get_device_id() const { return _dev; }

private:
    int _dev;
    hipStream_t _stream;
};

// Tensor<MI355X>: owns a device buffer; copy_from_host / copy_to_host play the role of Tensor::copy_from.
template <typename TargetType>
class Tensor {
public:
    Tensor() : _dtype(AK_FLOAT), _data(nullptr), _bytes(0) {}
    Tensor(const Shape& s, DataType t = AK_FLOAT) : _dtype(AK_FLOAT), _data(nullptr), _bytes(0) { re_alloc(s, t); }
    ~Tensor() {
        if (_data) (void)hipFree(_data);
    }
    Tensor(const Tensor&) = delete;
    Tensor& operator=(const Tensor&) = delete;
    SaberStatus re_alloc(const Shape& s, DataType t) {
        if (_data) (void)hipFree(_data);
        _shape = s;
        _dtype = t;
        _bytes = (size_t)s.count() * type_bytes(t);
        _data = nullptr;
        if (_bytes && hipMalloc(&_data, _bytes) != hipSuccess) return SaberOutOfMem;
        return SaberSuccess;
    }
    const void* data() const { return _data; }
    void* mutable_data() { return _data; }
    const Shape& shape() const { return _shape; }
    const Shape& valid_shape() const { return _shape; }
    long long valid_size() const { return _shape.count(); }
    DataType get_dtype() const { return _dtype; }
    LayoutType get_layout() const { return _shape.get_layout(); }
    int num() const { return _shape.num(); }
    int channel() const { return _shape.channel(); }
    int height() const { return _shape.height(); }
    int width() const { return _shape.width(); }
    void set_scale(const std::vector<float>& s) { _scale = s; }
    const std::vector<float>& get_scale() const { return _scale; }
    SaberStatus copy_from_host(const void* src) {
        return hipMemcpy(_data, src, _bytes, hipMemcpyHostToDevice) == hipSuccess ? SaberSuccess : SaberUnKownError;
    }
    SaberStatus copy_to_host(void* dst) const {
        return hipMemcpy(dst, _data, _bytes, hipMemcpyDeviceToHost) == hipSuccess ? SaberSuccess : SaberUnKownError;
    }

private:
    Shape _shape;
    DataType _dtype;
    void* _data;
    size_t _bytes;
    std::vector<float> _scale;
};

// Weights / bias live on the HOST in these param carriers (the PBlock's h_tensor, parameter.h:192+).
struct HostBlob {
    Shape shape;
    DataType dtype;
    std::vector<unsigned char> bytes;
    std::vector<float> scale;
    HostBlob() : dtype(AK_FLOAT) {}
    HostBlob(const Shape& s, DataType t, const void* src) : shape(s), dtype(t) {
        bytes.resize((size_t)s.count() * type_bytes(t));
        std::memcpy(bytes.data(), src, bytes.size());
    }
    const void* data() const { return bytes.data(); }
    long long valid_size() const { return shape.count(); }
};

template <typename TargetType>
struct ActivationParam {
    ActivationParam() : active(Active_unknow), negative_slope(0.f), has_active(false) {}
    explicit ActivationParam(ActiveType a, float slope = 0.f) : active(a), negative_slope(slope), has_active(true) {}
    ActiveType active;
    float negative_slope;
    bool has_active;
};

template <typename TargetType>
struct ConvParam {
    ConvParam() : group(-1), pad_h(-1), pad_w(-1), stride_h(-1), stride_w(-1), dilation_h(-1), dilation_w(-1),
                  weight_tensor(nullptr), bias_tensor(nullptr), alpha(1.f), beta(0.f) {}
    ConvParam(int group_in, int pad_h_in, int pad_w_in, int stride_h_in, int stride_w_in, int dilation_h_,
              int dilation_w_, HostBlob* weight, HostBlob* bias,
              ActivationParam<TargetType> activation_param_in = ActivationParam<TargetType>(), float alpha_in = 1.f,
              float beta_in = 0.f)
        : group(group_in), pad_h(pad_h_in), pad_w(pad_w_in), stride_h(stride_h_in), stride_w(stride_w_in),
          dilation_h(dilation_h_), dilation_w(dilation_w_), weight_tensor(weight), bias_tensor(bias),
          activation_param(activation_param_in), alpha(alpha_in), beta(beta_in) {}
    const HostBlob* weight() const { return weight_tensor; }
    const HostBlob* bias() const { return bias_tensor; }
    int group, pad_h, pad_w, stride_h, stride_w, dilation_h, dilation_w;
    HostBlob* weight_tensor;   // non-owning, as in the reference (saber_funcs_param.h:578-580)
    HostBlob* bias_tensor;
    ActivationParam<TargetType> activation_param;
    float alpha, beta;
    DataType beta_type = AK_FLOAT;   // dtype of the tensor being added (saber_funcs_param.h:574, default AK_FLOAT)
};

// INT8 conv + sum: the factor applied to the bytes already in the output, derived as the x86 impl does
// (jit_avx512_core_x8s8s32x_conv.cpp:174-189): the framework sets ConvParam::beta to the added tensor's scale
// (fusion_ops/conv_eltwise.cpp:185-187), the impl divides by the output scale and converts between the s8 (x/127) and
// u8 (x/255) conventions. Returns false for the combinations the reference rejects.
inline bool conv_sum_scale(float beta, DataType beta_type, DataType out_dtype, float out_scale, float* sum_scale) {
    if (beta_type == AK_INT8 && out_dtype == AK_UINT8) *sum_scale = beta * (255.f / 127.f) / out_scale;
    else if (beta_type == AK_UINT8 && out_dtype == AK_INT8) *sum_scale = beta * (127.f / 255.f) / out_scale;
    else if ((beta_type == AK_UINT8 && out_dtype == AK_UINT8) || (beta_type == AK_INT8 && out_dtype == AK_INT8))
        *sum_scale = beta / out_scale;
    else return false;
    return true;
}

template <typename TargetType>
struct EltwiseParam {
    EltwiseParam() : operation(Eltwise_unknow), has_eltwise(false) {}
    explicit EltwiseParam(EltwiseType op, std::vector<float> coeff_in = std::vector<float>({1.f, 1.f}),
                          ActivationParam<TargetType> act = ActivationParam<TargetType>())
        : operation(op), coeff(coeff_in), activation_param(act), has_eltwise(true) {}
    EltwiseType operation;
    std::vector<float> coeff;
    ActivationParam<TargetType> activation_param;
    bool has_eltwise;
};

template <typename TargetType>
struct ConvEltwiseParam {
    ConvEltwiseParam() {}
    ConvEltwiseParam(ConvParam<TargetType> c, EltwiseParam<TargetType> e) : conv_param(c), eltwise_param(e) {}
    ConvParam<TargetType> conv_param;
    EltwiseParam<TargetType> eltwise_param;
};

template <typename TargetType>
struct PoolingParam {   // saber_funcs_param.h:2085-2153
    PoolingParam() : window_h(-1), window_w(-1), pad_h(-1), pad_w(-1), stride_h(-1), stride_w(-1),
                     pooling_type(Pooling_unknow), global_pooling(false), cmp_out_shape_floor_as_conv(false) {}
    PoolingParam(int window_h_in, int window_w_in, int pad_h_in, int pad_w_in, int stride_h_in, int stride_w_in,
                 PoolingType type, bool global_pooling_in = false, bool cmp_out_shape_floor_as_conv_in = false)
        : window_h(window_h_in), window_w(window_w_in), pad_h(pad_h_in), pad_w(pad_w_in), stride_h(stride_h_in),
          stride_w(stride_w_in), pooling_type(type), global_pooling(global_pooling_in),
          cmp_out_shape_floor_as_conv(cmp_out_shape_floor_as_conv_in) {}
    int window_h, window_w, pad_h, pad_w, stride_h, stride_w;
    PoolingType pooling_type;
    bool global_pooling, cmp_out_shape_floor_as_conv;
};

template <typename TargetType>
struct ConvPoolingParam {   // saber_funcs_param.h:647-677
    ConvPoolingParam() {}
    ConvPoolingParam(ConvParam<TargetType> c, PoolingParam<TargetType> p) : conv_param(c), pooling_param(p) {}
    ConvParam<TargetType> conv_param;
    PoolingParam<TargetType> pooling_param;
};

template <typename TargetType>
struct FcParam {
    FcParam() : weights(nullptr), bias(nullptr), num_output(0), axis(1), is_transpose_weights(false) {}
    FcParam(HostBlob* w, HostBlob* b, int num_output_in, int axis_in = 1, bool is_transpose_weights_in = false)
        : weights(w), bias(b), num_output(num_output_in), axis(axis_in), is_transpose_weights(is_transpose_weights_in) {}
    HostBlob* weights;
    HostBlob* bias;
    int num_output, axis;
    bool is_transpose_weights;
};

// ---------------------------------------------------------------------------------------------------
// SaberConvEltwise<MI355X, OpDtype>: init / create / dispatch. SaberConv2D is the same class driven with an
// EltwiseParam whose has_eltwise is false (exactly how SaberConv2D<X86,AK_INT8> wraps its impls,
// saber/funcs/impl/x86/saber_conv.cpp:160-324).
// `residual`: optional third input enabling the bit-exact fused conv(->s8) + Eltwise<AK_INT8> epilogue
// (SABER_HIP_RES_ELTWISE); without it an Eltwise_sum param means the x86 in-place sum.
// ---------------------------------------------------------------------------------------------------
template <typename TargetType, DataType OpDtype>
class SaberConvEltwise {
public:
    SaberConvEltwise() : _op(nullptr), _ws(nullptr), _ctx(nullptr) {}
    ~SaberConvEltwise() {
        if (_op) saber_hip_conv2d_destroy(_op);
        if (_ws) (void)hipFree(_ws);
    }
    SaberStatus init(const std::vector<Tensor<TargetType>*>& inputs, std::vector<Tensor<TargetType>*>& outputs,
                     ConvEltwiseParam<TargetType>& param, Context<TargetType>& ctx) {
        _ctx = &ctx;
        return create(inputs, outputs, param, ctx);
    }
    SaberStatus create(const std::vector<Tensor<TargetType>*>& inputs, std::vector<Tensor<TargetType>*>& outputs,
                       ConvEltwiseParam<TargetType>& param, Context<TargetType>& ctx) {
        _ctx = &ctx;
        ConvParam<TargetType>& cp = param.conv_param;
        EltwiseParam<TargetType>& ep = param.eltwise_param;
        if (!cp.weight()) return SaberInvalidValue;
        saber_hip_conv_desc d;
        std::memset(&d, 0, sizeof d);
        const Tensor<TargetType>* in = inputs[0];
        const Tensor<TargetType>* out = outputs[0];
        d.n = in->num(); d.c = in->channel(); d.h = in->height(); d.w = in->width();
        d.k = cp.weight()->shape[0]; d.kh = cp.weight()->shape[2]; d.kw = cp.weight()->shape[3];
        d.pad_h = cp.pad_h; d.pad_w = cp.pad_w; d.stride_h = cp.stride_h; d.stride_w = cp.stride_w;
        d.dil_h = cp.dilation_h; d.dil_w = cp.dilation_w; d.group = cp.group;
        d.in_dtype = to_hip_dtype(in->get_dtype());
        d.out_dtype = to_hip_dtype(out->get_dtype());
        d.in_layout = in->get_layout() == Layout_NHWC ? SABER_HIP_NHWC : SABER_HIP_NCHW;
        d.out_layout = out->get_layout() == Layout_NHWC ? SABER_HIP_NHWC : SABER_HIP_NCHW;
        d.int8_weights = OpDtype == AK_INT8 ? 1 : 0;
        d.act = (cp.activation_param.has_active && cp.activation_param.active == Active_relu) ? SABER_HIP_ACT_RELU
                                                                                              : SABER_HIP_ACT_NONE;
        d.act_negative_slope = d.act == SABER_HIP_ACT_RELU ? cp.activation_param.negative_slope : 0.f;
        if (ep.has_eltwise && ep.operation == Eltwise_sum) {
            const bool relu = ep.activation_param.has_active && ep.activation_param.active == Active_relu;
            d.res_act = relu ? SABER_HIP_ACT_RELU : SABER_HIP_ACT_NONE;
            if (inputs.size() > 1) {   // fused two-op form
                d.res_mode = SABER_HIP_RES_ELTWISE;
                d.coeff_conv = ep.coeff[0];
                d.coeff_res = ep.coeff[1];
                d.scale_res = inputs[1]->get_scale().size() ? inputs[1]->get_scale()[0] : 1.f;
            } else {
                d.res_mode = SABER_HIP_RES_SUM_INPLACE;
                if (OpDtype == AK_INT8) {
                    const float out_scale = out->get_scale().size() ? out->get_scale()[0] : 1.f;
                    if (!conv_sum_scale(cp.beta, cp.beta_type, out->get_dtype(), out_scale, &d.sum_scale)) return SaberUnImplError;
                    d.res_has_dtype = 1;
                    d.res_dtype = to_hip_dtype(cp.beta_type);
                } else {
                    // FP32: out = act(conv + bias + 1 * out); the x86 impl adds the output whenever the eltwise is present
                    // (saber_conv_1x1.cpp:42-46), ConvParam::beta is only meaningful for INT8
                    if (ep.coeff.size() >= 2 && (ep.coeff[0] != 1.f || ep.coeff[1] != 1.f)) return SaberUnImplError;
                    d.sum_scale = 1.f;
                }
            }
        }
        if (_op) { saber_hip_conv2d_destroy(_op); _op = nullptr; }
        int rc = saber_hip_conv2d_create(&d, &_op);
        if (rc) return to_status(rc);
        const HostBlob* w = cp.weight();
        const HostBlob* b = cp.bias();
        rc = saber_hip_conv2d_set_weights(_op, w->data(), to_hip_dtype(w->dtype), w->scale.size() ? w->scale.data() : nullptr,
                                          (b && b->valid_size() > 0) ? (const float*)b->data() : nullptr,
                                          in->get_scale().size() ? in->get_scale()[0] : 1.f,
                                          out->get_scale().size() ? out->get_scale()[0] : 1.f);
        if (rc) return to_status(rc);
        if (_ws) { (void)hipFree(_ws); _ws = nullptr; }
        const size_t nb = saber_hip_conv2d_workspace_bytes(_op);
        if (nb && hipMalloc(&_ws, nb) != hipSuccess) return SaberOutOfMem;
        return SaberSuccess;
    }
    // enqueue on ctx.get_compute_stream(); the caller synchronises (Net::prediction records an event)
    SaberStatus dispatch(const std::vector<Tensor<TargetType>*>& inputs, std::vector<Tensor<TargetType>*>& outputs,
                         ConvEltwiseParam<TargetType>&) {
        if (!_op) return SaberNotInitialized;
        return to_status(saber_hip_conv2d_run(_op, inputs[0]->data(), outputs[0]->mutable_data(),
                                              inputs.size() > 1 ? inputs[1]->data() : nullptr, _ws,
                                              (saber_hip_stream_t)_ctx->get_compute_stream()));
    }
    const char* algo() const { return _op ? saber_hip_conv2d_algo(_op) : ""; }
    saber_hip_conv_t* handle() { return _op; }   // for the ops that build on a conv (SaberConv2DPooling)

private:
    saber_hip_conv_t* _op;
    void* _ws;
    Context<TargetType>* _ctx;
};

// ---------------------------------------------------------------------------------------------------
// SaberConv2DPooling<MI355X, AK_INT8> (saber/funcs/conv_pooling.h; x86: saber_conv_pooling.cpp). outputs[0] is the
// POOLED tensor. One fused kernel where saber_hip_conv2d_set_pooling has one (the ResNet stem + 3x3/2 max pooling);
// otherwise conv into an inner tensor + a pooling launch, the structure of SaberConv2DPooling<X86,AK_FLOAT> (:13-57).
// ---------------------------------------------------------------------------------------------------
template <typename TargetType, DataType OpDtype>
class SaberConv2DPooling {
public:
    SaberConv2DPooling() : _fused(false), _ctx(nullptr) {}
    SaberStatus init(const std::vector<Tensor<TargetType>*>& inputs, std::vector<Tensor<TargetType>*>& outputs,
                     ConvPoolingParam<TargetType>& param, Context<TargetType>& ctx) {
        return create(inputs, outputs, param, ctx);
    }
    SaberStatus create(const std::vector<Tensor<TargetType>*>& inputs, std::vector<Tensor<TargetType>*>& outputs,
                       ConvPoolingParam<TargetType>& param, Context<TargetType>& ctx) {
        if (OpDtype != AK_INT8) return SaberUnImplError;
        _ctx = &ctx;
        const ConvParam<TargetType>& cp = param.conv_param;
        const PoolingParam<TargetType>& pp = param.pooling_param;
        const Tensor<TargetType>* in = inputs[0];
        if (!cp.weight()) return SaberInvalidValue;
        const int k = cp.weight()->shape[0], kh = cp.weight()->shape[2], kw = cp.weight()->shape[3];
        _ch = (in->height() + 2 * cp.pad_h - (cp.dilation_h * (kh - 1) + 1)) / cp.stride_h + 1;   // funcs_utils.h:29-53
        _cw = (in->width() + 2 * cp.pad_w - (cp.dilation_w * (kw - 1) + 1)) / cp.stride_w + 1;
        // the conv's own output: same dtype and scale as the op's output (the pooling keeps both)
        _inner.re_alloc(Shape({in->num(), _ch, _cw, k}, Layout_NHWC), outputs[0]->get_dtype());
        _inner.set_scale(outputs[0]->get_scale());
        _inner_v.assign(1, &_inner);
        _cep = ConvEltwiseParam<TargetType>(cp, EltwiseParam<TargetType>());
        SaberStatus st = _conv.create(inputs, _inner_v, _cep, ctx);
        if (st != SaberSuccess) return st;
        _type = pp.pooling_type == Pooling_max ? SABER_HIP_POOL_MAX
                : (pp.pooling_type == Pooling_average_include_padding ? SABER_HIP_POOL_AVG_INCL : SABER_HIP_POOL_AVG_EXCL);
        _kh = pp.global_pooling ? _ch : pp.window_h; _kw = pp.global_pooling ? _cw : pp.window_w;
        _sh = pp.global_pooling ? _ch : pp.stride_h; _sw = pp.global_pooling ? _cw : pp.stride_w;
        _ph = pp.global_pooling ? 0 : pp.pad_h; _pw = pp.global_pooling ? 0 : pp.pad_w;
        const int rc = saber_hip_conv2d_set_pooling(_conv.handle(), _type, _kh, _kw, _sh, _sw, _ph, _pw,
                                                    pp.cmp_out_shape_floor_as_conv ? 1 : 0);
        _fused = rc == SABER_HIP_OK;
        if (!_fused && rc != SABER_HIP_UNIMPL) return to_status(rc);
        return SaberSuccess;
    }
    SaberStatus dispatch(const std::vector<Tensor<TargetType>*>& inputs, std::vector<Tensor<TargetType>*>& outputs,
                         ConvPoolingParam<TargetType>&) {
        if (_fused) return _conv.dispatch(inputs, outputs, _cep);
        SaberStatus st = _conv.dispatch(inputs, _inner_v, _cep);
        if (st != SaberSuccess) return st;
        const int dt = to_hip_dtype(outputs[0]->get_dtype());
        return to_status(saber_hip_pool2d_i8_nhwc(_inner.num(), _ch, _cw, _inner.channel(), outputs[0]->height(),
                                                  outputs[0]->width(), _kh, _kw, _sh, _sw, _ph, _pw, _type, dt, dt,
                                                  _inner.data(), outputs[0]->mutable_data(),
                                                  (saber_hip_stream_t)_ctx->get_compute_stream()));
    }
    bool fused() const { return _fused; }
    const char* algo() const { return _conv.algo(); }

private:
    SaberConvEltwise<TargetType, OpDtype> _conv;
    ConvEltwiseParam<TargetType> _cep;
    Tensor<TargetType> _inner;
    std::vector<Tensor<TargetType>*> _inner_v;
    bool _fused;
    int _type, _ch, _cw, _kh, _kw, _sh, _sw, _ph, _pw;
    Context<TargetType>* _ctx;
};

template <typename TargetType, DataType OpDtype>
class SaberConv2D {
public:
    SaberStatus init(const std::vector<Tensor<TargetType>*>& inputs, std::vector<Tensor<TargetType>*>& outputs,
                     ConvParam<TargetType>& param, Context<TargetType>& ctx) {
        _p = ConvEltwiseParam<TargetType>(param, EltwiseParam<TargetType>());
        return _impl.init(inputs, outputs, _p, ctx);
    }
    SaberStatus create(const std::vector<Tensor<TargetType>*>& inputs, std::vector<Tensor<TargetType>*>& outputs,
                       ConvParam<TargetType>& param, Context<TargetType>& ctx) {
        _p = ConvEltwiseParam<TargetType>(param, EltwiseParam<TargetType>());
        return _impl.create(inputs, outputs, _p, ctx);
    }
    SaberStatus dispatch(const std::vector<Tensor<TargetType>*>& inputs, std::vector<Tensor<TargetType>*>& outputs,
                         ConvParam<TargetType>&) {
        return _impl.dispatch(inputs, outputs, _p);
    }
    const char* algo() const { return _impl.algo(); }

private:
    SaberConvEltwise<TargetType, OpDtype> _impl;
    ConvEltwiseParam<TargetType> _p;
};

template <typename TargetType, DataType OpDtype>
class SaberFc {
public:
    SaberFc() : _op(nullptr), _ws(nullptr), _ctx(nullptr) {}
    ~SaberFc() {
        if (_op) saber_hip_fc_destroy(_op);
        if (_ws) (void)hipFree(_ws);
    }
    SaberStatus init(const std::vector<Tensor<TargetType>*>& inputs, std::vector<Tensor<TargetType>*>& outputs,
                     FcParam<TargetType>& param, Context<TargetType>& ctx) {
        _ctx = &ctx;
        saber_hip_fc_desc d;
        const Shape& s = inputs[0]->shape();
        long long m = 1, k = 1;
        for (int i = 0; i < (int)s.size(); ++i) (i < param.axis ? m : k) *= s[i];
        d.m = (int)m; d.k = (int)k; d.n = param.num_output;
        d.in_dtype = to_hip_dtype(inputs[0]->get_dtype());
        d.int8_weights = OpDtype == AK_INT8 ? 1 : 0;
        d.w_is_kn = param.is_transpose_weights ? 1 : 0;
        int rc = saber_hip_fc_create(&d, &_op);
        if (rc) return to_status(rc);
        const HostBlob* w = param.weights;
        const HostBlob* b = param.bias;
        rc = saber_hip_fc_set_weights(_op, w->data(), to_hip_dtype(w->dtype), w->scale.size() ? w->scale.data() : nullptr,
                                      (b && b->valid_size() > 0) ? (const float*)b->data() : nullptr,
                                      inputs[0]->get_scale().size() ? inputs[0]->get_scale()[0] : 1.f,
                                      outputs[0]->get_scale().size() ? outputs[0]->get_scale()[0] : 1.f);
        if (rc) return to_status(rc);
        const size_t nb = saber_hip_fc_workspace_bytes(_op);
        if (nb && hipMalloc(&_ws, nb) != hipSuccess) return SaberOutOfMem;
        return SaberSuccess;
    }
    SaberStatus dispatch(const std::vector<Tensor<TargetType>*>& inputs, std::vector<Tensor<TargetType>*>& outputs,
                         FcParam<TargetType>&) {
        if (!_op) return SaberNotInitialized;
        return to_status(saber_hip_fc_run(_op, inputs[0]->data(), (float*)outputs[0]->mutable_data(), _ws,
                                          (saber_hip_stream_t)_ctx->get_compute_stream()));
    }

private:
    saber_hip_fc_t* _op;
    void* _ws;
    Context<TargetType>* _ctx;
};

// Gemm<MI355X, SABER_IMPL, float, float>
template <typename TargetType>
class Gemm {
public:
    SaberStatus init(const bool trans_a, const bool trans_b, const int m, const int n, const int k,
                     Context<TargetType>& ctx) {
        _ta = trans_a; _tb = trans_b; _m = m; _n = n; _k = k; _ctx = &ctx;
        return SaberSuccess;
    }
    SaberStatus dispatch(const float alpha, const float beta, const float* a, const float* b, float* c) {
        return to_status(saber_hip_gemm_f32(_ta, _tb, _m, _n, _k, alpha, a, b, beta, c,
                                            (saber_hip_stream_t)_ctx->get_compute_stream()));
    }

private:
    bool _ta, _tb;
    int _m, _n, _k;
    Context<TargetType>* _ctx;
};

}  // namespace saber
}  // namespace anakin
