// include/saber_mi355x_impl.h - the MI355X Saber implementations: `SaberXxxMI355X<TargetType, OpDtype>` marshal the reference's
// Tensor / Param / Context objects into the POD descriptors of saber_hip.h; all arithmetic lives behind the C ABI.
//
// ONE copy, two users:
//   * integration/saber_mi355x_adaptor.h includes it under the REFERENCE's own headers (saber/funcs/impl/impl_base.h,
//     saber_funcs_param.h ...): that is the code a maintainer drops into saber/funcs/impl/mi355x/ and the binaries of
//     integration/build_mi355x_test.sh run (the reference's test ladders and Net<MI355X>);
//   * include/saber_mi355x.hpp includes it under a minimal stand-in for those headers (same names, same members) so that
//     a C++ program - tests/cpp - can drive the target through the Saber interface without the reference tree.
// It relies only on names both provide in namespace anakin::saber: SaberStatus / DataType / LayoutType and the Active_ /
// Eltwise_ / Pooling_ enumerators, Shape, Tensor<T>, Context<T>, TargetWrapper<T>, TargetTypeTraits<T>, ImplBase<T, Op, Param>
// and the *Param structs (member names of saber/saber_funcs_param.h).
#ifndef SABER_MI355X_IMPL_H
#define SABER_MI355X_IMPL_H

#include "saber_hip.h"

#include <cstring>
#include <vector>

namespace anakin {
namespace saber {

inline SaberStatus mi355x_status(int rc) {   // saber_types.h:223-233
    switch (rc) {
    case SABER_HIP_OK: return SaberSuccess;
    case SABER_HIP_INVALID_VALUE: return SaberInvalidValue;
    case SABER_HIP_UNIMPL: return SaberUnImplError;
    case SABER_HIP_OUT_OF_MEM: return SaberOutOfMem;
    default: return SaberUnKownError;
    }
}
inline int mi355x_dtype(DataType t) {
    return t == AK_FLOAT ? SABER_HIP_F32 : (t == AK_INT8 ? SABER_HIP_S8 : (t == AK_UINT8 ? SABER_HIP_U8 : -1));
}
inline int mi355x_layout(LayoutType l) { return l == Layout_NHWC ? SABER_HIP_NHWC : SABER_HIP_NCHW; }

// Host view of a parameter tensor (weights / bias / scales are consumed on the host by saber_hip_*_set_weights): a host
// target hands out its own pointer, a device target (MI355X) is copied down with the target's TargetWrapper, exactly once
// per init/create (cold path).
template <typename TargetType>
inline const void* mi355x_host_view(const Tensor<TargetType>& t, std::vector<char>& buf, __host_target) {
    return t.data();
}
template <typename TargetType>
inline const void* mi355x_host_view(const Tensor<TargetType>& t, std::vector<char>& buf, __device_target) {
    const size_t bytes = (size_t)t.valid_size() * t.get_dtype_size();
    buf.resize(bytes ? bytes : 1);
    TargetWrapper<TargetType>::sync_memcpy(buf.data(), 0, 0, t.data(), 0, t.device_id(), bytes, __DtoH());
    return buf.data();
}
template <typename TargetType>
inline const void* mi355x_host_view(const Tensor<TargetType>& t, std::vector<char>& buf) {
    return mi355x_host_view(t, buf, typename TargetTypeTraits<TargetType>::target_category());
}

// The factor the INT8 conv + sum post-op applies to the bytes already in the output tensor, derived as the x86 impl does
// (jit_avx512_core_x8s8s32x_conv.cpp:174-189): the framework sets ConvParam::beta to the added tensor's scale
// (framework/operators/fusion_ops/conv_eltwise.cpp:185-187); the impl divides by the output scale and converts between
// the s8 (x/127) and u8 (x/255) conventions. false for the dtype pairs the reference rejects.
inline bool mi355x_conv_sum_scale(float beta, DataType beta_type, DataType out_dtype, float out_scale, float* sum_scale) {
    if (beta_type == AK_INT8 && out_dtype == AK_UINT8) *sum_scale = beta * (255.f / 127.f) / out_scale;
    else if (beta_type == AK_UINT8 && out_dtype == AK_INT8) *sum_scale = beta * (127.f / 255.f) / out_scale;
    else if ((beta_type == AK_UINT8 && out_dtype == AK_UINT8) || (beta_type == AK_INT8 && out_dtype == AK_INT8))
        *sum_scale = beta / out_scale;
    else return false;
    return true;
}

// The activation types the target runs on f32 tensors (saber_hip_activation_f32 / saber_hip_prelu_f32)
inline bool mi355x_activation_supported(ActiveType a) {
    return a == Active_sigmoid || a == Active_relu || a == Active_tanh || a == Active_clipped_relu || a == Active_elu ||
           a == Active_stanh || a == Active_prelu || a == Active_gelu || a == Active_swish;
}
// one Activation<AK_FLOAT> on `in` -> `out` (the same tensor: in place); PReLU reads its slope tensor (device memory) and the
// channel position from the tensor's layout
template <typename TargetType>
inline int mi355x_run_activation(ActivationParam<TargetType>& p, Tensor<TargetType>& in, Tensor<TargetType>& out,
                                 saber_hip_stream_t stream) {
    const size_t count = (size_t)in.valid_size();
    if (p.active == Active_prelu) {
        PreluParam<TargetType> pr = p.prelu_param;
        if (!pr.slope) return SABER_HIP_INVALID_VALUE;
        const int inner = in.get_layout() == Layout_NHWC ? 1 : in.height() * in.width();
        return saber_hip_prelu_f32(count, in.channel(), inner, pr.channel_shared ? 1 : 0, (const float*)pr.slope->data(),
                                   (const float*)in.data(), (float*)out.mutable_data(), stream);
    }
    return saber_hip_activation_f32((int)p.active, count, p.negative_slope, p.coef, (const float*)in.data(),
                                    (float*)out.mutable_data(), stream);
}

// SaberConv2D<MI355X, OpDtype> and SaberConvEltwise<MI355X, OpDtype> share this body
// (ConvEltwiseParam = ConvParam + EltwiseParam, saber_funcs_param.h:586-615).
template <typename TargetType, DataType OpDtype>
class SaberConvEltwiseMI355X : public ImplBase<TargetType, OpDtype, ConvEltwiseParam<TargetType> > {
public:
    SaberConvEltwiseMI355X() : _op(nullptr), _elt_input(false) {}
    // Extension beyond the reference's in-place form: with it on, an Eltwise_sum param and a SECOND input tensor mean
    // `Eltwise<AK_INT8>(conv(inputs[0]) -> s8, inputs[1])` computed in the conv's epilogue (SABER_HIP_RES_ELTWISE), the bits
    // of the two reference operators run one after the other. Off (default): the x86 in-place sum onto outputs[0].
    void fuse_eltwise_input(bool on) { _elt_input = on; }
    ~SaberConvEltwiseMI355X() {
        if (_op) saber_hip_conv2d_destroy(_op);
    }

    virtual SaberStatus init(const std::vector<Tensor<TargetType>*>& inputs,
                             std::vector<Tensor<TargetType>*>& outputs,
                             ConvEltwiseParam<TargetType>& param, Context<TargetType>& ctx) {
        this->_ctx = &ctx;
        return create(inputs, outputs, param, ctx);
    }

    // init/create: geometry + algorithm choice + weight quantise/repack (cold path, host pointers)
    virtual SaberStatus create(const std::vector<Tensor<TargetType>*>& inputs,
                               std::vector<Tensor<TargetType>*>& outputs,
                               ConvEltwiseParam<TargetType>& param, Context<TargetType>& ctx) {
        this->_ctx = &ctx;
        ConvParam<TargetType>& cp = param.conv_param;
        EltwiseParam<TargetType>& ep = param.eltwise_param;
        Tensor<TargetType>* in = inputs[0];
        Tensor<TargetType>* out = outputs[0];
        const float in_scale = in->get_scale().size() ? in->get_scale()[0] : 1.f;
        const float out_scale = out->get_scale().size() ? out->get_scale()[0] : 1.f;
        saber_hip_conv_desc d;
        memset(&d, 0, sizeof d);
        d.n = in->num(); d.c = in->channel(); d.h = in->height(); d.w = in->width();
        d.k = cp.weight()->num(); d.kh = cp.weight()->height(); d.kw = cp.weight()->width();
        d.pad_h = cp.pad_h; d.pad_w = cp.pad_w; d.stride_h = cp.stride_h; d.stride_w = cp.stride_w;
        d.dil_h = cp.dilation_h; d.dil_w = cp.dilation_w; d.group = cp.group;
        d.in_dtype = mi355x_dtype(in->get_dtype());
        d.out_dtype = mi355x_dtype(out->get_dtype());
        d.in_layout = mi355x_layout(in->get_layout());
        d.out_layout = mi355x_layout(out->get_layout());
        d.int8_weights = (OpDtype == AK_INT8) ? 1 : 0;
        d.act = (cp.activation_param.has_active && cp.activation_param.active == Active_relu) ? SABER_HIP_ACT_RELU
                                                                                              : SABER_HIP_ACT_NONE;
        d.act_negative_slope = d.act == SABER_HIP_ACT_RELU ? cp.activation_param.negative_slope : 0.f;
        // Any other activation type behind an FP32 conv without an eltwise: the convolution runs without one and the activation
        // follows in place on its output - the structure of the NV impl (saber/funcs/impl/cuda/saber_conv.cpp: `_saber_act` after
        // the conv kernel). INT8 / with an eltwise: the x86 impls know only relu there.
        _post_act = cp.activation_param.has_active && cp.activation_param.active != Active_relu;
        if (_post_act && (OpDtype != AK_FLOAT || ep.has_eltwise || out->get_dtype() != AK_FLOAT ||
                          !mi355x_activation_supported(cp.activation_param.active)))
            return SaberUnImplError;
        _act = cp.activation_param;
        if (ep.has_eltwise && ep.operation == Eltwise_sum && _elt_input && inputs.size() > 1) {
            d.res_mode = SABER_HIP_RES_ELTWISE;
            d.res_act = (ep.activation_param.has_active && ep.activation_param.active == Active_relu)
                            ? SABER_HIP_ACT_RELU : SABER_HIP_ACT_NONE;
            d.coeff_conv = ep.coeff.size() > 0 ? ep.coeff[0] : 1.f;
            d.coeff_res = ep.coeff.size() > 1 ? ep.coeff[1] : 1.f;
            d.scale_res = inputs[1]->get_scale().size() ? inputs[1]->get_scale()[0] : 1.f;
        } else if (ep.has_eltwise && ep.operation == Eltwise_sum) {
            // x86 semantics: in-place sum onto the output tensor (saber_conv_eltwise.cpp:40-151; JIT with_sum)
            d.res_mode = SABER_HIP_RES_SUM_INPLACE;
            d.res_act = (ep.activation_param.has_active && ep.activation_param.active == Active_relu)
                            ? SABER_HIP_ACT_RELU : SABER_HIP_ACT_NONE;
            if (OpDtype == AK_INT8) {
                if (!mi355x_conv_sum_scale(cp.beta, cp.beta_type, out->get_dtype(), out_scale, &d.sum_scale))
                    return SaberUnImplError;
                d.res_has_dtype = 1;
                d.res_dtype = mi355x_dtype(cp.beta_type);   // the bytes in y may be s8 under a u8 output and vice versa
            } else {
                // FP32: out = act(conv + bias + 1 * out). The x86 impl adds the output whenever the eltwise is present
                // (saber_conv_1x1.cpp:42-46 `_add_output = 1.f`; saber_conv_eltwise.cpp:139-143 SaberEltwise with the
                // eltwise's coefficients) — ConvParam::beta is only meaningful for INT8. Coefficients other than (1, 1): no.
                if (ep.coeff.size() >= 2 && (ep.coeff[0] != 1.f || ep.coeff[1] != 1.f)) return SaberUnImplError;
                d.sum_scale = 1.f;
            }
        }
        if (_op) { saber_hip_conv2d_destroy(_op); _op = nullptr; }
        int rc = saber_hip_conv2d_create(&d, &_op);
        if (rc) return mi355x_status(rc);
        const Tensor<TargetType>* w = cp.weight();
        const Tensor<TargetType>* b = cp.bias();
        std::vector<char> wbuf, bbuf;
        const void* wh = mi355x_host_view(*w, wbuf);
        const void* bh = (b && b->valid_size() > 0) ? mi355x_host_view(*b, bbuf) : nullptr;
        rc = saber_hip_conv2d_set_weights(_op, wh, mi355x_dtype(w->get_dtype()),
                                          w->get_scale().size() ? w->get_scale().data() : nullptr, (const float*)bh,
                                          in_scale, out_scale);
        if (rc) return mi355x_status(rc);
        // workspace (f32 NCHW inputs are quantised / transposed into it): a tensor of the target, so its memory comes from
        // the target's TargetWrapper::mem_alloc and is released with the impl
        const size_t ws_bytes = saber_hip_conv2d_workspace_bytes(_op);
        if (ws_bytes) _ws.re_alloc(Shape({1, 1, 1, (int)ws_bytes}, Layout_NCHW), AK_INT8);
        return SaberSuccess;
    }

    // dispatch: enqueue on the context's compute stream, never sync (net.cpp:456-458 records the event)
    virtual SaberStatus dispatch(const std::vector<Tensor<TargetType>*>& inputs,
                                 std::vector<Tensor<TargetType>*>& outputs,
                                 ConvEltwiseParam<TargetType>& param) {
        // BaseFunc::init ORs the impls' statuses into SaberSuccess (= -1, base.h:126-128), so a failed create() is not
        // reported there: refuse here instead of running without an operator
        if (!_op) return SaberNotInitialized;
        saber_hip_stream_t stream = (saber_hip_stream_t)this->_ctx->get_compute_stream();
        void* ws = saber_hip_conv2d_workspace_bytes(_op) ? _ws.mutable_data() : nullptr;
        const void* res = (_elt_input && inputs.size() > 1) ? inputs[1]->data() : nullptr;
        int rc = saber_hip_conv2d_run(_op, inputs[0]->data(), outputs[0]->mutable_data(), res, ws, stream);
        if (rc == SABER_HIP_OK && _post_act) rc = mi355x_run_activation(_act, *outputs[0], *outputs[0], stream);
        return mi355x_status(rc);
    }

    // Conv<>::trans_weights static_casts to this (conv.h:103-119): the repack already happened in create()
    SaberStatus trans_weights(Tensor<TargetType>&, Tensor<TargetType>&, int, int, int, int, int, int, int) {
        return SaberSuccess;
    }
    const char* algo() const { return _op ? saber_hip_conv2d_algo(_op) : ""; }

private:
    saber_hip_conv_t* _op;
    bool _elt_input;
    bool _post_act = false;
    ActivationParam<TargetType> _act;
    Tensor<TargetType> _ws;
};

// SaberConv2DPooling<MI355X, AK_INT8> (saber/funcs/conv_pooling.h; x86: saber_conv_pooling.cpp). One fused kernel when
// saber_hip_conv2d_set_pooling accepts the combination (the ResNet stem + 3x3/2 max pooling); otherwise the conv runs
// into an inner tensor and the pooling is a second launch, the structure of SaberConv2DPooling<X86,AK_FLOAT> (:13-57).
template <typename TargetType, DataType OpDtype>
class SaberConv2DPoolingMI355X : public ImplBase<TargetType, OpDtype, ConvPoolingParam<TargetType> > {
public:
    SaberConv2DPoolingMI355X() : _op(nullptr), _ws(nullptr), _fused(false) {}
    ~SaberConv2DPoolingMI355X() { if (_op) saber_hip_conv2d_destroy(_op); }

    virtual SaberStatus init(const std::vector<Tensor<TargetType>*>& inputs, std::vector<Tensor<TargetType>*>& outputs,
                             ConvPoolingParam<TargetType>& param, Context<TargetType>& ctx) {
        this->_ctx = &ctx;
        return create(inputs, outputs, param, ctx);
    }
    virtual SaberStatus create(const std::vector<Tensor<TargetType>*>& inputs, std::vector<Tensor<TargetType>*>& outputs,
                               ConvPoolingParam<TargetType>& param, Context<TargetType>& ctx) {
        this->_ctx = &ctx;
        ConvParam<TargetType>& cp = param.conv_param;
        PoolingParam<TargetType>& pp = param.pooling_param;
        Tensor<TargetType>* in = inputs[0];
        Tensor<TargetType>* out = outputs[0];
        if (OpDtype != AK_INT8 || in->get_layout() != Layout_NHWC && in->get_dtype() != AK_FLOAT) return SaberUnImplError;
        saber_hip_conv_desc d;
        memset(&d, 0, sizeof d);
        d.n = in->num(); d.c = in->channel(); d.h = in->height(); d.w = in->width();
        d.k = cp.weight()->num(); d.kh = cp.weight()->height(); d.kw = cp.weight()->width();
        d.pad_h = cp.pad_h; d.pad_w = cp.pad_w; d.stride_h = cp.stride_h; d.stride_w = cp.stride_w;
        d.dil_h = cp.dilation_h; d.dil_w = cp.dilation_w; d.group = cp.group;
        d.in_dtype = mi355x_dtype(in->get_dtype());
        d.out_dtype = mi355x_dtype(out->get_dtype());
        d.in_layout = mi355x_layout(in->get_layout());
        d.out_layout = SABER_HIP_NHWC;
        d.int8_weights = 1;
        d.act = (cp.activation_param.has_active && cp.activation_param.active == Active_relu) ? SABER_HIP_ACT_RELU
                                                                                              : SABER_HIP_ACT_NONE;
        if (_op) { saber_hip_conv2d_destroy(_op); _op = nullptr; }
        int rc = saber_hip_conv2d_create(&d, &_op);
        if (rc) return mi355x_status(rc);
        const Tensor<TargetType>* w = cp.weight();
        const Tensor<TargetType>* b = cp.bias();
        // the pooling keeps the conv's scale (SaberPooling<X86,AK_INT8>::init): the op's output scale is the conv's
        std::vector<char> wbuf, bbuf;
        const void* wh = mi355x_host_view(*w, wbuf);
        const void* bh = (b && b->valid_size() > 0) ? mi355x_host_view(*b, bbuf) : nullptr;
        rc = saber_hip_conv2d_set_weights(_op, wh, mi355x_dtype(w->get_dtype()),
                                          w->get_scale().size() ? w->get_scale().data() : nullptr, (const float*)bh,
                                          in->get_scale().size() ? in->get_scale()[0] : 1.f,
                                          out->get_scale().size() ? out->get_scale()[0] : 1.f);
        if (rc) return mi355x_status(rc);
        const size_t ws_bytes = saber_hip_conv2d_workspace_bytes(_op);
        if (ws_bytes) _wst.re_alloc(Shape({1, 1, 1, (int)ws_bytes}, Layout_NCHW), AK_INT8);
        _ws = ws_bytes ? _wst.mutable_data() : nullptr;
        _type = pp.pooling_type == Pooling_max ? SABER_HIP_POOL_MAX
                : (pp.pooling_type == Pooling_average_include_padding ? SABER_HIP_POOL_AVG_INCL : SABER_HIP_POOL_AVG_EXCL);
        saber_hip_conv2d_out_shape(_op, &_ch, &_cw);
        _kh = pp.global_pooling ? _ch : pp.window_h; _kw = pp.global_pooling ? _cw : pp.window_w;
        _sh = pp.global_pooling ? _ch : pp.stride_h; _sw = pp.global_pooling ? _cw : pp.stride_w;
        _ph = pp.global_pooling ? 0 : pp.pad_h; _pw = pp.global_pooling ? 0 : pp.pad_w;
        rc = saber_hip_conv2d_set_pooling(_op, _type, _kh, _kw, _sh, _sw, _ph, _pw, pp.cmp_out_shape_floor_as_conv ? 1 : 0);
        _fused = rc == SABER_HIP_OK;
        if (!_fused && rc != SABER_HIP_UNIMPL) return mi355x_status(rc);
        if (!_fused) {   // inner tensor of the conv's shape, NHWC, the output's dtype (target allocator in the real target)
            _inner.re_alloc(Shape({d.n, _ch, _cw, d.k}, Layout_NHWC), out->get_dtype());
        }
        return SaberSuccess;
    }
    bool fused() const { return _fused; }
    const char* algo() const { return _op ? saber_hip_conv2d_algo(_op) : ""; }
    virtual SaberStatus dispatch(const std::vector<Tensor<TargetType>*>& inputs, std::vector<Tensor<TargetType>*>& outputs,
                                 ConvPoolingParam<TargetType>& param) {
        if (!_op) return SaberNotInitialized;
        saber_hip_stream_t stream = (saber_hip_stream_t)this->_ctx->get_compute_stream();
        if (_fused)
            return mi355x_status(saber_hip_conv2d_run(_op, inputs[0]->data(), outputs[0]->mutable_data(), nullptr, _ws, stream));
        int rc = saber_hip_conv2d_run(_op, inputs[0]->data(), _inner.mutable_data(), nullptr, _ws, stream);
        if (rc) return mi355x_status(rc);
        const int dt = mi355x_dtype(outputs[0]->get_dtype());
        return mi355x_status(saber_hip_pool2d_i8_nhwc(_inner.num(), _ch, _cw, _inner.channel(), outputs[0]->height(),
                                                      outputs[0]->width(), _kh, _kw, _sh, _sw, _ph, _pw, _type, dt, dt,
                                                      _inner.data(), outputs[0]->mutable_data(), stream));
    }

private:
    saber_hip_conv_t* _op;
    void* _ws;
    bool _fused;
    int _type, _ch, _cw, _kh, _kw, _sh, _sw, _ph, _pw;
    Tensor<TargetType> _inner, _wst;
};

// Fc<MI355X, OpDtype> (saber/funcs/fc.h:48-127)
template <typename TargetType, DataType OpDtype>
class SaberFcMI355X : public ImplBase<TargetType, OpDtype, FcParam<TargetType> > {
public:
    SaberFcMI355X() : _op(nullptr), _ws(nullptr) {}
    ~SaberFcMI355X() { if (_op) saber_hip_fc_destroy(_op); }

    virtual SaberStatus init(const std::vector<Tensor<TargetType>*>& inputs,
                             std::vector<Tensor<TargetType>*>& outputs, FcParam<TargetType>& param,
                             Context<TargetType>& ctx) {
        this->_ctx = &ctx;
        return create(inputs, outputs, param, ctx);
    }
    virtual SaberStatus create(const std::vector<Tensor<TargetType>*>& inputs,
                               std::vector<Tensor<TargetType>*>& outputs, FcParam<TargetType>& param,
                               Context<TargetType>& ctx) {
        this->_ctx = &ctx;
        saber_hip_fc_desc d;
        d.m = inputs[0]->count_valid(0, param.axis);
        d.k = inputs[0]->count_valid(param.axis, inputs[0]->dims());
        d.n = param.num_output;
        d.in_dtype = mi355x_dtype(inputs[0]->get_dtype());
        d.int8_weights = (OpDtype == AK_INT8) ? 1 : 0;
        d.w_is_kn = param.is_transpose_weights ? 1 : 0;
        if (_op) { saber_hip_fc_destroy(_op); _op = nullptr; }
        int rc = saber_hip_fc_create(&d, &_op);
        if (rc) return mi355x_status(rc);
        const Tensor<TargetType>* w = param.weights;
        const Tensor<TargetType>* b = param.bias;
        std::vector<char> wbuf, bbuf;
        const void* wh = mi355x_host_view(*w, wbuf);
        const void* bh = (b && b->valid_size() > 0) ? mi355x_host_view(*b, bbuf) : nullptr;
        rc = saber_hip_fc_set_weights(_op, wh, mi355x_dtype(w->get_dtype()),
                                      w->get_scale().size() ? w->get_scale().data() : nullptr, (const float*)bh,
                                      inputs[0]->get_scale().size() ? inputs[0]->get_scale()[0] : 1.f,
                                      outputs[0]->get_scale().size() ? outputs[0]->get_scale()[0] : 1.f);
        if (rc) return mi355x_status(rc);
        const size_t ws_bytes = saber_hip_fc_workspace_bytes(_op);   // an f32 input of an INT8 fc is quantised into it
        if (ws_bytes) _wst.re_alloc(Shape({1, 1, 1, (int)ws_bytes}, Layout_NCHW), AK_INT8);
        _ws = ws_bytes ? _wst.mutable_data() : nullptr;
        return SaberSuccess;
    }
    virtual SaberStatus dispatch(const std::vector<Tensor<TargetType>*>& inputs,
                                 std::vector<Tensor<TargetType>*>& outputs, FcParam<TargetType>& param) {
        if (!_op) return SaberNotInitialized;
        saber_hip_stream_t stream = (saber_hip_stream_t)this->_ctx->get_compute_stream();
        return mi355x_status(saber_hip_fc_run(_op, inputs[0]->data(), (float*)outputs[0]->mutable_data(), _ws, stream));
    }

private:
    saber_hip_fc_t* _op;
    void* _ws;
    Tensor<TargetType> _wst;
};

// Pooling<MI355X, OpDtype> (saber/funcs/pooling.h:69-130; x86: saber_pooling.cpp:312-654).
//   AK_INT8 op : s8/u8 NHWC in -> s8/u8 NHWC out, the output inherits the input's scale (SaberPooling<X86,AK_INT8>::init).
//   AK_FLOAT op: f32 in (NCHW or NHWC) -> f32 out of the same layout; an 8-bit NHWC input is dequantised on entry and the
//                result is f32 NCHW (saber_pooling.cpp:399-402).
template <typename TargetType, DataType OpDtype>
class SaberPoolingMI355X : public ImplBase<TargetType, OpDtype, PoolingParam<TargetType> > {
public:
    virtual SaberStatus init(const std::vector<Tensor<TargetType>*>& inputs, std::vector<Tensor<TargetType>*>& outputs,
                             PoolingParam<TargetType>& param, Context<TargetType>& ctx) {
        this->_ctx = &ctx;
        if (OpDtype == AK_INT8) outputs[0]->set_scale(inputs[0]->get_scale());
        return create(inputs, outputs, param, ctx);
    }
    virtual SaberStatus create(const std::vector<Tensor<TargetType>*>& inputs, std::vector<Tensor<TargetType>*>& outputs,
                               PoolingParam<TargetType>& param, Context<TargetType>& ctx) {
        this->_ctx = &ctx;
        const DataType it = inputs[0]->get_dtype(), ot = outputs[0]->get_dtype();
        const bool in8 = it == AK_INT8 || it == AK_UINT8, out8 = ot == AK_INT8 || ot == AK_UINT8;
        if (OpDtype == AK_INT8) {
            if (!in8 || !out8 || inputs[0]->get_layout() != Layout_NHWC || outputs[0]->get_layout() != Layout_NHWC)
                return SaberUnImplError;
        } else {
            if (ot != AK_FLOAT) return SaberUnImplError;
            if (in8 && inputs[0]->get_layout() != Layout_NHWC) return SaberUnImplError;
            if (!in8 && (it != AK_FLOAT || inputs[0]->get_layout() != outputs[0]->get_layout())) return SaberUnImplError;
        }
        _type = param.pooling_type == Pooling_max ? SABER_HIP_POOL_MAX
                : (param.pooling_type == Pooling_average_include_padding ? SABER_HIP_POOL_AVG_INCL
                                                                         : SABER_HIP_POOL_AVG_EXCL);
        return SaberSuccess;
    }
    virtual SaberStatus dispatch(const std::vector<Tensor<TargetType>*>& inputs, std::vector<Tensor<TargetType>*>& outputs,
                                 PoolingParam<TargetType>& p) {
        saber_hip_stream_t stream = (saber_hip_stream_t)this->_ctx->get_compute_stream();
        Tensor<TargetType>* in = inputs[0];
        Tensor<TargetType>* out = outputs[0];
        const int n = in->num(), c = in->channel(), h = in->height(), w = in->width(), oh = out->height(), ow = out->width();
        const DataType it = in->get_dtype();
        if (OpDtype == AK_INT8)
            return mi355x_status(saber_hip_pool2d_i8_nhwc(n, h, w, c, oh, ow, p.window_h, p.window_w, p.stride_h, p.stride_w,
                                                          p.pad_h, p.pad_w, _type, mi355x_dtype(it),
                                                          mi355x_dtype(out->get_dtype()), in->data(), out->mutable_data(),
                                                          stream));
        if (it == AK_INT8 || it == AK_UINT8)
            return mi355x_status(saber_hip_pool2d_f32_from_i8(n, h, w, c, oh, ow, p.window_h, p.window_w, p.stride_h,
                                                              p.stride_w, p.pad_h, p.pad_w, _type, mi355x_dtype(it),
                                                              in->get_scale().size() ? in->get_scale()[0] : 1.f, in->data(),
                                                              (float*)out->mutable_data(), stream));
        return mi355x_status(saber_hip_pool2d_f32(n, h, w, c, oh, ow, p.window_h, p.window_w, p.stride_h, p.stride_w, p.pad_h,
                                                  p.pad_w, _type, mi355x_layout(in->get_layout()), (const float*)in->data(),
                                                  (float*)out->mutable_data(), stream));
    }

private:
    int _type;
};

// Eltwise<MI355X, OpDtype> (saber/funcs/eltwise.h; x86: saber_eltwise.cpp:40-113): the two-input sum (+ relu) of a residual
// block. AK_INT8: s8 NHWC inputs and output, `saturate(roundf(relu(c0*q0*s0 + c1*q1*s1)))` — the output scale is not
// applied by the reference (saber_eltwise.cpp:85), callers fold it into the coefficients.
template <typename TargetType, DataType OpDtype>
class SaberEltwiseMI355X : public ImplBase<TargetType, OpDtype, EltwiseParam<TargetType> > {
public:
    virtual SaberStatus init(const std::vector<Tensor<TargetType>*>& inputs, std::vector<Tensor<TargetType>*>& outputs,
                             EltwiseParam<TargetType>& param, Context<TargetType>& ctx) {
        this->_ctx = &ctx;
        return create(inputs, outputs, param, ctx);
    }
    virtual SaberStatus create(const std::vector<Tensor<TargetType>*>& inputs, std::vector<Tensor<TargetType>*>& outputs,
                               EltwiseParam<TargetType>& param, Context<TargetType>& ctx) {
        this->_ctx = &ctx;
        if (param.operation != Eltwise_sum || inputs.size() != 2 || param.coeff.size() < 2) return SaberUnImplError;
        const ActivationParam<TargetType>& ap = param.activation_param;
        if (param.has_eltwise && ap.has_active && ap.active != Active_relu) return SaberUnImplError;
        const DataType want = OpDtype == AK_INT8 ? AK_INT8 : AK_FLOAT;
        for (size_t i = 0; i < inputs.size(); ++i) {
            if (inputs[i]->get_dtype() != want) return SaberUnImplError;
            if (OpDtype == AK_INT8 && (inputs[i]->get_layout() != Layout_NHWC || inputs[i]->get_scale().empty()))
                return SaberUnImplError;
            if (inputs[i]->get_layout() != inputs[0]->get_layout()) return SaberUnImplError;
        }
        if (outputs[0]->get_dtype() != want || outputs[0]->get_layout() != inputs[0]->get_layout()) return SaberUnImplError;
        return SaberSuccess;
    }
    virtual SaberStatus dispatch(const std::vector<Tensor<TargetType>*>& inputs, std::vector<Tensor<TargetType>*>& outputs,
                                 EltwiseParam<TargetType>& param) {
        saber_hip_stream_t stream = (saber_hip_stream_t)this->_ctx->get_compute_stream();
        const int relu = (param.has_eltwise && param.activation_param.has_active &&
                          param.activation_param.active == Active_relu) ? 1 : 0;
        const size_t count = (size_t)inputs[0]->valid_size();
        if (OpDtype == AK_INT8)
            return mi355x_status(saber_hip_eltwise_sum_i8(count, (const int8_t*)inputs[0]->data(),
                                                          (const int8_t*)inputs[1]->data(), inputs[0]->get_scale()[0],
                                                          inputs[1]->get_scale()[0], param.coeff[0], param.coeff[1], relu,
                                                          (int8_t*)outputs[0]->mutable_data(), stream));
        return mi355x_status(saber_hip_eltwise_sum_f32(count, (const float*)inputs[0]->data(), (const float*)inputs[1]->data(),
                                                       param.coeff[0], param.coeff[1], relu,
                                                       (float*)outputs[0]->mutable_data(), stream));
    }
};

// Softmax<MI355X, AK_FLOAT> (saber/funcs/softmax.h; x86: saber_softmax.cpp) over `axis` when everything after it is 1
// (the classifier head: [n, classes, 1, 1], axis 1)
template <typename TargetType, DataType OpDtype>
class SaberSoftmaxMI355X : public ImplBase<TargetType, OpDtype, SoftmaxParam<TargetType> > {
public:
    virtual SaberStatus init(const std::vector<Tensor<TargetType>*>& inputs, std::vector<Tensor<TargetType>*>& outputs,
                             SoftmaxParam<TargetType>& param, Context<TargetType>& ctx) {
        this->_ctx = &ctx;
        return create(inputs, outputs, param, ctx);
    }
    virtual SaberStatus create(const std::vector<Tensor<TargetType>*>& inputs, std::vector<Tensor<TargetType>*>& outputs,
                               SoftmaxParam<TargetType>& param, Context<TargetType>& ctx) {
        this->_ctx = &ctx;
        if (OpDtype != AK_FLOAT || inputs[0]->get_dtype() != AK_FLOAT || outputs[0]->get_dtype() != AK_FLOAT)
            return SaberUnImplError;
        _rows = inputs[0]->count_valid(0, param.axis);
        _cols = inputs[0]->valid_shape()[param.axis];
        if (inputs[0]->count_valid(param.axis + 1, inputs[0]->dims()) != 1) return SaberUnImplError;
        return SaberSuccess;
    }
    virtual SaberStatus dispatch(const std::vector<Tensor<TargetType>*>& inputs, std::vector<Tensor<TargetType>*>& outputs,
                                 SoftmaxParam<TargetType>& param) {
        return mi355x_status(saber_hip_softmax_f32(_rows, _cols, (const float*)inputs[0]->data(),
                                                   (float*)outputs[0]->mutable_data(),
                                                   (saber_hip_stream_t)this->_ctx->get_compute_stream()));
    }

private:
    int _rows, _cols;
};

// Activation<MI355X, AK_FLOAT> (saber/funcs/activation.h; x86: saber_activation.cpp:136-262): relu (the standalone ReLU operator,
// framework/operators/relu.cpp), sigmoid, tanh, clipped relu, elu, stanh, swish, gelu, prelu. INT8 tensors: SaberUnImplError.
template <typename TargetType, DataType OpDtype>
class SaberActivationMI355X : public ImplBase<TargetType, OpDtype, ActivationParam<TargetType> > {
public:
    virtual SaberStatus init(const std::vector<Tensor<TargetType>*>& inputs, std::vector<Tensor<TargetType>*>& outputs,
                             ActivationParam<TargetType>& param, Context<TargetType>& ctx) {
        this->_ctx = &ctx;
        return create(inputs, outputs, param, ctx);
    }
    virtual SaberStatus create(const std::vector<Tensor<TargetType>*>& inputs, std::vector<Tensor<TargetType>*>& outputs,
                               ActivationParam<TargetType>& param, Context<TargetType>& ctx) {
        this->_ctx = &ctx;
        if (OpDtype != AK_FLOAT || !mi355x_activation_supported(param.active)) return SaberUnImplError;
        if (inputs[0]->get_dtype() != AK_FLOAT || outputs[0]->get_dtype() != AK_FLOAT) return SaberUnImplError;
        return SaberSuccess;
    }
    virtual SaberStatus dispatch(const std::vector<Tensor<TargetType>*>& inputs, std::vector<Tensor<TargetType>*>& outputs,
                                 ActivationParam<TargetType>& param) {
        saber_hip_stream_t stream = (saber_hip_stream_t)this->_ctx->get_compute_stream();
        for (size_t i = 0; i < inputs.size(); ++i) {
            int rc = mi355x_run_activation(param, *inputs[i], *outputs[i], stream);
            if (rc) return mi355x_status(rc);
        }
        return SaberSuccess;
    }
};

// Gemm<MI355X, SABER_IMPL, float, float> (saber/funcs/gemm.h:27-66): raw row-major pointers
template <typename TargetType>
class SaberGemmMI355X {
public:
    SaberStatus init(const bool trans_a, const bool trans_b, const int m, const int n, const int k,
                     Context<TargetType> ctx) {
        _ta = trans_a; _tb = trans_b; _m = m; _n = n; _k = k; _ctx = ctx;
        return SaberSuccess;
    }
    SaberStatus dispatch(const float alpha, const float beta, const float* a, const float* b, float* c) {
        return mi355x_status(saber_hip_gemm_f32(_ta, _tb, _m, _n, _k, alpha, a, b, beta, c,
                                                (saber_hip_stream_t)_ctx.get_compute_stream()));
    }

private:
    bool _ta, _tb;
    int _m, _n, _k;
    Context<TargetType> _ctx;
};

}  // namespace saber
}  // namespace anakin

#endif
