// include/saber_mi355x.hpp — the MI355X Saber target driven from C++ WITHOUT the reference tree.
//
// The operator implementations are include/saber_mi355x_impl.h — the very code integration/saber_mi355x_adaptor.h compiles
// under the reference's own headers. This file only supplies a minimal stand-in for those headers: the same names, members
// and error behaviour, so that host code and tests read like the reference's own:
//   Shape / Tensor<T> / Context<T> / TargetWrapper<T>   saber/core/shape.h, tensor.h:28, context.h:29, target_wrapper.h
//   SaberStatus (SaberSuccess == -1!), DataType,
//   LayoutType, ActiveType, EltwiseType, PoolingType     saber/saber_types.h:21-36,69-87,205-233,305-311
//   ActivationParam / ConvParam / EltwiseParam / ConvEltwiseParam / PoolingParam / ConvPoolingParam / FcParam /
//   SoftmaxParam                                         saber/saber_funcs_param.h:48-110,470-677,1077-1140,1236-1279,2085-2153
//   ImplBase<T, Op, Param>                               saber/funcs/impl/impl_base.h:33-69
//   SaberConv2D / SaberConvEltwise / SaberConv2DPooling / SaberFc / SaberPooling / SaberEltwise / SaberSoftmax /
//   SaberActivation <MI355X, OpDtype>, Gemm<MI355X>      the per-target specialisations of saber/funcs/impl/<target>/
// As in the reference, weights and bias are TENSORS OF THE TARGET (device memory here); the implementations copy them to
// the host once per init (saber_mi355x_impl.h: mi355x_host_view). HostBlob below is a convenience that builds such a tensor
// from host data.
#pragma once
#include <hip/hip_runtime.h>

#include <cstring>
#include <memory>
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
enum ActiveType { Active_unknow = 0, Active_sigmoid = 1, Active_relu = 2, Active_tanh = 3, Active_clipped_relu = 4, Active_elu = 5,
                  Active_identity = 6, Active_stanh = 9, Active_prelu = 10, Active_gelu = 11, Active_swish = 12 };   // saber_types.h:283-293
enum EltwiseType { Eltwise_unknow = 0, Eltwise_prod = 1, Eltwise_sum = 2, Eltwise_max = 3 };
enum PoolingType { Pooling_unknow = 0, Pooling_max = 1, Pooling_average_include_padding = 2,
                   Pooling_average_exclude_padding = 3 };   // saber_types.h:305-311

inline size_t type_bytes(DataType t) { return t == AK_FLOAT || t == AK_INT32 ? 4 : 1; }

// target categories (saber/core/target_traits.h): MI355X is a device target - parameter tensors are copied down
struct __host_target {};
struct __device_target {};
struct __DtoH {};
template <typename TargetType>
struct TargetTypeTraits {
    typedef __device_target target_category;
};
template <typename TargetType>
struct TargetWrapper {   // the one entry point the implementations use (target_wrapper.h: sync_memcpy)
    static void sync_memcpy(void* dst, size_t dst_off, int, const void* src, size_t src_off, int, size_t count, __DtoH) {
        (void)hipMemcpy((char*)dst + dst_off, (const char*)src + src_off, count, hipMemcpyDeviceToHost);
    }
};

class Shape : public std::vector<int> {
public:
    Shape() : _layout(Layout_NCHW) {}
    Shape(std::initializer_list<int> d, LayoutType l = Layout_NCHW) : std::vector<int>(d), _layout(l) {}
    LayoutType get_layout() const { return _layout; }
    long long count() const { return count(0, (int)size()); }
    long long count(int start, int end) const {
        long long c = 1;
        for (int i = start; i < end && i < (int)size(); ++i) c *= (*this)[i];
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
class Context {   // copyable like the reference's (the copies share the stream)
public:
    explicit Context(int device_id = 0, int = 0, int = 0) : _dev(device_id) {
        (void)hipSetDevice(device_id);
        hipStream_t s = nullptr;
        (void)hipStreamCreate(&s);
        _stream = std::shared_ptr<void>((void*)s, [](void* p) {
            if (p) (void)hipStreamDestroy((hipStream_t)p);
        });
    }
    hipStream_t get_compute_stream() const { return (hipStream_t)_stream.get(); }
    int get_device_id() const { return _dev; }

private:
    int _dev;
    std::shared_ptr<void> _stream;
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
    long long count_valid(int start, int end) const { return _shape.count(start, end); }
    int dims() const { return (int)_shape.size(); }
    DataType get_dtype() const { return _dtype; }
    size_t get_dtype_size() const { return type_bytes(_dtype); }
    int device_id() const { return 0; }
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

// A parameter tensor of the target filled from host data (the PBlock's d_tensor, parameter.h:192+): weights are
// [K, C, kh, kw] (NCHW-shaped), bias [1, K, 1, 1]; `scale` = per-output-channel weight scales of pre-quantised s8 weights.
struct HostBlob : public Tensor<MI355X> {
    HostBlob(const Shape& s, DataType t, const void* src, const std::vector<float>& scale = std::vector<float>())
        : Tensor<MI355X>(s, t) {
        (void)copy_from_host(src);
        if (!scale.empty()) set_scale(scale);
    }
};

template <typename TargetType, DataType OpDtype, typename Param>
class ImplBase {   // saber/funcs/impl/impl_base.h:33-69
public:
    ImplBase() : _ctx(nullptr) {}
    virtual ~ImplBase() {}
    virtual SaberStatus init(const std::vector<Tensor<TargetType>*>& inputs, std::vector<Tensor<TargetType>*>& outputs,
                             Param& param, Context<TargetType>& ctx) { return SaberUnImplError; }
    virtual SaberStatus create(const std::vector<Tensor<TargetType>*>& inputs, std::vector<Tensor<TargetType>*>& outputs,
                               Param& param, Context<TargetType>& ctx) { return SaberUnImplError; }
    virtual SaberStatus dispatch(const std::vector<Tensor<TargetType>*>& inputs, std::vector<Tensor<TargetType>*>& outputs,
                                 Param& param) { return SaberUnImplError; }

protected:
    Context<TargetType>* _ctx;
};

template <typename TargetType>
struct PreluParam {                     // saber_funcs_param.h: channel_shared + the slope tensor (non-owning)
    PreluParam() : channel_shared(false), slope(nullptr) {}
    PreluParam(bool shared, Tensor<TargetType>* slope_in) : channel_shared(shared), slope(slope_in) {}
    bool channel_shared;
    Tensor<TargetType>* slope;
};
template <typename TargetType>
struct ActivationParam {                // saber_funcs_param.h:47-110
    ActivationParam() : active(Active_unknow), negative_slope(0.f), coef(1.f), has_active(false) {}
    explicit ActivationParam(ActiveType a, float slope = 0.f, float co = 1.f, PreluParam<TargetType> prelu = PreluParam<TargetType>())
        : active(a), negative_slope(slope), coef(co), has_active(true), prelu_param(prelu) {}
    ActiveType active;
    float negative_slope;
    float coef;
    bool has_active;
    PreluParam<TargetType> prelu_param;
};

template <typename TargetType>
struct ConvParam {
    ConvParam() : group(-1), pad_h(-1), pad_w(-1), stride_h(-1), stride_w(-1), dilation_h(-1), dilation_w(-1),
                  weight_tensor(nullptr), bias_tensor(nullptr), alpha(1.f), beta(0.f) {}
    ConvParam(int group_in, int pad_h_in, int pad_w_in, int stride_h_in, int stride_w_in, int dilation_h_,
              int dilation_w_, Tensor<TargetType>* weight, Tensor<TargetType>* bias,
              ActivationParam<TargetType> activation_param_in = ActivationParam<TargetType>(), float alpha_in = 1.f,
              float beta_in = 0.f)
        : group(group_in), pad_h(pad_h_in), pad_w(pad_w_in), stride_h(stride_h_in), stride_w(stride_w_in),
          dilation_h(dilation_h_), dilation_w(dilation_w_), weight_tensor(weight), bias_tensor(bias),
          activation_param(activation_param_in), alpha(alpha_in), beta(beta_in) {}
    const Tensor<TargetType>* weight() const { return weight_tensor; }
    const Tensor<TargetType>* bias() const { return bias_tensor; }
    int group, pad_h, pad_w, stride_h, stride_w, dilation_h, dilation_w;
    Tensor<TargetType>* weight_tensor;   // non-owning, as in the reference (saber_funcs_param.h:578-580)
    Tensor<TargetType>* bias_tensor;
    ActivationParam<TargetType> activation_param;
    float alpha, beta;
    DataType beta_type = AK_FLOAT;   // dtype of the tensor being added (saber_funcs_param.h:574, default AK_FLOAT)
};

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
    FcParam(Tensor<TargetType>* w, Tensor<TargetType>* b, int num_output_in, int axis_in = 1, bool is_transpose_weights_in = false)
        : weights(w), bias(b), num_output(num_output_in), axis(axis_in), is_transpose_weights(is_transpose_weights_in) {}
    Tensor<TargetType>* weights;
    Tensor<TargetType>* bias;
    int num_output, axis;
    bool is_transpose_weights;
};

template <typename TargetType>
struct SoftmaxParam {   // saber_funcs_param.h: axis only
    explicit SoftmaxParam(int axis_in = 1) : axis(axis_in) {}
    int axis;
};

}  // namespace saber
}  // namespace anakin

#include "saber_mi355x_impl.h"

namespace anakin {
namespace saber {

// ---- the per-target class names of saber/funcs/impl/<target>/ (integration/mi355x/funcs/*.h in the reference tree) -------
// SaberConvEltwise: an Eltwise_sum param with a second input tensor means the fused conv(->s8) + Eltwise<AK_INT8> epilogue
// here (SaberConvEltwiseMI355X::fuse_eltwise_input); with one input, the x86 in-place sum.
template <typename TargetType, DataType OpDtype>
class SaberConvEltwise : public SaberConvEltwiseMI355X<TargetType, OpDtype> {
public:
    SaberConvEltwise() { this->fuse_eltwise_input(true); }
};

// SaberConv2D wraps its ConvParam into a ConvEltwiseParam without eltwise (saber/funcs/impl/x86/saber_conv.h:24-69)
template <typename TargetType, DataType OpDtype>
class SaberConv2D : public ImplBase<TargetType, OpDtype, ConvParam<TargetType> > {
public:
    virtual SaberStatus init(const std::vector<Tensor<TargetType>*>& inputs, std::vector<Tensor<TargetType>*>& outputs,
                             ConvParam<TargetType>& param, Context<TargetType>& ctx) {
        return create(inputs, outputs, param, ctx);
    }
    virtual SaberStatus create(const std::vector<Tensor<TargetType>*>& inputs, std::vector<Tensor<TargetType>*>& outputs,
                               ConvParam<TargetType>& param, Context<TargetType>& ctx) {
        this->_ctx = &ctx;
        EltwiseParam<TargetType> ep(Eltwise_sum);
        ep.has_eltwise = false;
        _cep = ConvEltwiseParam<TargetType>(param, ep);
        return _impl.create(inputs, outputs, _cep, ctx);
    }
    virtual SaberStatus dispatch(const std::vector<Tensor<TargetType>*>& inputs, std::vector<Tensor<TargetType>*>& outputs,
                                 ConvParam<TargetType>&) {
        return _impl.dispatch(inputs, outputs, _cep);
    }
    const char* algo() const { return _impl.algo(); }

private:
    SaberConvEltwiseMI355X<TargetType, OpDtype> _impl;
    ConvEltwiseParam<TargetType> _cep;
};

template <typename TargetType, DataType OpDtype>
using SaberConv2DPooling = SaberConv2DPoolingMI355X<TargetType, OpDtype>;
template <typename TargetType, DataType OpDtype>
using SaberFc = SaberFcMI355X<TargetType, OpDtype>;
template <typename TargetType, DataType OpDtype>
using SaberPooling = SaberPoolingMI355X<TargetType, OpDtype>;
template <typename TargetType, DataType OpDtype>
using SaberEltwise = SaberEltwiseMI355X<TargetType, OpDtype>;
template <typename TargetType, DataType OpDtype>
using SaberSoftmax = SaberSoftmaxMI355X<TargetType, OpDtype>;
template <typename TargetType, DataType OpDtype>
using SaberActivation = SaberActivationMI355X<TargetType, OpDtype>;
template <typename TargetType>
using Gemm = SaberGemmMI355X<TargetType>;

}  // namespace saber
}  // namespace anakin
