// saber/funcs/impl/mi355x/saber_conv.h — SaberConv2D<MI355X, OpDtype>, the Saber implementation Conv<MI355X, OpDtype>
// instantiates for SABER_IMPL (saber/funcs/conv.h:82-96). Pattern: saber/funcs/impl/x86/saber_conv.h:24-69 — the
// x86 class wraps its ConvParam into a ConvEltwiseParam without eltwise and forwards to a ConvEltwise implementation;
// here that implementation is the C-ABI binding of integration/saber_mi355x_adaptor.h.
#ifndef ANAKIN_SABER_FUNCS_IMPL_MI355X_SABER_CONV_H
#define ANAKIN_SABER_FUNCS_IMPL_MI355X_SABER_CONV_H

#include "saber/funcs/impl/impl_conv.h"
#include "saber_mi355x_adaptor.h"

namespace anakin {
namespace saber {

template <DataType OpDtype>
class SaberConv2D<MI355X, OpDtype> : public ImplBase<MI355X, OpDtype, ConvParam<MI355X> > {
public:
    SaberConv2D() {}
    ~SaberConv2D() {}

    virtual SaberStatus init(const std::vector<Tensor<MI355X>*>& inputs, std::vector<Tensor<MI355X>*>& outputs,
                             ConvParam<MI355X>& param, Context<MI355X>& ctx) {
        this->_ctx = &ctx;
        return create(inputs, outputs, param, ctx);
    }
    virtual SaberStatus create(const std::vector<Tensor<MI355X>*>& inputs, std::vector<Tensor<MI355X>*>& outputs,
                               ConvParam<MI355X>& param, Context<MI355X>& ctx) {
        this->_ctx = &ctx;
        EltwiseParam<MI355X> ep(Eltwise_sum);
        ep.has_eltwise = false;
        _cep = ConvEltwiseParam<MI355X>(param, ep);
        return _impl.create(inputs, outputs, _cep, ctx);
    }
    virtual SaberStatus dispatch(const std::vector<Tensor<MI355X>*>& inputs, std::vector<Tensor<MI355X>*>& outputs,
                                 ConvParam<MI355X>& param) {
        return _impl.dispatch(inputs, outputs, _cep);
    }
    SaberStatus trans_weights(Tensor<MI355X>&, Tensor<MI355X>&, int, int, int, int, int, int, int) { return SaberSuccess; }
    const char* algo() const { return _impl.algo(); }

private:
    SaberConvEltwiseMI355X<MI355X, OpDtype> _impl;
    ConvEltwiseParam<MI355X> _cep;
};

}  // namespace saber
}  // namespace anakin
#endif
