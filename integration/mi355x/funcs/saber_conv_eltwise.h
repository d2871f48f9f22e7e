// saber/funcs/impl/mi355x/saber_conv_eltwise.h — SaberConvEltwise<MI355X, OpDtype>, the implementation
// ConvEltwise<MI355X, OpDtype> instantiates (saber/funcs/conv_eltwise.h): conv + in-place sum (+relu) onto the output
// tensor, the x86 semantics (saber_conv_eltwise.cpp:40-217). The whole body is the C-ABI binding shared with SaberConv2D.
#ifndef ANAKIN_SABER_FUNCS_IMPL_MI355X_SABER_CONV_ELTWISE_H
#define ANAKIN_SABER_FUNCS_IMPL_MI355X_SABER_CONV_ELTWISE_H

#include "saber/funcs/impl/impl_conv_eltwise.h"
#include "saber_mi355x_adaptor.h"

namespace anakin {
namespace saber {

template <DataType OpDtype>
class SaberConvEltwise<MI355X, OpDtype> : public SaberConvEltwiseMI355X<MI355X, OpDtype> {};

}  // namespace saber
}  // namespace anakin
#endif
