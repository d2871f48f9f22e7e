// saber/funcs/impl/mi355x/saber_softmax.h — SaberSoftmax / VenderSoftmax<MI355X, OpDtype> (facade: saber/funcs/softmax.h)
#ifndef ANAKIN_SABER_FUNCS_IMPL_MI355X_SABER_SOFTMAX_H
#define ANAKIN_SABER_FUNCS_IMPL_MI355X_SABER_SOFTMAX_H

#include "saber/funcs/impl/impl_softmax.h"
#include "saber_mi355x_adaptor.h"

namespace anakin {
namespace saber {

template <DataType OpDtype>
class SaberSoftmax<MI355X, OpDtype> : public SaberSoftmaxMI355X<MI355X, OpDtype> {};
template <DataType OpDtype>
class VenderSoftmax<MI355X, OpDtype> : public SaberSoftmaxMI355X<MI355X, OpDtype> {};

}  // namespace saber
}  // namespace anakin
#endif
