// saber/funcs/impl/mi355x/saber_activation.h — SaberActivation / VenderActivation<MI355X, OpDtype> (facade:
// saber/funcs/activation.h): the standalone ReLU operator.
#ifndef ANAKIN_SABER_FUNCS_IMPL_MI355X_SABER_ACTIVATION_H
#define ANAKIN_SABER_FUNCS_IMPL_MI355X_SABER_ACTIVATION_H

#include "saber/funcs/impl/impl_activation.h"
#include "saber_mi355x_adaptor.h"

namespace anakin {
namespace saber {

template <DataType OpDtype>
class SaberActivation<MI355X, OpDtype> : public SaberActivationMI355X<MI355X, OpDtype> {};
template <DataType OpDtype>
class VenderActivation<MI355X, OpDtype> : public SaberActivationMI355X<MI355X, OpDtype> {};

}  // namespace saber
}  // namespace anakin
#endif
