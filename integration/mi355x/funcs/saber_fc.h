// saber/funcs/impl/mi355x/saber_fc.h — SaberFc / VenderFc<MI355X, OpDtype> (facade: saber/funcs/fc.h:48-127; the
// framework's Dense operator asks for VENDER_IMPL on host-like targets, framework/operators/dense.cpp:61-83, and for
// SABER_IMPL elsewhere — both are the same binding here).
#ifndef ANAKIN_SABER_FUNCS_IMPL_MI355X_SABER_FC_H
#define ANAKIN_SABER_FUNCS_IMPL_MI355X_SABER_FC_H

#include "saber/funcs/impl/impl_fc.h"
#include "saber_mi355x_adaptor.h"

namespace anakin {
namespace saber {

template <DataType OpDtype>
class SaberFc<MI355X, OpDtype> : public SaberFcMI355X<MI355X, OpDtype> {};
template <DataType OpDtype>
class VenderFc<MI355X, OpDtype> : public SaberFcMI355X<MI355X, OpDtype> {};

}  // namespace saber
}  // namespace anakin
#endif
