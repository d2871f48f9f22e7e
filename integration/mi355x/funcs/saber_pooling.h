// saber/funcs/impl/mi355x/saber_pooling.h — SaberPooling<MI355X, OpDtype> (facade: saber/funcs/pooling.h:69-130,
// pattern: saber/funcs/impl/x86/saber_pooling.h). The body is the C-ABI binding in saber_mi355x_adaptor.h.
#ifndef ANAKIN_SABER_FUNCS_IMPL_MI355X_SABER_POOLING_H
#define ANAKIN_SABER_FUNCS_IMPL_MI355X_SABER_POOLING_H

#include "saber/funcs/impl/impl_pooling.h"
#include "saber_mi355x_adaptor.h"

namespace anakin {
namespace saber {

template <DataType OpDtype>
class SaberPooling<MI355X, OpDtype> : public SaberPoolingMI355X<MI355X, OpDtype> {};
template <DataType OpDtype>
class VenderPooling<MI355X, OpDtype> : public SaberPoolingMI355X<MI355X, OpDtype> {};

}  // namespace saber
}  // namespace anakin
#endif
