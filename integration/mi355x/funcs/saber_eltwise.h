// saber/funcs/impl/mi355x/saber_eltwise.h — SaberEltwise<MI355X, OpDtype> (facade: saber/funcs/eltwise.h,
// pattern: saber/funcs/impl/x86/saber_eltwise.h)
#ifndef ANAKIN_SABER_FUNCS_IMPL_MI355X_SABER_ELTWISE_H
#define ANAKIN_SABER_FUNCS_IMPL_MI355X_SABER_ELTWISE_H

#include "saber/funcs/impl/impl_eltwise.h"
#include "saber_mi355x_adaptor.h"

namespace anakin {
namespace saber {

template <DataType OpDtype>
class SaberEltwise<MI355X, OpDtype> : public SaberEltwiseMI355X<MI355X, OpDtype> {};

}  // namespace saber
}  // namespace anakin
#endif
