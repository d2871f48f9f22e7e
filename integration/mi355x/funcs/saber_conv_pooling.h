// saber/funcs/impl/mi355x/saber_conv_pooling.h — SaberConv2DPooling<MI355X, OpDtype> (facade: saber/funcs/conv_pooling.h;
// x86: saber_conv_pooling.cpp:13-160)
#ifndef ANAKIN_SABER_FUNCS_IMPL_MI355X_SABER_CONV_POOLING_H
#define ANAKIN_SABER_FUNCS_IMPL_MI355X_SABER_CONV_POOLING_H

#include "saber/funcs/impl/impl_conv_pooling.h"
#include "saber_mi355x_adaptor.h"

namespace anakin {
namespace saber {

template <DataType OpDtype>
class SaberConv2DPooling<MI355X, OpDtype> : public SaberConv2DPoolingMI355X<MI355X, OpDtype> {
public:
    SaberStatus trans_weights(Tensor<MI355X>&, Tensor<MI355X>&, int, int, int, int, int, int, int) { return SaberSuccess; }
};

}  // namespace saber
}  // namespace anakin
#endif
