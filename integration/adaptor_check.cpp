// Compile check of integration/saber_mi355x_adaptor.h against the reference's own headers with
// TargetType = X86 standing in for the (not yet existing) MI355X target type. Never linked or run.
#include "anakin_config.h"
#include "saber/core/tensor.h"
#include "saber/core/context.h"
#include "saber_mi355x_adaptor.h"

using namespace anakin::saber;
template class anakin::saber::SaberConvEltwiseMI355X<X86, AK_INT8>;
template class anakin::saber::SaberConvEltwiseMI355X<X86, AK_FLOAT>;
template class anakin::saber::SaberConv2DPoolingMI355X<X86, AK_INT8>;
template class anakin::saber::SaberFcMI355X<X86, AK_INT8>;
template class anakin::saber::SaberFcMI355X<X86, AK_FLOAT>;
template class anakin::saber::SaberGemmMI355X<X86>;
