// integration/saber_mi355x_adaptor.h — the reference-side binding of the MI355X Saber target.
//
// This is the code a maintainer drops into the reference as
//   saber/funcs/impl/mi355x/saber_conv.h, saber_conv_eltwise.h, saber_conv_pooling.h, saber_fc.h, saber_gemm.h
// (one `SaberXxx<MI355X, OpDtype>` partial specialisation per operator, the pattern of
// saber/funcs/impl/x86/saber_conv.h:24-69) and includes from the facade headers' target ladder
// (saber/funcs/conv.h:23-50). It only marshals Tensor / Param objects into the POD descriptors of
// include/saber_hip.h; all arithmetic lives behind the C ABI.
//
// It is written against the reference's own headers and is templated on the target type so that it can be
// COMPILE-CHECKED in this repository with TargetType = X86 (`make -C oracle adaptor_check`, run by
// __graft_entry__.build() when /root/reference is present). With the real MI355X target
// (docs/Manual/addCustomDevice.md: TargetTypeEnum, TargetWrapper<MI355X> on hipMalloc/hipMemcpyAsync/
// hipStream_t, Env/Device/Context), `TargetType` is MI355X, Tensor::data() returns device pointers and
// Context::get_compute_stream() a hipStream_t.
#ifndef SABER_MI355X_ADAPTOR_H
#define SABER_MI355X_ADAPTOR_H

#include "saber/funcs/impl/impl_base.h"
#include "saber/saber_funcs_param.h"
#include "saber_mi355x_impl.h"   // the implementations (shared with include/saber_mi355x.hpp)

#endif
