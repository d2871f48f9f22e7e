// oracle/ref_gemm_conv_u8s8.cpp — TEST INFRASTRUCTURE ONLY.
//
// Compiles the reference's UNMODIFIED saber/funcs/impl/x86/gemm_x8s8s32x_conv.cpp (included where it lies under
// /root/reference; this TU replaces a direct compile of that file in oracle/Makefile) and additionally instantiates
// the public member template GemmX8S8S32XConv::sub_dispatch<uint8_t, int8_t> (gemm_x8s8s32x_conv.h:72, defined at
// gemm_x8s8s32x_conv.cpp:187-288). The reference's create() computes the u8 -> s8 requantisation scale
// (gemm_x8s8s32x_conv.cpp:163-166) but its dispatch() has no branch for that dtype pair (:290-308, LOG(FATAL)
// "not support"); only the xbyak JIT path (not buildable here) runs it. Instantiating the template gives a
// reference-EXECUTED vector for the 17 u8 -> s8 convolutions of the ResNet50 INT8 list (every branch2c and
// res2a_branch1) from the reference's own arithmetic: exact s32 GEMM, float bias add, float scale multiply,
// nearbyintf, cast.
#include "saber/funcs/impl/x86/gemm_x8s8s32x_conv.cpp"

namespace anakin {
namespace saber {
template SaberStatus GemmX8S8S32XConv::sub_dispatch<uint8_t, int8_t>(const std::vector<Tensor<X86>*>&,
                                                                     std::vector<Tensor<X86>*>&,
                                                                     ConvEltwiseParam<X86>&);
}  // namespace saber
}  // namespace anakin
