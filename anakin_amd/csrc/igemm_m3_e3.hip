// Instantiates the implicit-GEMM kernels for operand mode 3 (FP32 on three bf16 planes) / epilogue kind 3 (see conv_igemm_impl.h).
#include "conv_igemm_impl.h"
namespace saber_mi355x {
hipError_t launch_igemm_m3_e3(int tile, int ks, const ConvKArgs& a, hipStream_t s) {
    return launch_igemm_inst<3, 3>(tile, ks, a, s);
}
}  // namespace saber_mi355x
