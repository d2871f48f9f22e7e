// Instantiates the implicit-GEMM kernels for operand mode 2 / epilogue kind 3 (see conv_igemm_impl.h).
#include "conv_igemm_impl.h"
namespace saber_mi355x {
hipError_t launch_igemm_m2_e3(int tile, int ks, const ConvKArgs& a, hipStream_t s) {
    return launch_igemm_inst<2, 3>(tile, ks, a, s);
}
}  // namespace saber_mi355x
