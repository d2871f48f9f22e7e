// Instantiates the implicit-GEMM kernels for operand mode 1 / epilogue kind 2 (see conv_igemm_impl.h).
#include "conv_igemm_impl.h"
namespace saber_mi355x {
hipError_t launch_igemm_m1_e2(int tile, int ks, const ConvKArgs& a, hipStream_t s) {
    return launch_igemm_inst<1, 2>(tile, ks, a, s);
}
}  // namespace saber_mi355x
