// Instantiates the implicit-GEMM kernels for operand mode 0 / epilogue kind 4 (sibling pair; conv_igemm_impl.h).
#include "conv_igemm_impl.h"
namespace saber_mi355x {
hipError_t launch_igemm_m0_e4(int tile, int ks, const ConvKArgs& a, hipStream_t s) {
    return launch_igemm_inst<0, 4>(tile, ks, a, s);
}
}  // namespace saber_mi355x
