// Instantiates the LDS-DMA implicit-GEMM kernels for operand mode 0 / epilogue kind 4 (sibling pair; conv_igemm_dma.h).
#include "conv_igemm_dma.h"
namespace saber_mi355x {
hipError_t launch_igemm_dma_m0_e4(int tile, int ks, int wg, const ConvKArgs& a, hipStream_t s) {
    return launch_igemm_dma_inst<0, 4>(tile, ks, wg, a, s);
}
}  // namespace saber_mi355x
