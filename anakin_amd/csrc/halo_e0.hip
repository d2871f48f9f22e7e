// Instantiates the LDS-halo 3x3 kernels for epilogue kind 0 (conv3x3_halo.h).
#include "conv3x3_halo.h"
namespace saber_mi355x {
hipError_t launch_halo_e0(int th, const ConvKArgs& a, hipStream_t s) { return launch_conv3x3_halo_inst<0>(th, a, s); }
}  // namespace saber_mi355x
