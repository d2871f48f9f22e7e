// Instantiates the stem kernels for epilogue kind 3 (conv_stem.h).
#include "conv_stem.h"
namespace saber_mi355x {
hipError_t launch_stem_e3(int f32_in, const ConvKArgs& a, hipStream_t s) { return launch_conv_stem_inst<3>(f32_in, a, s); }
}  // namespace saber_mi355x
