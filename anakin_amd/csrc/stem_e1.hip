// Instantiates the stem kernels for epilogue kind 1 (conv_stem.h).
#include "conv_stem.h"
namespace saber_mi355x {
hipError_t launch_stem_e1(int f32_in, const ConvKArgs& a, hipStream_t s) { return launch_conv_stem_inst<1>(f32_in, a, s); }
}  // namespace saber_mi355x
