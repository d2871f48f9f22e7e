// Instantiates the fused stem conv + max pooling kernels (conv_stem.h).
#include "conv_stem.h"
namespace saber_mi355x {
hipError_t launch_conv_stem_pool(int f32_in, const ConvKArgs& a, hipStream_t s) { return launch_conv_stem_pool_inst(f32_in, a, s); }
hipError_t launch_conv_stem_pool_pair(int f32_in, const StemPairKArgs& a, hipStream_t s) { return launch_conv_stem_pool_pair_inst(f32_in, a, s); }
}  // namespace saber_mi355x
