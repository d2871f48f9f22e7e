// Instantiates the LDS-resident small-image 3x3 kernels for epilogue kind 3 (conv3x3_img.h).
#include "conv3x3_img.h"
namespace saber_mi355x {
hipError_t launch_img_e3(const ConvKArgs& a, int nw, int ib, int rb, hipStream_t s) {
    return launch_conv3x3_img_inst<3>(a, nw, ib, rb, s);
}
}  // namespace saber_mi355x
