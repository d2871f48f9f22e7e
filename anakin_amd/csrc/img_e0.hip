// Instantiates the LDS-resident small-image 3x3 kernels for epilogue kind 0 (conv3x3_img.h).
#include "conv3x3_img.h"
namespace saber_mi355x {
hipError_t launch_img_e0(const ConvKArgs& a, int nw, int ib, int rb, hipStream_t s) {
    return launch_conv3x3_img_inst<0>(a, nw, ib, rb, s);
}
bool conv3x3_img_feasible(int C, int OW, int OH, int N, int nw, int ib, int rb) { return img_shape(C, OW, OH, N, nw, ib, rb, nullptr); }
}  // namespace saber_mi355x
