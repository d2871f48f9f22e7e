// anakin_amd/csrc/elementwise.hip — HBM-bound kernels around the conv/GEMM hot path (gfx950):
// quantise / dequantise + NCHW<->NHWC (reorder_nhwc_nchw, saber/funcs/saber_util.h:637-803),
// INT8 / FP32 eltwise sum(+relu) (impl/x86/saber_eltwise.cpp:40-113), 8-bit / FP32 pooling
// (impl/x86/saber_pooling.cpp:385-654 + kernel/jit_avx512_core_8bit_pooling_kernel.cpp:166-287),
// softmax (impl/x86/saber_softmax.cpp), and a plain FP32 GEMM (saber/funcs/gemm.h:27-66).
// All are streaming kernels: coalesced 4..16-byte accesses per lane, no LDS reuse to exploit.
#include "kernels.h"

namespace saber_mi355x {

__device__ __forceinline__ int q_sat_s8(float v) {
    v = v < -128.f ? -128.f : v;
    v = v > 127.f ? 127.f : v;
    return (int)v;
}
__device__ __forceinline__ int q_sat_u8(float v) {
    v = v < 0.f ? 0.f : v;
    v = v > 255.f ? 255.f : v;
    return (int)v;
}
// roundf() as three VALU ops; identical to roundf for |t| < 2^23 (exhaustively verified on the CPU,
// oracle orc_check_round_identity), and larger magnitudes are integers already.
__device__ __forceinline__ float round_away(float t) { return truncf(t + copysignf(0x1.fffffep-2f, t)); }

// ---- f32 NCHW -> s8/u8 NHWC(c_pad) : saturate(roundf(x * inv)) -------------------------------
__global__ __launch_bounds__(256) void quantize_nchw_to_nhwc_kernel(int n, int c, int hw, int c_pad, int u8,
                                                                    float inv, const float* __restrict__ x,
                                                                    unsigned* __restrict__ y) {
    const int groups = c_pad >> 2;  // dwords per pixel
    const size_t total = (size_t)n * hw * groups;
    for (size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x; gid < total; gid += (size_t)gridDim.x * 256) {
        // pixel fastest within a channel group -> coalesced plane reads
        const size_t per_img = (size_t)hw * groups;
        const int img = (int)(gid / per_img);
        const size_t r = gid - (size_t)img * per_img;
        const int g = (int)(r / hw);
        const int p = (int)(r - (size_t)g * hw);
        unsigned pk = 0;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int ch = g * 4 + t;
            int q = 0;
            if (ch < c) {
                const float v = round_away(__fmul_rn(x[((size_t)img * c + ch) * hw + p], inv));
                q = u8 ? q_sat_u8(v) : q_sat_s8(v);
            }
            pk |= (unsigned)(q & 0xff) << (8 * t);
        }
        y[((size_t)img * hw + p) * groups + g] = pk;
    }
}

__global__ void quantize_nchw_to_nhwc_bytes_kernel(int n, int c, int hw, int c_pad, int u8, float inv,
                                                   const float* __restrict__ x, uint8_t* __restrict__ y);

hipError_t launch_quantize_nchw_to_nhwc(int n, int c, int h, int w, int c_pad, int out_dtype, float scale,
                                        const float* x, void* y, hipStream_t s) {
    const float inv = out_dtype == DT_U8 ? 1.f / (scale * (127.f / 255.f)) : 1.f / scale;
    if (c_pad & 3) {
        const size_t tb = (size_t)n * h * w * c_pad;
        const unsigned nb = (unsigned)((tb + 255) / 256 > 8192 ? 8192 : (tb + 255) / 256);
        hipLaunchKernelGGL(quantize_nchw_to_nhwc_bytes_kernel, dim3(nb ? nb : 1), dim3(256), 0, s, n, c, h * w,
                           c_pad, out_dtype == DT_U8, inv, x, (uint8_t*)y);
        return hipGetLastError();
    }
    const size_t total = (size_t)n * h * w * (c_pad >> 2);
    const unsigned blocks = (unsigned)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
    hipLaunchKernelGGL(quantize_nchw_to_nhwc_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, s, n, c, h * w,
                       c_pad, out_dtype == DT_U8, inv, x, (unsigned*)y);
    return hipGetLastError();
}

// channel count not a multiple of 4 on the NHWC side: byte-granular variant
__global__ __launch_bounds__(256) void quantize_nchw_to_nhwc_bytes_kernel(int n, int c, int hw, int c_pad, int u8,
                                                                          float inv, const float* __restrict__ x,
                                                                          uint8_t* __restrict__ y) {
    const size_t total = (size_t)n * hw * c_pad;
    for (size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x; gid < total; gid += (size_t)gridDim.x * 256) {
        const int ch = (int)(gid % c_pad);
        const size_t pp = gid / c_pad;
        const int img = (int)(pp / hw);
        const int p = (int)(pp - (size_t)img * hw);
        int q = 0;
        if (ch < c) {
            const float v = round_away(__fmul_rn(x[((size_t)img * c + ch) * hw + p], inv));
            q = u8 ? q_sat_u8(v) : q_sat_s8(v);
        }
        y[gid] = (uint8_t)q;
    }
}

// ---- s8/u8 NHWC -> f32 NCHW : q * s ----------------------------------------------------------
__global__ __launch_bounds__(256) void dequantize_nhwc_to_nchw_kernel(int n, int c, int hw, int u8, float s,
                                                                      const uint8_t* __restrict__ x,
                                                                      float* __restrict__ y) {
    const size_t total = (size_t)n * c * hw;
    for (size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x; gid < total; gid += (size_t)gridDim.x * 256) {
        const int p = (int)(gid % hw);
        const size_t r = gid / hw;
        const int ch = (int)(r % c);
        const int img = (int)(r / c);
        const uint8_t b = x[((size_t)img * hw + p) * c + ch];
        const float q = u8 ? (float)b : (float)(int8_t)b;
        y[gid] = __fmul_rn(q, s);
    }
}

hipError_t launch_dequantize_nhwc_to_nchw(int n, int c, int h, int w, int in_dtype, float scale, const void* x,
                                          float* y, hipStream_t s) {
    const float sc = in_dtype == DT_U8 ? scale * (127.f / 255.f) : scale;
    const size_t total = (size_t)n * c * h * w;
    const unsigned blocks = (unsigned)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
    hipLaunchKernelGGL(dequantize_nhwc_to_nchw_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, s, n, c, h * w,
                       in_dtype == DT_U8, sc, (const uint8_t*)x, y);
    return hipGetLastError();
}

// ---- f32 layout transforms -------------------------------------------------------------------
__global__ __launch_bounds__(256) void nchw_to_nhwc_f32_kernel(int n, int c, int hw, int c_pad,
                                                               const float* __restrict__ x, float* __restrict__ y) {
    const size_t total = (size_t)n * hw * c_pad;
    for (size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x; gid < total; gid += (size_t)gridDim.x * 256) {
        // thread order: pixel fastest within channel (coalesced reads)
        const size_t per_img = (size_t)hw * c_pad;
        const int img = (int)(gid / per_img);
        const size_t r = gid - (size_t)img * per_img;
        const int ch = (int)(r / hw);
        const int p = (int)(r - (size_t)ch * hw);
        y[((size_t)img * hw + p) * c_pad + ch] = ch < c ? x[((size_t)img * c + ch) * hw + p] : 0.f;
    }
}
__global__ __launch_bounds__(256) void nhwc_to_nchw_f32_kernel(int n, int c, int hw, int c_pad,
                                                               const float* __restrict__ x, float* __restrict__ y) {
    const size_t total = (size_t)n * c * hw;
    for (size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x; gid < total; gid += (size_t)gridDim.x * 256) {
        const int p = (int)(gid % hw);
        const size_t r = gid / hw;
        const int ch = (int)(r % c);
        const int img = (int)(r / c);
        y[gid] = x[((size_t)img * hw + p) * c_pad + ch];
    }
}
static unsigned grid_for(size_t total) {
    size_t b = (total + 255) / 256;
    if (b > 8192) b = 8192;
    if (b == 0) b = 1;
    return (unsigned)b;
}
hipError_t launch_transpose_nchw_to_nhwc_f32(int n, int c, int h, int w, int c_pad, const float* x, float* y,
                                             hipStream_t s) {
    hipLaunchKernelGGL(nchw_to_nhwc_f32_kernel, dim3(grid_for((size_t)n * h * w * c_pad)), dim3(256), 0, s, n, c,
                       h * w, c_pad, x, y);
    return hipGetLastError();
}
hipError_t launch_transpose_nhwc_to_nchw_f32(int n, int c, int h, int w, int c_pad, const float* x, float* y,
                                             hipStream_t s) {
    hipLaunchKernelGGL(nhwc_to_nchw_f32_kernel, dim3(grid_for((size_t)n * c * h * w)), dim3(256), 0, s, n, c,
                       h * w, c_pad, x, y);
    return hipGetLastError();
}

// ---- 8-bit NHWC channel padding (first-layer path: C=3 -> 4) ---------------------------------
__global__ __launch_bounds__(256) void pad_channels_i8_kernel(size_t pixels, int c, int c_pad,
                                                              const uint8_t* __restrict__ x, uint8_t* __restrict__ y) {
    const size_t total = pixels * c_pad;
    for (size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x; gid < total; gid += (size_t)gridDim.x * 256) {
        const int ch = (int)(gid % c_pad);
        const size_t p = gid / c_pad;
        y[gid] = ch < c ? x[p * c + ch] : 0;
    }
}
hipError_t launch_pad_channels_i8(size_t pixels, int c, int c_pad, const void* x, void* y, hipStream_t s) {
    hipLaunchKernelGGL(pad_channels_i8_kernel, dim3(grid_for(pixels * c_pad)), dim3(256), 0, s, pixels, c, c_pad,
                       (const uint8_t*)x, (uint8_t*)y);
    return hipGetLastError();
}

// ---- byte transpose [rows][cols] -> [cols][rows_pad] through a 32x33 LDS tile (coalesced both ways) ----------------
__global__ __launch_bounds__(256) void transpose_bytes_kernel(int rows, int cols, int rows_pad, const uint8_t* __restrict__ x,
                                                              uint8_t* __restrict__ y) {
    __shared__ uint8_t tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
#pragma unroll
    for (int i = 0; i < 32; i += 8) {
        const int r = r0 + ty + i, c = c0 + tx;
        tile[ty + i][tx] = (r < rows && c < cols) ? x[(size_t)r * cols + c] : 0;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 32; i += 8) {
        const int c = c0 + ty + i, r = r0 + tx;
        if (c < cols && r < rows_pad) y[(size_t)c * rows_pad + r] = tile[tx][ty + i];
    }
}
hipError_t launch_transpose_bytes(int rows, int cols, int rows_pad, const void* x, void* y, hipStream_t s) {
    hipLaunchKernelGGL(transpose_bytes_kernel, dim3((cols + 31) / 32, (rows_pad + 31) / 32), dim3(256), 0, s, rows, cols,
                       rows_pad, (const uint8_t*)x, (uint8_t*)y);
    return hipGetLastError();
}

// ---- flat f32 -> s8, ScaleUtils::scale_fp32_int8 (x86_utils.h:325-346) ------------------------
__global__ __launch_bounds__(256) void quantize_flat_s8_kernel(size_t count, float inv, const float* __restrict__ x,
                                                               int8_t* __restrict__ y) {
    for (size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x; gid < count; gid += (size_t)gridDim.x * 256) {
        int t = (int)round_away(__fmul_rn(x[gid], inv));
        t = t > 127 ? 127 : t;
        t = t < -128 ? -128 : t;
        y[gid] = (int8_t)t;
    }
}
hipError_t launch_quantize_flat_s8(size_t count, float scale, const float* x, int8_t* y, hipStream_t s) {
    hipLaunchKernelGGL(quantize_flat_s8_kernel, dim3(grid_for(count)), dim3(256), 0, s, count, 1.f / scale, x, y);
    return hipGetLastError();
}

// ---- eltwise sum ------------------------------------------------------------------------------
__device__ __forceinline__ int elt_i8_one(int a, int b, float sa, float sb, float c0, float c1, int relu) {
    float t = __fmul_rn(__fmul_rn(c0, (float)a), sa);
    t = __fadd_rn(t, __fmul_rn(__fmul_rn(c1, (float)b), sb));
    if (relu) t = t > 0.f ? t : 0.f;
    return q_sat_s8(round_away(t));
}
__global__ __launch_bounds__(256) void eltwise_sum_i8_kernel(size_t count, const int8_t* __restrict__ a,
                                                             const int8_t* __restrict__ b, float sa, float sb,
                                                             float c0, float c1, int relu, int8_t* __restrict__ y) {
    const size_t vec = count >> 4;  // 16 bytes per lane
    for (size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x; gid < vec; gid += (size_t)gridDim.x * 256) {
        const uint4 va = ((const uint4*)a)[gid];
        const uint4 vb = ((const uint4*)b)[gid];
        const unsigned wa[4] = {va.x, va.y, va.z, va.w};
        const unsigned wb[4] = {vb.x, vb.y, vb.z, vb.w};
        unsigned wo[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            unsigned o = 0;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int ai = (int)(int8_t)(wa[d] >> (8 * t));
                const int bi = (int)(int8_t)(wb[d] >> (8 * t));
                o |= (unsigned)(elt_i8_one(ai, bi, sa, sb, c0, c1, relu) & 0xff) << (8 * t);
            }
            wo[d] = o;
        }
        ((uint4*)y)[gid] = make_uint4(wo[0], wo[1], wo[2], wo[3]);
    }
    // tail
    const size_t base = vec << 4;
    for (size_t i = base + (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256) {
        y[i] = (int8_t)elt_i8_one(a[i], b[i], sa, sb, c0, c1, relu);
    }
}
hipError_t launch_eltwise_sum_i8(size_t count, const int8_t* a, const int8_t* b, float sa, float sb, float c0,
                                 float c1, int relu, int8_t* y, hipStream_t s) {
    hipLaunchKernelGGL(eltwise_sum_i8_kernel, dim3(grid_for((count + 15) / 16)), dim3(256), 0, s, count, a, b, sa,
                       sb, c0, c1, relu, y);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void eltwise_sum_f32_kernel(size_t count, const float* __restrict__ a,
                                                              const float* __restrict__ b, float c0, float c1,
                                                              int relu, float* __restrict__ y) {
    const size_t vec = count >> 2;
    for (size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x; gid < vec; gid += (size_t)gridDim.x * 256) {
        const float4 va = ((const float4*)a)[gid];
        const float4 vb = ((const float4*)b)[gid];
        float o[4] = {__fadd_rn(__fmul_rn(c0, va.x), __fmul_rn(c1, vb.x)),
                      __fadd_rn(__fmul_rn(c0, va.y), __fmul_rn(c1, vb.y)),
                      __fadd_rn(__fmul_rn(c0, va.z), __fmul_rn(c1, vb.z)),
                      __fadd_rn(__fmul_rn(c0, va.w), __fmul_rn(c1, vb.w))};
        if (relu) {
#pragma unroll
            for (int t = 0; t < 4; ++t) o[t] = o[t] > 0.f ? o[t] : 0.f;
        }
        ((float4*)y)[gid] = make_float4(o[0], o[1], o[2], o[3]);
    }
    const size_t base = vec << 2;
    for (size_t i = base + (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256) {
        float t = __fadd_rn(__fmul_rn(c0, a[i]), __fmul_rn(c1, b[i]));
        y[i] = relu ? (t > 0.f ? t : 0.f) : t;
    }
}
hipError_t launch_eltwise_sum_f32(size_t count, const float* a, const float* b, float c0, float c1, int relu,
                                  float* y, hipStream_t s) {
    hipLaunchKernelGGL(eltwise_sum_f32_kernel, dim3(grid_for((count + 3) / 4)), dim3(256), 0, s, count, a, b, c0,
                       c1, relu, y);
    return hipGetLastError();
}

// ---- standalone ReLU (SaberActivation<X86,AK_FLOAT>, Active_relu: saber_activation.cpp:136-154) ---------------
__global__ __launch_bounds__(256) void relu_f32_kernel(size_t count, const float* __restrict__ x, float* __restrict__ y) {
    const size_t vec = count >> 2;
    for (size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x; gid < vec; gid += (size_t)gridDim.x * 256) {
        const float4 v = ((const float4*)x)[gid];
        ((float4*)y)[gid] = make_float4(v.x > 0.f ? v.x : 0.f, v.y > 0.f ? v.y : 0.f, v.z > 0.f ? v.z : 0.f,
                                        v.w > 0.f ? v.w : 0.f);
    }
    for (size_t i = (vec << 2) + (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256)
        y[i] = x[i] > 0.f ? x[i] : 0.f;
}
hipError_t launch_relu_f32(size_t count, const float* x, float* y, hipStream_t s) {
    hipLaunchKernelGGL(relu_f32_kernel, dim3(grid_for((count + 3) / 4)), dim3(256), 0, s, count, x, y);
    return hipGetLastError();
}

// ---- the other activation types of SaberActivation<X86, AK_FLOAT> (saber_activation.cpp:156-262; the scalar formulas of
// test/saber/test_saber_activation.cpp:17-115): elementwise, HBM-bound, in place allowed. `active` = the reference's ActiveType value.
template <int ACTIVE>
__device__ __forceinline__ float act_f32(float v, float slope, float coef) {
    if constexpr (ACTIVE == 1) return 1.0f / (1.0f + expf(-v));                         // sigmoid
    else if constexpr (ACTIVE == 3) return tanhf(v);                                    // tanh
    else if constexpr (ACTIVE == 4) { const float r = v > 0.f ? v : 0.f; return r < coef ? r : coef; }   // clipped relu
    else if constexpr (ACTIVE == 5) return v > 0.f ? v : coef * (expf(v) - 1.f);        // elu
    else if constexpr (ACTIVE == 9) return coef * tanhf(slope * v);                     // stanh
    else if constexpr (ACTIVE == 11) return v * (0.5f * (erff(v / sqrtf(2.f)) + 1.f));  // gelu
    else return v / (1.0f + expf(-v * coef));                                           // swish (12)
}
template <int ACTIVE>
__global__ __launch_bounds__(256) void activation_f32_kernel(size_t count, float slope, float coef, const float* __restrict__ x,
                                                             float* __restrict__ y) {
    const size_t vec = count >> 2;
    for (size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x; gid < vec; gid += (size_t)gridDim.x * 256) {
        const float4 v = ((const float4*)x)[gid];
        ((float4*)y)[gid] = make_float4(act_f32<ACTIVE>(v.x, slope, coef), act_f32<ACTIVE>(v.y, slope, coef),
                                        act_f32<ACTIVE>(v.z, slope, coef), act_f32<ACTIVE>(v.w, slope, coef));
    }
    for (size_t i = (vec << 2) + (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256)
        y[i] = act_f32<ACTIVE>(x[i], slope, coef);
}
hipError_t launch_activation_f32(int active, size_t count, float slope, float coef, const float* x, float* y, hipStream_t s) {
    const dim3 grid(grid_for((count + 3) / 4)), block(256);
    switch (active) {
    case 1: hipLaunchKernelGGL(activation_f32_kernel<1>, grid, block, 0, s, count, slope, coef, x, y); break;
    case 3: hipLaunchKernelGGL(activation_f32_kernel<3>, grid, block, 0, s, count, slope, coef, x, y); break;
    case 4: hipLaunchKernelGGL(activation_f32_kernel<4>, grid, block, 0, s, count, slope, coef, x, y); break;
    case 5: hipLaunchKernelGGL(activation_f32_kernel<5>, grid, block, 0, s, count, slope, coef, x, y); break;
    case 9: hipLaunchKernelGGL(activation_f32_kernel<9>, grid, block, 0, s, count, slope, coef, x, y); break;
    case 11: hipLaunchKernelGGL(activation_f32_kernel<11>, grid, block, 0, s, count, slope, coef, x, y); break;
    case 12: hipLaunchKernelGGL(activation_f32_kernel<12>, grid, block, 0, s, count, slope, coef, x, y); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
// PReLU (excute_prelu, saber_activation.cpp:38-132): y = x > 0 ? x : x * slope[channel]; channel = (i / inner) % channels
// (NCHW: inner = H * W, NHWC: inner = 1); channel_shared: slope[0]
__global__ __launch_bounds__(256) void prelu_f32_kernel(size_t count, int channels, int inner, int shared, const float* __restrict__ slope,
                                                        const float* __restrict__ x, float* __restrict__ y) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256) {
        const float v = x[i];
        const float sl = shared ? slope[0] : slope[(i / (size_t)inner) % (size_t)channels];
        y[i] = v > 0.f ? v : v * sl;
    }
}
hipError_t launch_prelu_f32(size_t count, int channels, int inner, int shared, const float* slope, const float* x, float* y, hipStream_t s) {
    hipLaunchKernelGGL(prelu_f32_kernel, dim3(grid_for(count)), dim3(256), 0, s, count, channels, inner, shared, slope, x, y);
    return hipGetLastError();
}

// ---- 8-bit NHWC pooling -----------------------------------------------------------------------
// One lane per (output pixel, 4-channel dword). JIT semantics: int32 window sum, (float)sum * idivider,
// round-to-nearest-even, saturate; max by signed/unsigned compare.
__global__ __launch_bounds__(256) void pool2d_i8_nhwc_kernel(int n, int h, int w, int c, int oh, int ow, int kh,
                                                             int kw, int sh, int sw, int ph, int pw, int type,
                                                             int in_u8, int out_dtype,
                                                             const uint8_t* __restrict__ x, void* __restrict__ y) {
    const int cg = (c + 3) >> 2;
    const size_t total = (size_t)n * oh * ow * cg;
    for (size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x; gid < total; gid += (size_t)gridDim.x * 256) {
        const int g = (int)(gid % cg);
        size_t r = gid / cg;
        const int ox = (int)(r % ow);
        r /= ow;
        const int oy = (int)(r % oh);
        const int img = (int)(r / oh);
        int hs = oy * sh - ph, ws = ox * sw - pw;
        int he = hs + kh, we = ws + kw;
        hs = hs < 0 ? 0 : hs;
        ws = ws < 0 ? 0 : ws;
        he = he > h ? h : he;
        we = we > w ? w : we;
        const int nch = (c - g * 4) < 4 ? (c - g * 4) : 4;
        int sum[4] = {0, 0, 0, 0};
        int mx[4];
        for (int t = 0; t < 4; ++t) mx[t] = in_u8 ? 0 : -128;
        for (int iy = hs; iy < he; ++iy) {
            for (int ix = ws; ix < we; ++ix) {
                const uint8_t* px = x + (((size_t)img * h + iy) * w + ix) * c + g * 4;
                unsigned v = 0;
                if (nch == 4 && (c & 3) == 0) {
                    v = *(const unsigned*)px;
                } else {
                    for (int t = 0; t < nch; ++t) v |= (unsigned)px[t] << (8 * t);
                }
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int b = (v >> (8 * t)) & 0xff;
                    const int val = in_u8 ? b : (int)(int8_t)b;
                    sum[t] += val;
                    mx[t] = val > mx[t] ? val : mx[t];
                }
            }
        }
        const float idiv = 1.0f / (float)(type == 2 ? (he - hs) * (we - ws) : kh * kw);
        const size_t o = (((size_t)img * oh + oy) * ow + ox) * c + g * 4;
        int q[4];
        float f[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (type == 0) {
                q[t] = mx[t];
            } else {
                f[t] = __fmul_rn((float)sum[t], idiv);
                q[t] = out_dtype == DT_U8 ? q_sat_u8(rintf(f[t])) : q_sat_s8(rintf(f[t]));
            }
        }
        if (out_dtype == DT_F32) {
            for (int t = 0; t < nch; ++t) ((float*)y)[o + t] = f[t];
        } else if (nch == 4 && (c & 3) == 0) {
            *(unsigned*)((uint8_t*)y + o) = (q[0] & 0xff) | ((q[1] & 0xff) << 8) | ((q[2] & 0xff) << 16) |
                                            ((unsigned)(q[3] & 0xff) << 24);
        } else {
            for (int t = 0; t < nch; ++t) ((uint8_t*)y)[o + t] = (uint8_t)q[t];
        }
    }
}

// Max pooling, channel count a multiple of 16: one lane per (output pixel, 16 channels), 16-byte loads,
// byte-wise max as packed 16-bit maxima of the even / odd bytes (s8 is mapped to u8 order by XOR 0x80).
typedef unsigned short us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pk_max_u16(unsigned a, unsigned b) {
    const us2 r = __builtin_elementwise_max(__builtin_bit_cast(us2, a), __builtin_bit_cast(us2, b));
    return __builtin_bit_cast(unsigned, r);
}
__global__ __launch_bounds__(256) void maxpool_i8_nhwc_vec16_kernel(int n, int h, int w, int c, int oh, int ow, int kh,
                                                                    int kw, int sh, int sw, int ph, int pw, int in_u8,
                                                                    const uint4* __restrict__ x, uint4* __restrict__ y) {
    const int cg = c >> 4;
    const size_t total = (size_t)n * oh * ow * cg;
    const unsigned flip = in_u8 ? 0u : 0x80808080u;
    for (size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x; gid < total; gid += (size_t)gridDim.x * 256) {
        const int g = (int)(gid % cg);
        size_t r = gid / cg;
        const int ox = (int)(r % ow);
        r /= ow;
        const int oy = (int)(r % oh);
        const int img = (int)(r / oh);
        int hs = oy * sh - ph, ws = ox * sw - pw;
        int he = hs + kh, we = ws + kw;
        hs = hs < 0 ? 0 : hs;
        ws = ws < 0 ? 0 : ws;
        he = he > h ? h : he;
        we = we > w ? w : we;
        unsigned ev[4] = {0, 0, 0, 0}, od[4] = {0, 0, 0, 0};   // running maxima of even / odd bytes (u8 order)
        for (int iy = hs; iy < he; ++iy)
            for (int ix = ws; ix < we; ++ix) {
                const uint4 v = x[(((size_t)img * h + iy) * w + ix) * cg + g];
                const unsigned d[4] = {v.x ^ flip, v.y ^ flip, v.z ^ flip, v.w ^ flip};
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    ev[t] = pk_max_u16(ev[t], d[t] & 0x00ff00ffu);
                    od[t] = pk_max_u16(od[t], (d[t] >> 8) & 0x00ff00ffu);
                }
            }
        uint4 o;
        o.x = (ev[0] | (od[0] << 8)) ^ flip;
        o.y = (ev[1] | (od[1] << 8)) ^ flip;
        o.z = (ev[2] | (od[2] << 8)) ^ flip;
        o.w = (ev[3] | (od[3] << 8)) ^ flip;
        y[(((size_t)img * oh + oy) * ow + ox) * cg + g] = o;
    }
}
// INT8 global average pooling (the tail of the graph the reference's optimiser emits: pool5 as an AK_INT8 op, 7x7 -> 1x1):
// a workgroup owns 128 channels of one image; 8 pixel groups x 32 channel quads sum their share of the h*w pixels in
// int32 (exact, order independent), the partials meet in LDS and 32 lanes finish with the JIT kernel's arithmetic
// ((float)sum * (1 / count), round to nearest even, saturate). All of a lane's loads are independent: the generic
// kernel above walks the 49 pixels of a window serially in one lane (13.9 us at batch 8; this: one latency).
__global__ __launch_bounds__(256) void gpool_i8_nhwc_kernel(int hw, int c, float idiv, int in_u8, int out_dtype,
                                                            const uint8_t* __restrict__ x, void* __restrict__ y) {
    __shared__ int part[8][32][4];
    const int img = blockIdx.y, cq = threadIdx.x & 31, pg = threadIdx.x >> 5;
    const int ch = blockIdx.x * 128 + cq * 4;
    int sum[4] = {0, 0, 0, 0};
    if (ch < c) {
        const uint8_t* base = x + (size_t)img * hw * c + ch;
        for (int px = pg; px < hw; px += 8) {
            const unsigned v = *(const unsigned*)(base + (size_t)px * c);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int b = (v >> (8 * t)) & 0xff;
                sum[t] += in_u8 ? b : (int)(int8_t)b;
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) part[pg][cq][t] = sum[t];
    __syncthreads();
    if (pg == 0 && ch < c) {
        int q[4];
        float f[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            int tot = 0;
#pragma unroll
            for (int g = 0; g < 8; ++g) tot += part[g][cq][t];
            f[t] = __fmul_rn((float)tot, idiv);
            q[t] = out_dtype == DT_U8 ? q_sat_u8(rintf(f[t])) : q_sat_s8(rintf(f[t]));
        }
        const size_t o = (size_t)img * c + ch;
        if (out_dtype == DT_F32) {
            *(float4*)((float*)y + o) = make_float4(f[0], f[1], f[2], f[3]);
        } else {
            *(unsigned*)((uint8_t*)y + o) = (q[0] & 0xff) | ((q[1] & 0xff) << 8) | ((q[2] & 0xff) << 16) |
                                            ((unsigned)(q[3] & 0xff) << 24);
        }
    }
}

hipError_t launch_pool2d_i8_nhwc(int n, int h, int w, int c, int oh, int ow, int kh, int kw, int sh, int sw,
                                 int ph, int pw, int type, int in_dtype, int out_dtype, const void* x, void* y,
                                 hipStream_t s) {
    if (type == 0 && out_dtype == DT_F32) return hipErrorInvalidValue;  // as the reference (pooling kernel :425)
    if (type != 0 && oh == 1 && ow == 1 && kh == h && kw == w && ph == 0 && pw == 0 && (c & 3) == 0 && n <= 65535) {
        hipLaunchKernelGGL(gpool_i8_nhwc_kernel, dim3((c + 127) / 128, n), dim3(256), 0, s, h * w, c,
                           1.0f / (float)(kh * kw), in_dtype == DT_U8, out_dtype, (const uint8_t*)x, y);
        return hipGetLastError();
    }
    if (type == 0 && (c & 15) == 0 && in_dtype == out_dtype) {
        hipLaunchKernelGGL(maxpool_i8_nhwc_vec16_kernel, dim3(grid_for((size_t)n * oh * ow * (c >> 4))), dim3(256), 0, s,
                           n, h, w, c, oh, ow, kh, kw, sh, sw, ph, pw, in_dtype == DT_U8, (const uint4*)x, (uint4*)y);
        return hipGetLastError();
    }
    const size_t total = (size_t)n * oh * ow * ((c + 3) >> 2);
    hipLaunchKernelGGL(pool2d_i8_nhwc_kernel, dim3(grid_for(total)), dim3(256), 0, s, n, h, w, c, oh, ow, kh, kw,
                       sh, sw, ph, pw, type, in_dtype == DT_U8, out_dtype, (const uint8_t*)x, y);
    return hipGetLastError();
}

// ---- FP32 pooling (NHWC or NCHW) ---------------------------------------------------------------
__global__ __launch_bounds__(256) void pool2d_f32_kernel(int n, int h, int w, int c, int oh, int ow, int kh, int kw,
                                                         int sh, int sw, int ph, int pw, int type, int nchw,
                                                         const float* __restrict__ x, float* __restrict__ y) {
    const size_t total = (size_t)n * oh * ow * c;
    for (size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x; gid < total; gid += (size_t)gridDim.x * 256) {
        int ch, ox, oy, img;
        size_t r = gid;
        if (nchw) {
            ox = (int)(r % ow); r /= ow;
            oy = (int)(r % oh); r /= oh;
            ch = (int)(r % c);
            img = (int)(r / c);
        } else {
            ch = (int)(r % c); r /= c;
            ox = (int)(r % ow); r /= ow;
            oy = (int)(r % oh);
            img = (int)(r / oh);
        }
        int hs = oy * sh - ph, ws = ox * sw - pw;
        int he = hs + kh, we = ws + kw;
        hs = hs < 0 ? 0 : hs;
        ws = ws < 0 ? 0 : ws;
        he = he > h ? h : he;
        we = we > w ? w : we;
        float acc = 0.f;
        bool first = true;
        for (int iy = hs; iy < he; ++iy)
            for (int ix = ws; ix < we; ++ix) {
                const float v = nchw ? x[(((size_t)img * c + ch) * h + iy) * w + ix]
                                     : x[(((size_t)img * h + iy) * w + ix) * c + ch];
                if (type == 0) {
                    acc = first ? v : (acc >= v ? acc : v);
                    first = false;
                } else {
                    acc = __fadd_rn(acc, v);
                }
            }
        if (type == 1) {  // divisor clipped at in+pad on the far edge (saber_pooling.cpp:466-480)
            int bh = kh, bw = kw;
            if (we == w) bw = (ws + kw >= w + pw ? w + pw : ws + kw) - ws;
            if (he == h) bh = (hs + kh >= h + ph ? h + ph : hs + kh) - hs;
            acc = acc / (float)(bh * bw);
        }
        if (type == 2) acc = acc / (float)((he - hs) * (we - ws));
        y[gid] = acc;
    }
}
// NHWC with c % 4 == 0: one lane per (output pixel, 4 channels), 16-byte loads; per channel the same window order
// and the same float operations as the scalar kernel above.
__global__ __launch_bounds__(256) void pool2d_f32_nhwc_vec4_kernel(int n, int h, int w, int c, int oh, int ow, int kh,
                                                                   int kw, int sh, int sw, int ph, int pw, int type,
                                                                   const float4* __restrict__ x, float4* __restrict__ y) {
    const int cg = c >> 2;
    const size_t total = (size_t)n * oh * ow * cg;
    for (size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x; gid < total; gid += (size_t)gridDim.x * 256) {
        size_t r = gid;
        const int g = (int)(r % cg); r /= cg;
        const int ox = (int)(r % ow); r /= ow;
        const int oy = (int)(r % oh);
        const int img = (int)(r / oh);
        int hs = oy * sh - ph, ws = ox * sw - pw;
        int he = hs + kh, we = ws + kw;
        hs = hs < 0 ? 0 : hs;
        ws = ws < 0 ? 0 : ws;
        he = he > h ? h : he;
        we = we > w ? w : we;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        bool first = true;
        for (int iy = hs; iy < he; ++iy)
            for (int ix = ws; ix < we; ++ix) {
                const float4 q = x[(((size_t)img * h + iy) * w + ix) * cg + g];
                const float v[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    if (type == 0) acc[t] = first ? v[t] : (acc[t] >= v[t] ? acc[t] : v[t]);
                    else acc[t] = __fadd_rn(acc[t], v[t]);
                }
                first = false;
            }
        float div = 1.f;
        if (type == 1) {  // divisor clipped at in+pad on the far edge (saber_pooling.cpp:466-480)
            int bh = kh, bw = kw;
            if (we == w) bw = (ws + kw >= w + pw ? w + pw : ws + kw) - ws;
            if (he == h) bh = (hs + kh >= h + ph ? h + ph : hs + kh) - hs;
            div = (float)(bh * bw);
        }
        if (type == 2) div = (float)((he - hs) * (we - ws));
        if (type != 0) {
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = acc[t] / div;
        }
        y[gid] = make_float4(acc[0], acc[1], acc[2], acc[3]);
    }
}
// Global pooling of an NHWC tensor with <= 64 pixels per image (ResNet's 7x7 pool5): one lane per (image, channel), ALL of
// its pixel loads in flight together, then the reference's accumulation order (pixel 0, 1, 2 ... - the generic kernel's
// window loop) - same bits, one memory latency instead of one per pixel (17.8 -> ~5 us for [8, 7, 7, 2048]).
__global__ __launch_bounds__(256) void gpool_f32_nhwc_kernel(int n, int hw, int c, int type, const float* __restrict__ x,
                                                             float* __restrict__ y) {
    const int gid = blockIdx.x * 256 + threadIdx.x;
    if (gid >= n * c) return;
    const int img = gid / c, ch = gid - img * c;
    const float* p = x + (size_t)img * hw * c + ch;
    float v[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) v[i] = i < hw ? p[(size_t)i * c] : 0.f;
    float acc = v[0];
    if (type == 0) {
#pragma unroll
        for (int i = 1; i < 64; ++i)
            if (i < hw) acc = acc >= v[i] ? acc : v[i];
    } else {
        acc = __fadd_rn(0.f, v[0]);
#pragma unroll
        for (int i = 1; i < 64; ++i)
            if (i < hw) acc = __fadd_rn(acc, v[i]);
        acc = acc / (float)hw;
    }
    y[gid] = acc;
}
hipError_t launch_pool2d_f32(int n, int h, int w, int c, int oh, int ow, int kh, int kw, int sh, int sw, int ph,
                             int pw, int type, int nchw, const float* x, float* y, hipStream_t s) {
    if (!nchw && oh == 1 && ow == 1 && ph == 0 && pw == 0 && kh == h && kw == w && h * w <= 64 && (size_t)n * c < (1u << 30)) {
        hipLaunchKernelGGL(gpool_f32_nhwc_kernel, dim3((n * c + 255) / 256), dim3(256), 0, s, n, h * w, c, type, x, y);
        return hipGetLastError();
    }
    if (!nchw && (c & 3) == 0) {
        hipLaunchKernelGGL(pool2d_f32_nhwc_vec4_kernel, dim3(grid_for((size_t)n * oh * ow * (c >> 2))), dim3(256), 0, s,
                           n, h, w, c, oh, ow, kh, kw, sh, sw, ph, pw, type, (const float4*)x, (float4*)y);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(pool2d_f32_kernel, dim3(grid_for((size_t)n * oh * ow * c)), dim3(256), 0, s, n, h, w, c, oh,
                       ow, kh, kw, sh, sw, ph, pw, type, nchw, x, y);
    return hipGetLastError();
}


// ---- FP32 pooling fed an 8-bit NHWC tensor: SaberPooling<X86,AK_FLOAT>::dispatch dequantises on entry
// (reorder_nhwc_nchw, saber_pooling.cpp:399-402) and pools in NCHW; fused here, same float sequence:
// v = (float)q * s ; window accumulated in (h, w) order ; / area. Output NCHW f32.
__global__ __launch_bounds__(256) void pool2d_f32_from_i8_kernel(int n, int h, int w, int c, int oh, int ow, int kh,
                                                                 int kw, int sh, int sw, int ph, int pw, int type,
                                                                 int in_u8, float s, const uint8_t* __restrict__ x,
                                                                 float* __restrict__ y) {
    const size_t total = (size_t)n * oh * ow * c;
    for (size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x; gid < total; gid += (size_t)gridDim.x * 256) {
        size_t r = gid;
        const int ch = (int)(r % c); r /= c;      // channel fastest: coalesced NHWC reads
        const int ox = (int)(r % ow); r /= ow;
        const int oy = (int)(r % oh);
        const int img = (int)(r / oh);
        int hs = oy * sh - ph, ws = ox * sw - pw;
        int he = hs + kh, we = ws + kw;
        hs = hs < 0 ? 0 : hs;
        ws = ws < 0 ? 0 : ws;
        he = he > h ? h : he;
        we = we > w ? w : we;
        float acc = 0.f;
        bool first = true;
        for (int iy = hs; iy < he; ++iy)
            for (int ix = ws; ix < we; ++ix) {
                const uint8_t b = x[(((size_t)img * h + iy) * w + ix) * c + ch];
                const float v = __fmul_rn(in_u8 ? (float)b : (float)(int8_t)b, s);
                if (type == 0) {
                    acc = first ? v : (acc >= v ? acc : v);
                    first = false;
                } else {
                    acc = __fadd_rn(acc, v);
                }
            }
        if (type == 1) {
            int bh = kh, bw = kw;
            if (we == w) bw = (ws + kw >= w + pw ? w + pw : ws + kw) - ws;
            if (he == h) bh = (hs + kh >= h + ph ? h + ph : hs + kh) - hs;
            acc = acc / (float)(bh * bw);
        }
        if (type == 2) acc = acc / (float)((he - hs) * (we - ws));
        y[(((size_t)img * c + ch) * oh + oy) * ow + ox] = acc;
    }
}

// Global pool specialisation of the kernel above (window = whole image of <= 64 pixels, no padding). The float
// accumulation order of the reference (h, w order, one running sum per channel) is a serial chain and has to stay, but
// the memory side does not: a workgroup owns one image x 128 channels, its 256 threads fetch the 49 (7x7) pixel rows
// 8 at a time into LDS (every load of the image in flight at once: one exposed memory latency), then 128 threads walk
// one channel each through LDS in (h, w) order. 8 x 16 workgroups at batch 8 (the round-1 kernel: one lane per 4
// channels serially fetching 49 strided dwords, 64 workgroups of one wave, 8 us).
// yq (optional): the s8 quantisation of the result with scale 1/qinv - the quantise-on-entry of a following
// INT8 op (scale_fp32_int8, x86_utils.h:325-346) fused into the store.
__global__ __launch_bounds__(256) void gpool_f32_from_i8_kernel(int n, int hw, int c, int type, int in_u8, float s,
                                                               const uint8_t* __restrict__ x, float* __restrict__ y,
                                                               float qinv, int8_t* __restrict__ yq) {
    __shared__ unsigned raw[64][32];
    const int cgs = c >> 2;                          // dwords per pixel
    const int nblk = (cgs + 31) >> 5;
    const int img = blockIdx.x / nblk, cb = blockIdx.x - img * nblk;
    {
        const int l = threadIdx.x & 31, slot = threadIdx.x >> 5;
        const int g = cb * 32 + l;
        const unsigned* px = (const unsigned*)(x + (size_t)img * hw * c);
        if (g < cgs)
            for (int p = slot; p < hw; p += 8) raw[p][l] = px[(size_t)p * cgs + g];
    }
    __syncthreads();
    const int ch = threadIdx.x;                      // channel within the 128-channel window
    if (ch >= 128 || cb * 128 + ch >= c) return;
    float acc = 0.f;
    for (int p = 0; p < hw; ++p) {
        const int q = (raw[p][ch >> 2] >> (8 * (ch & 3))) & 0xff;
        const float f = __fmul_rn(in_u8 ? (float)q : (float)(int8_t)q, s);
        if (type == 0) acc = p == 0 ? f : (acc >= f ? acc : f);
        else acc = __fadd_rn(acc, f);
    }
    const float r = type == 0 ? acc : acc / (float)hw;
    const size_t o = (size_t)img * c + cb * 128 + ch;
    y[o] = r;
    if (yq) {
        int t = (int)round_away(__fmul_rn(r, qinv));
        t = t > 127 ? 127 : t;
        t = t < -128 ? -128 : t;
        yq[o] = (int8_t)t;
    }
}
hipError_t launch_pool2d_f32_from_i8(int n, int h, int w, int c, int oh, int ow, int kh, int kw, int sh, int sw, int ph,
                                     int pw, int type, int in_dtype, float scale, const void* x, float* y,
                                     float q_scale, int8_t* yq, hipStream_t s) {
    const float sc = in_dtype == DT_U8 ? scale * (127.f / 255.f) : scale;
    if (oh == 1 && ow == 1 && kh == h && kw == w && ph == 0 && pw == 0 && (c & 3) == 0 && h * w <= 64) {
        hipLaunchKernelGGL(gpool_f32_from_i8_kernel, dim3(n * (((c >> 2) + 31) / 32)), dim3(256), 0, s, n, h * w, c, type,
                           in_dtype == DT_U8, sc, (const uint8_t*)x, y, yq ? 1.f / q_scale : 0.f, yq);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(pool2d_f32_from_i8_kernel, dim3(grid_for((size_t)n * oh * ow * c)), dim3(256), 0, s, n, h, w, c,
                       oh, ow, kh, kw, sh, sw, ph, pw, type, in_dtype == DT_U8, sc, (const uint8_t*)x, y);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && yq) e = launch_quantize_flat_s8((size_t)n * c * oh * ow, q_scale, y, yq, s);
    return e;
}

// ---- softmax over the last axis: one 256-thread block per row, wavefront shuffles ---------------
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__global__ __launch_bounds__(256) void softmax_f32_kernel(int cols, const float* __restrict__ x,
                                                          float* __restrict__ y) {
    __shared__ float red[4];
    const float* xr = x + (size_t)blockIdx.x * cols;
    float* yr = y + (size_t)blockIdx.x * cols;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float m = -3.4e38f;
    for (int i = threadIdx.x; i < cols; i += 256) m = fmaxf(m, xr[i]);
    m = wave_max(m);
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float sum = 0.f;
    for (int i = threadIdx.x; i < cols; i += 256) sum += expf(xr[i] - m);
    sum = wave_sum(sum);
    if (lane == 0) red[wave] = sum;
    __syncthreads();
    sum = (red[0] + red[1]) + (red[2] + red[3]);
    for (int i = threadIdx.x; i < cols; i += 256) yr[i] = expf(xr[i] - m) / sum;
}
hipError_t launch_softmax_f32(int rows, int cols, const float* x, float* y, hipStream_t s) {
    hipLaunchKernelGGL(softmax_f32_kernel, dim3(rows), dim3(256), 0, s, cols, x, y);
    return hipGetLastError();
}

// ---- FP32 GEMM on v_mfma_f32_16x16x4_f32: C = alpha*op(A)*op(B) + beta*C (row-major) ------------
// 64x64 block tile, 2x2 waves of 32x32, K-step 16; operands staged through LDS as [rows][16 f32]
// with the same chunk swizzle as the conv kernel (each lane's 16-byte chunk feeds 4 MFMA k-groups).
typedef int g4i __attribute__((ext_vector_type(4)));
typedef float g4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int gswz(int row, int q) { return ((0x9C >> (2 * q)) & 3) ^ ((row >> 2) & 3); }

__global__ __launch_bounds__(256) void gemm_f32_kernel(int ta, int tb, int M, int N, int K, float alpha,
                                                       const float* __restrict__ A, const float* __restrict__ B,
                                                       float beta, float* __restrict__ C) {
    __shared__ float lds[2][64 * 16];  // [0]=A tile rows m, [1]=B tile rows n ; each row 16 k
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    g4f acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) acc[i][j] = g4f{0, 0, 0, 0};
    const int frow = lane & 15, fq = lane >> 4;
    for (int k0 = 0; k0 < K; k0 += 16) {
        // stage: 64 rows x 16 k per operand = 1024 floats each, 4 per thread
        for (int e = tid; e < 1024; e += 256) {
            const int r = e >> 4, kk = e & 15;
            const int k = k0 + kk;
            float va = 0.f, vb = 0.f;
            if (k < K) {
                if (m0 + r < M) va = ta ? A[(size_t)k * M + m0 + r] : A[(size_t)(m0 + r) * K + k];
                if (n0 + r < N) vb = tb ? B[(size_t)(n0 + r) * K + k] : B[(size_t)k * N + n0 + r];
            }
            const int q = kk >> 2, w = kk & 3;
            lds[0][r * 16 + gswz(r, q) * 4 + w] = va;
            lds[1][r * 16 + gswz(r, q) * 4 + w] = vb;
        }
        __syncthreads();
        g4f af[2], bf[2];
        for (int i = 0; i < 2; ++i) {
            const int row = (wm * 2 + i) * 16 + frow;
            af[i] = *(const g4f*)&lds[0][row * 16 + gswz(row, fq) * 4];
        }
        for (int j = 0; j < 2; ++j) {
            const int row = (wn * 2 + j) * 16 + frow;
            bf[j] = *(const g4f*)&lds[1][row * 16 + gswz(row, fq) * 4];
        }
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 2; ++j) {
                // rows of D = n (B operand as MFMA "A"), cols = m: lane gets 4 consecutive n for one m
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[j].x, af[i].x, acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[j].y, af[i].y, acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[j].z, af[i].z, acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[j].w, af[i].w, acc[i][j], 0, 0, 0);
            }
        __syncthreads();
    }
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j) {
            const int m = m0 + (wm * 2 + i) * 16 + frow;
            const int nb = n0 + (wn * 2 + j) * 16 + fq * 4;
            if (m >= M) continue;
            for (int r = 0; r < 4; ++r) {
                const int nn = nb + r;
                if (nn < N) {
                    const size_t o = (size_t)m * N + nn;
                    const float v = __fmul_rn(alpha, acc[i][j][r]);
                    C[o] = beta == 0.f ? v : __fadd_rn(v, __fmul_rn(beta, C[o]));
                }
            }
        }
}
hipError_t launch_gemm_f32(int ta, int tb, int m, int n, int k, float alpha, const float* a, const float* b,
                           float beta, float* c, hipStream_t s) {
    dim3 grid((n + 63) / 64, (m + 63) / 64);
    hipLaunchKernelGGL(gemm_f32_kernel, grid, dim3(256), 0, s, ta, tb, m, n, k, alpha, a, b, beta, c);
    return hipGetLastError();
}


// ---------------------------------------------------------------------------------------------
// L2 flush for cold-operand timing (see kernels.h)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void l2_flush_kernel(const uint4* buf, size_t n, unsigned* sink) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const uint4 v = buf[i];
        acc ^= v.x ^ v.w;
    }
    if (acc == 0x9e3779b9u) sink[0] = acc;      // keeps the loads alive; practically never taken
}
hipError_t launch_l2_flush(const void* buf, size_t bytes, unsigned* sink, hipStream_t s) {
    hipLaunchKernelGGL(l2_flush_kernel, dim3(2048), dim3(256), 0, s, (const uint4*)buf, bytes / 16, sink);
    return hipGetLastError();
}

// Workgroup b of a 1-D grid runs on XCD b % 8: the placement the XCD-aware tile orders assume and the FP32 split-K hand-off
// RELIES on (its partial sums meet in one XCD's L2). out[b] = HW_REG_XCC_ID of workgroup b.
__global__ void xcd_map_probe_kernel(unsigned* out) {
    if (threadIdx.x == 0) {
        unsigned v;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
        out[blockIdx.x] = v & 0xfu;
    }
}
hipError_t launch_xcd_map_probe(unsigned* out, int blocks, hipStream_t s) {
    hipLaunchKernelGGL(xcd_map_probe_kernel, dim3(blocks), dim3(64), 0, s, out);
    return hipGetLastError();
}

}  // namespace saber_mi355x
