// anakin_amd/csrc/conv_stem_f32.hip - the FP32 ResNet stem as ONE launch: NCHW f32 image -> conv1 (7x7 / stride 2 / pad 3, 3 -> 64) + bias +
// relu -> 3x3 / stride-2 max pooling -> NHWC f32 (round 6; round-5 verdict "next" item 2 iii).
//
// Role: SaberConv2DPooling<AK_FLOAT> (saber/funcs/conv_pooling.h; x86: saber_conv_pooling.cpp:13-57 runs the conv into an inner tensor and
// pools it; NV fuses the tail in the kernel: third-party/sass/include/sass_funcs.h:366-427 direct_conv_bias_relu_maxpool2k2s0p). Through round 5
// the FP32 op list spent three launches here - transpose_nchw_to_nhwc_f32 (the image into NHWC4) 10.3 us, the implicit-GEMM conv on the f32
// MFMA 35.5 us, pool2d_f32 10.5 us at batch 8 (profiles/r05_resnet50_fp32/sequence.txt #0 - #2) - and wrote + re-read the 25.7 MB conv
// output; the INT8 list has had its one-launch stem since round 2 (conv_stem.h).
//
// One workgroup (4 waves) owns a PTH x PTW tile of POOLED pixels x all 64 channels:
//   1. stage: the (2 CH + 5) x (2 CW + 6) input patch of the CH x CW = (2 PTH + 1) x (2 PTW + 1) conv pixels the tile needs is read from the
//      NCHW image (coalesced along x, zeros outside the image), each value split EXACTLY into three bf16 terms (x = h + m + l,
//      split3_pair) and stored to LDS as three planes of [row][column][4 channels] bf16 (channel 3 = 0) - once per tile, not per tap;
//   2. conv on the bf16 matrix cores (v_mfma_f32_16x16x32_bf16, six plane products per f32 product, small terms first: the arithmetic of
//      every bf16x3 kernel here, conv_igemm_impl.h MODE 3): the reduction is ordered k = (kh, kw', c) with kw' = kw + 1 in 0 .. 7 (kw' = 0 is
//      a zero weight) and c in 0 .. 3 (c = 3 is a zero weight) - one 32-deep MFMA step per filter ROW, and a lane's eight k values of pixel
//      (oy, ox) and row kh are the two input columns 2 ox - 4 + 2 kg, + 1 times four channels = 16 contiguous, 16-byte aligned LDS bytes
//      per plane: one ds_read_b128, no gather. Wave w owns output channels 16 w .. 16 w + 15, keeps their 7 x 3 weight fragments (84
//      VGPRs) for the whole launch, and walks over the tile's 16-pixel groups two at a time (two independent accumulators);
//   3. + bias, relu, the conv tile goes to LDS ([pixel][64 + 4 pad] f32); 4. the 3x3 / 2 maximum over it (windows clipped to the conv image:
//      ceil-mode shapes, pooling.h:109-115) is written NHWC, 256 contiguous bytes per pooled pixel.
// The conv pixels on a tile's seam are computed by both neighbours ((2 P + 1)^2 / (2 P)^2: 13 % at 8 x 8); nothing but the pooled tensor is
// written. Accumulation order differs from the implicit-GEMM kernels' (filter row major, the padded taps add exact zeros): inside the 1e-4
// FP32 contract like every FP32 kernel here; tests/test_gpu_parity.py::test_stem_f32_* pin it to the oracle and to the three separate ops.
#include "conv_igemm_impl.h"

#include <cstdlib>
#include <cstring>
#include <vector>

namespace saber_mi355x {

struct StemF32Args {
    const float* x;        // [N][3][H][W]
    const v4i* w;          // [4 channel tiles][7 filter rows][3 planes][64 lanes] x 8 bf16 (stem_f32_pack)
    const float* bias;     // [64] or null
    float* y;              // [N][PH][PW][64]
    int N, H, W, OH, OW, PH, PW, tiles_y, tiles_x;
};

template <int PTH, int PTW>
__global__ __launch_bounds__(256) void conv_stem_f32_pool_kernel(const StemF32Args a) {
    constexpr int CH = 2 * PTH + 1, CW = 2 * PTW + 1, NPIX = CH * CW;
    constexpr int IH = 2 * CH + 5, IW = 2 * CW + 6, NPOS = IH * IW;
    constexpr int NPT = (NPIX + 15) / 16;
    constexpr int CP = 68;                          // floats per conv pixel in LDS (64 + 4: a 16-lane store group covers all 64 banks)
    extern __shared__ v4i stem32_lds[];
    uint2* planes = (uint2*)stem32_lds;             // [3][NPOS]: four bf16 channels per position
    float* ctile = (float*)(planes + 3 * ((NPOS + 1) & ~1));      // [NPIX][CP]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, kg = lane >> 4;
    int b = blockIdx.x;
    const int tx = b % a.tiles_x; b /= a.tiles_x;
    const int ty = b % a.tiles_y;
    const int n = b / a.tiles_y;
    const int py0 = ty * PTH, px0 = tx * PTW;        // pooled origin; conv origin = 2 x that; input origin = 4 x that - (3, 4)
    const int iy0 = 4 * py0 - 3, ix0 = 4 * px0 - 4;

    // the wave's weight fragments: requested first, needed after the staging barrier
    v4i wreg[7][3];
#pragma unroll
    for (int kh = 0; kh < 7; ++kh)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) wreg[kh][pl] = a.w[((wave * 7 + kh) * 3 + pl) * 64 + lane];
    const float4 bs = a.bias ? *(const float4*)(a.bias + wave * 16 + kg * 4) : make_float4(0.f, 0.f, 0.f, 0.f);

    // 1. the input patch -> three bf16 planes
    const float* xn = a.x + (size_t)n * 3 * a.H * a.W;
    const size_t plane_hw = (size_t)a.H * a.W;
    // (every load of the patch is requested before the first one is used: NIT x 3 loads per thread in flight - a loop that loads, splits
    // and stores position by position pays one memory round trip per iteration)
    constexpr int NIT = (NPOS + 255) / 256;
    float cv[NIT][3];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int pos = tid + it * 256;
        const int liy = pos / IW, lix = pos - liy * IW;
        const int gy = iy0 + liy, gx = ix0 + lix;
        const bool in = pos < NPOS && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
        const float* p = xn + (in ? (size_t)gy * a.W + gx : 0);
        const float v0 = p[0], v1 = p[plane_hw], v2 = p[2 * plane_hw];      // unconditional (in-bounds) loads, selected below
        cv[it][0] = in ? v0 : 0.f;
        cv[it][1] = in ? v1 : 0.f;
        cv[it][2] = in ? v2 : 0.f;
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int pos = tid + it * 256;
        unsigned h01, m01, l01, h2, m2, l2;
        split3_pair(cv[it][0], cv[it][1], h01, m01, l01);
        split3_pair(cv[it][2], 0.f, h2, m2, l2);
        if (pos < NPOS) {
            planes[pos] = make_uint2(h01, h2);
            planes[((NPOS + 1) & ~1) + pos] = make_uint2(m01, m2);
            planes[2 * ((NPOS + 1) & ~1) + pos] = make_uint2(l01, l2);
        }
    }
    __syncthreads();

    // 2. + 3. conv pixels in groups of 16, two groups at a time
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};      // (weights, activations): small terms first
    const bool relu = true;
    // software-pipelined over (pixel-group pair, filter row) steps: the six fragments of step t + 1 are requested before the twelve
    // MFMAs of step t (a scheduling barrier keeps the compiler from sinking them back to their first use)
    constexpr int NPAIR = (NPT + 1) / 2, NSTEP = NPAIR * 7;
    auto frag_base = [&](int jj, int u) {
        int pp = (2 * jj + u) * 16 + frow;
        pp = pp < NPIX ? pp : NPIX - 1;              // (groups / pixels past the tile re-read its last pixel; never stored)
        const int cy = pp / CW, cx = pp - cy * CW;
        return (2 * cy) * IW + 2 * cx + 2 * kg;
    };
    auto load_step = [&](int base0, int base1, int kh, v4i (&bf)[2][3]) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            bf[0][pl] = *(const v4i*)(planes + pl * ((NPOS + 1) & ~1) + base0 + kh * IW);
            bf[1][pl] = *(const v4i*)(planes + pl * ((NPOS + 1) & ~1) + base1 + kh * IW);
        }
    };
    v4i bfr[2][2][3];
    int base_cur[2] = {frag_base(0, 0), frag_base(0, 1)};
    load_step(base_cur[0], base_cur[1], 0, bfr[0]);
    v4f acc[2] = {v4f{0.f, 0.f, 0.f, 0.f}, v4f{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int t = 0; t < NSTEP; ++t) {
        const int jj = t / 7, kh = t % 7;
        if (t + 1 < NSTEP) {
            if (kh == 6) { base_cur[0] = frag_base(jj + 1, 0); base_cur[1] = frag_base(jj + 1, 1); }
            load_step(base_cur[0], base_cur[1], kh == 6 ? 0 : kh + 1, bfr[(t + 1) & 1]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tt = 0; tt < 6; ++tt)
#pragma unroll
            for (int u = 0; u < 2; ++u)
                acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, wreg[kh][PA[tt]]),
                                                                 __builtin_bit_cast(v8bf, bfr[t & 1][u][PB[tt]]), acc[u], 0, 0, 0);
        if (kh == 6) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int p = (2 * jj + u) * 16 + frow;
                if (p < NPIX) {
                    float o[4] = {acc[u][0], acc[u][1], acc[u][2], acc[u][3]};
                    const float b4[4] = {bs.x, bs.y, bs.z, bs.w};
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float d = __fadd_rn(o[r], b4[r]);
                        if (relu) d = d > 0.f ? d : 0.f;
                        o[r] = d;
                    }
                    *(float4*)(ctile + p * CP + wave * 16 + kg * 4) = make_float4(o[0], o[1], o[2], o[3]);
                }
                acc[u] = v4f{0.f, 0.f, 0.f, 0.f};
            }
        }
    }
    __syncthreads();

    // 4. 3x3 / stride-2 maximum over the conv tile (relu'd values: >= 0), windows clipped to the conv image
    float* yn = a.y + (size_t)n * a.PH * a.PW * 64;
    for (int item = tid; item < PTH * PTW * 16; item += 256) {
        const int q = item & 15, pp = item >> 4;
        const int ppy = pp / PTW, ppx = pp - ppy * PTW;
        const int gpy = py0 + ppy, gpx = px0 + ppx;
        if (gpy >= a.PH || gpx >= a.PW) continue;
        float4 m = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                if (2 * gpy + dy < a.OH && 2 * gpx + dx < a.OW) {
                    const float4 v = *(const float4*)(ctile + ((2 * ppy + dy) * CW + 2 * ppx + dx) * CP + q * 4);
                    m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
                }
            }
        *(float4*)(yn + ((size_t)gpy * a.PW + gpx) * 64 + q * 4) = m;
    }
}

// host side ---------------------------------------------------------------------------------------------------------------------------
// w_oihw: [64][3][7][7] f32 -> [4 channel tiles][7 filter rows][3 planes][64 lanes] x 8 bf16: lane (r = lane & 15, kg = lane >> 4) of tile
// ct holds, for element j, the weight of output channel 16 ct + r, input channel c = j & 3 (0 for c = 3), tap (kh, kw = 2 kg + (j >> 2) - 1)
// (0 for kw = -1), split into the three bf16 terms with round-to-nearest-even conversions and exact subtractions (as set_weights does
// for d_w3)
void stem_f32_pack(const float* w_oihw, std::vector<uint8_t>& out) {
    auto rne = [](float x) {
        uint32_t u;
        std::memcpy(&u, &x, 4);
        u += 0x7fffu + ((u >> 16) & 1u);
        return (uint16_t)(u >> 16);
    };
    auto bf = [](uint16_t h) {
        const uint32_t u = (uint32_t)h << 16;
        float f;
        std::memcpy(&f, &u, 4);
        return f;
    };
    out.assign((size_t)4 * 7 * 3 * 64 * 16, 0);
    uint16_t* o = (uint16_t*)out.data();
    for (int ct = 0; ct < 4; ++ct)
        for (int kh = 0; kh < 7; ++kh)
            for (int lane = 0; lane < 64; ++lane) {
                const int r = lane & 15, kg = lane >> 4, ch = ct * 16 + r;
                for (int j = 0; j < 8; ++j) {
                    const int c = j & 3, kw = 2 * kg + (j >> 2) - 1;
                    const float w = (c < 3 && kw >= 0) ? w_oihw[(((size_t)ch * 3 + c) * 7 + kh) * 7 + kw] : 0.f;
                    const uint16_t h = rne(w);
                    const float r1 = w - bf(h);
                    const uint16_t m = rne(r1);
                    const float r2 = r1 - bf(m);
                    const uint16_t pl[3] = {h, m, rne(r2)};
                    for (int p3 = 0; p3 < 3; ++p3) o[((((size_t)ct * 7 + kh) * 3 + p3) * 64 + lane) * 8 + j] = pl[p3];
                }
            }
}

template <int PTH, int PTW>
static hipError_t launch_stem_f32_t(const StemF32Args& a0, hipStream_t s) {
    constexpr int CH = 2 * PTH + 1, CW = 2 * PTW + 1, IH = 2 * CH + 5, IW = 2 * CW + 6;
    constexpr size_t lds = (size_t)3 * ((IH * IW + 1) & ~1) * 8 + (size_t)CH * CW * 68 * 4;
    static bool once = false;
    if (!once) {
        hipError_t e = hipFuncSetAttribute((const void*)conv_stem_f32_pool_kernel<PTH, PTW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        once = true;
    }
    StemF32Args a = a0;
    a.tiles_y = (a.PH + PTH - 1) / PTH;
    a.tiles_x = (a.PW + PTW - 1) / PTW;
    hipLaunchKernelGGL((conv_stem_f32_pool_kernel<PTH, PTW>), dim3(a.N * a.tiles_y * a.tiles_x), dim3(256), lds, s, a);
    return hipGetLastError();
}

// variant 0: by launch size - 4 x 8 pooled pixels per workgroup (64 KB of LDS: two workgroups per CU overlap their staging / MFMA / pooling
// phases; measured 25.5 us against 36.4 for 8 x 8 at batch 8, profiles/r06/stem_f32.txt), 4 x 4 while that leaves CUs without a
// workgroup; 1 / 2 / 3 force 8 x 8 / 4 x 8 / 4 x 4
hipError_t launch_conv_stem_f32_pool(const StemF32Args& a, int variant, hipStream_t s) {
    if (variant == 0) {
        static const int forced = [] { const char* e = std::getenv("SABER_HIP_STEM_F32_TILE"); return e ? std::atoi(e) : 0; }();
        variant = forced ? forced : ((long)a.N * ((a.PH + 3) / 4) * ((a.PW + 7) / 8) >= 256 ? 2 : 3);
    }
    return variant == 1 ? launch_stem_f32_t<8, 8>(a, s) : (variant == 2 ? launch_stem_f32_t<4, 8>(a, s) : launch_stem_f32_t<4, 4>(a, s));
}

hipError_t launch_conv_stem_f32_pool_raw(const float* x, const void* w, const float* bias, float* y, int n, int h, int w_, int oh, int ow, int ph,
                                         int pw, int variant, hipStream_t s) {
    StemF32Args a;
    a.x = x; a.w = (const v4i*)w; a.bias = bias; a.y = y;
    a.N = n; a.H = h; a.W = w_; a.OH = oh; a.OW = ow; a.PH = ph; a.PW = pw; a.tiles_y = a.tiles_x = 0;
    return launch_conv_stem_f32_pool(a, variant, s);
}

}  // namespace saber_mi355x
