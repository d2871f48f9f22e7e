// anakin_amd/csrc/conv3x3_b3h.hip — FP32 3x3 / stride-1 / pad-1 convolution on the bf16 matrix cores with an LDS-resident input
// halo (gfx950). The FP32 counterpart of conv3x3_halo.h, for the three-plane scheme of conv_igemm_impl.h MODE 3.
//
// Why: the implicit-GEMM bf16-plane kernel re-gathers every input pixel once per filter tap and splits it into its three bf16
// planes each time; one 32-deep stage moves 72 KB through LDS (both operands, three planes) for 24 MFMAs per wave and is
// LDS-bound (profiles/r03/timeline_f32_bf16x3.txt: 0.56 us per stage against 0.16 us of matrix work). Here
//   * a workgroup owns a TH x 16 pixel tile of one image x BMK output channels; the (TH + 2) x 18 input halo of a 32-channel
//     chunk is loaded, split into (h, m, l) bf16 planes and stored to LDS ONCE, and serves all 9 taps (the B fragment of tap
//     (dy, dx) is the same LDS bytes at a shifted pixel): 9 x fewer gathers, splits and LDS stores, one barrier per chunk;
//   * the weights never touch LDS: the host packs the three planes in MFMA A-fragment order ([16-row tile][chunk][tap][plane]
//     [lane] x 16 B), a wave loads the fragments of its row tiles straight into registers, one tap ahead;
//   * eight waves (4 x 2, two per SIMD) or four (2 x 2): one wave's MFMAs cover the other's fragment traffic.
// Arithmetic: x = h + m + l exactly, six plane products per f32 product accumulated in f32 in mma_step3's order, taps in
// row-major order, channel chunks ascending - the same sum as MODE 3 with one 32-deep slab per stage except that MODE 3 walks
// (tap, chunk) tap-major; both are within ~1e-6 of the f64 result (the FP32 contract is 1e-4).
// Epilogue: epilogue_f32 of conv_igemm_impl.h (bias, relu, in-place residual sum), or the fused 2x2 max pooling; NHWC f32.
#include "conv_igemm_impl.h"

namespace saber_mi355x {

namespace {

// 64-byte-row swizzle (phys_chunk<4>): chunk q of row r lives at g(q) ^ ((r >> 2) & 3)
__device__ __forceinline__ int b3h_swz(int row, int q) { return ((0x9C >> (2 * q)) & 3) ^ ((row >> 2) & 3); }

// NWM x NWN waves; TM: 16-channel tiles per wave; TH: tile rows (pixels = TH x 16); the NWN wave columns split the rows: wave
// column wn owns rows [wn * TH / NWN, (wn + 1) * TH / NWN)
// KS = 1: the same structure for a 1x1 stride-1 convolution: the "tile" is a run of TH * 16 consecutive pixels of the [N*H*W] list
// (no halo, one tap per chunk), C % 64 == 0.
template <int NWM, int TM, int TH, int NWN = 2, int KS = 3>
__global__ __launch_bounds__(NWM * NWN * 64) void conv3x3_b3h_kernel(const ConvKArgs a) {
    constexpr int NTHR = NWM * NWN * 64;
    constexpr bool PW = KS == 1;                     // pointwise
    constexpr int TW = 16, HW_ = PW ? TW : TW + 2, HP = PW ? TH * TW : (TH + 2) * HW_;
    constexpr int XCH = HP * 4;                      // 16-byte chunks of one plane of the halo (32 bf16 = 64 B per pixel)
    constexpr int XIT = (XCH + NTHR - 1) / NTHR;
    constexpr int TN = TH / NWN, NV = TM * 4;
    constexpr int BMK = NWM * TM * 16;

    __shared__ v4i lds[2][3][XCH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / NWN, wn = wave % NWN;
    const int frow = lane & 15, fq = lane >> 4;

    int ptile, tile_ky;
    xcd_tile(a, ptile, tile_ky);                     // a.npx = N * tiles_y * tiles_x (pointwise: ceil(M / HP)), a.nky = ceil(K / BMK)
    const int tiles_x = PW ? 1 : (a.OW + TW - 1) / TW;
    const int tiles_y = PW ? 1 : (a.OH + TH - 1) / TH;
    const int per_img = tiles_x * tiles_y;
    const int n = PW ? 0 : ptile / per_img;
    const int trem = ptile - n * per_img;
    const int ty0 = (trem / tiles_x) * TH, tx0 = (trem % tiles_x) * TW;
    const int p0 = ptile * HP;                       // pointwise: first pixel of the run
    const int k_base = tile_ky * BMK;
    const int kb = k_base + wm * (TM * 16) + fq * NV;

    // ---- halo staging: thread -> (halo pixel, 8-channel group) items, fixed for all chunks --------------------------------
    int x_off[XIT], x_dst[XIT];
#pragma unroll
    for (int it = 0; it < XIT; ++it) {
        const int idx = tid + it * NTHR;
        const int hp = idx >> 2, q = idx & 3;
        const int hy = hp / HW_, hx = hp - hy * HW_;
        const int iy = ty0 - 1 + hy, ix = tx0 - 1 + hx;
        const bool ok = PW ? (idx < XCH && p0 + hp < a.M) : ((idx < XCH) && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W);
        x_off[it] = ok ? (PW ? (p0 + hp) : ((n * a.H + iy) * a.W + ix)) * a.C + q * 8 : -1;       // in f32 elements
        x_dst[it] = idx < XCH ? hp * 4 + b3h_swz(hp, q) : -1;
    }
    const float* xg = (const float*)a.x;
    v4i xv[XIT][2];
    auto load_chunk = [&](int cc) {
#pragma unroll
        for (int it = 0; it < XIT; ++it) {
            xv[it][0] = v4i{0, 0, 0, 0};
            xv[it][1] = v4i{0, 0, 0, 0};
            if (x_off[it] >= 0) {
                const v4i* p = (const v4i*)(xg + x_off[it] + cc * 32);
                xv[it][0] = p[0];
                xv[it][1] = p[1];
            }
        }
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int it = 0; it < XIT; ++it) {
            if (x_dst[it] < 0) continue;
            const v4f f0 = __builtin_bit_cast(v4f, xv[it][0]), f1 = __builtin_bit_cast(v4f, xv[it][1]);
            unsigned h[4], m[4], l[4];
            split3_pair(f0.x, f0.y, h[0], m[0], l[0]);
            split3_pair(f0.z, f0.w, h[1], m[1], l[1]);
            split3_pair(f1.x, f1.y, h[2], m[2], l[2]);
            split3_pair(f1.z, f1.w, h[3], m[3], l[3]);
            lds[buf][0][x_dst[it]] = v4i{(int)h[0], (int)h[1], (int)h[2], (int)h[3]};
            lds[buf][1][x_dst[it]] = v4i{(int)m[0], (int)m[1], (int)m[2], (int)m[3]};
            lds[buf][2][x_dst[it]] = v4i{(int)l[0], (int)l[1], (int)l[2], (int)l[3]};
        }
    };

    // ---- weights: [16-row tile][chunk][tap][plane][lane] fragments, this wave's TM row tiles ---------------------------------
    const int nchunks = a.C >> 5;
    const v4i* wf = (const v4i*)a.w + lane;
    size_t w_tile[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) w_tile[i] = (size_t)(k_base / 16 + wm * TM + i) * nchunks * (KS * KS * 3 * 64);
    v4i af[2][TM][3];
    auto load_w = [&](int cc, int t, auto set_c) {
        constexpr int SET = decltype(set_c)::value;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const v4i* p = wf + w_tile[i] + (size_t)(cc * (KS * KS) + t) * (3 * 64);
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) af[SET][i][pl] = p[pl * 64];
        }
    };

    v4f acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = v4f{0.f, 0.f, 0.f, 0.f};
    int b_hp[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) b_hp[j] = (wn * TN + j) * HW_ + frow;      // halo pixel of tap (0, 0) for pixel row j

    using std::integral_constant;
    load_chunk(0);
    load_w(0, 0, integral_constant<int, 0>{});
    ChanParams<NV> cp;
    load_chan_params<NV>(a, kb, cp);
    store_chunk(0);
    __syncthreads();

    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};      // mma_step3's order: small terms first
    auto mma_tap = [&](const v4i (*L)[XCH], int t, auto set_c) {
        constexpr int SET = decltype(set_c)::value;
        const int ti = t / 3, tj = t - ti * 3;
        v4i bf[TN][3];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int hp = b_hp[j] + ti * HW_ + tj;
            const int ci = hp * 4 + b3h_swz(hp, fq);
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) bf[j][pl] = L[pl][ci];
        }
#pragma unroll
        for (int tt = 0; tt < 6; ++tt)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, af[SET][i][PA[tt]]),
                                                                        __builtin_bit_cast(v8bf, bf[j][PB[tt]]), acc[i][j], 0, 0, 0);
    };

    if constexpr (PW) {
        // one tap per chunk: chunks cc (set 0, buffer 0) and cc + 1 (set 1, buffer 1) per iteration (C % 64 == 0), the next chunk's
        // pixels and weight fragments requested before the current one is multiplied
        for (int cc = 0; cc < nchunks; cc += 2) {
            load_chunk(cc + 1);
            load_w(cc + 1, 0, integral_constant<int, 1>{});
            mma_tap(lds[0], 0, integral_constant<int, 0>{});
            store_chunk(1);
            __syncthreads();
            const bool more = cc + 2 < nchunks;
            if (more) {
                load_chunk(cc + 2);
                load_w(cc + 2, 0, integral_constant<int, 0>{});
            }
            mma_tap(lds[1], 0, integral_constant<int, 1>{});
            if (more) {
                store_chunk(0);
                __syncthreads();
            }
        }
    } else
    for (int cc = 0; cc < nchunks; ++cc) {
        const int buf = cc & 1;
        const bool more = cc + 1 < nchunks;
        if (more) load_chunk(cc + 1);
        const v4i (*L)[XCH] = lds[buf];
        // taps 0..8, the next tap's weight fragments requested before the current tap is multiplied (tap 8: the next chunk's tap 0;
        // past the end the request is clamped to the last chunk and never used)
        const int ccn = more ? cc + 1 : cc;
#define B3H_TAP(t, S0, S1, CN, TNX)                                  \
        load_w(CN, TNX, integral_constant<int, S1>{});               \
        mma_tap(L, t, integral_constant<int, S0>{});
        B3H_TAP(0, 0, 1, cc, 1)
        B3H_TAP(1, 1, 0, cc, 2)
        B3H_TAP(2, 0, 1, cc, 3)
        B3H_TAP(3, 1, 0, cc, 4)
        B3H_TAP(4, 0, 1, cc, 5)
        B3H_TAP(5, 1, 0, cc, 6)
        B3H_TAP(6, 0, 1, cc, 7)
        B3H_TAP(7, 1, 0, cc, 8)
        // tap 8 of this chunk sits in set 0; the next chunk's tap 0 must land in set 0 as well: request it into set 1, copy over
        load_w(ccn, 0, integral_constant<int, 1>{});
        mma_tap(L, 8, integral_constant<int, 0>{});
#undef B3H_TAP
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) af[0][i][pl] = af[1][i][pl];
        if (more) {
            store_chunk(buf ^ 1);
            __syncthreads();
        }
    }

    // ---- epilogue ----------------------------------------------------------------------------------------------------------
    if (!PW && a.pool_ow) {
        // SaberConv2DPooling<AK_FLOAT>: relu'd conv + 2x2 / stride-2 max pooling (OH, OW even). A window = rows j, j + 1 of this
        // wave (its TN rows start at an even row) x lanes frow, frow ^ 1: max in registers, then across the lane pair; the max of
        // the four relu(conv + bias) values, like epilogue_f32_pool2.
        static_assert(TN % 2 == 0, "a pooling window's two rows belong to one wave");
#pragma unroll
        for (int j = 0; j < TN; j += 2) {
            const int oy = ty0 + wn * TN + j, ox = tx0 + frow;
            float o[NV];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float d0 = __fadd_rn(acc[i][j][r], cp.bias[i * 4 + r]), d1 = __fadd_rn(acc[i][j + 1][r], cp.bias[i * 4 + r]);
                    d0 = d0 > 0.f ? d0 : 0.f;
                    d1 = d1 > 0.f ? d1 : 0.f;
                    float d = fmaxf(d0, d1);
                    const int t = __float_as_int(d);
                    d = fmaxf(d, __int_as_float(__builtin_amdgcn_update_dpp(t, t, 0xB1, 0xF, 0xF, false)));   // quad_perm [1,0,3,2]
                    o[i * 4 + r] = d;
                }
            if ((frow & 1) || oy >= a.OH || ox >= a.OW || kb >= a.K) continue;
            float* y = (float*)a.y + ((size_t)(n * a.pool_oh + (oy >> 1)) * a.pool_ow + (ox >> 1)) * a.K + kb;
            if ((kb + NV <= a.K) && ((a.K & 3) == 0)) {
#pragma unroll
                for (int v = 0; v < NV; v += 4) *(float4*)(y + v) = make_float4(o[v], o[v + 1], o[v + 2], o[v + 3]);
            } else {
                for (int r = 0; r < NV; ++r)
                    if (kb + r < a.K) y[r] = o[r];
            }
        }
        return;
    }
    const int ohw = a.OH * a.OW;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int oy = ty0 + wn * TN + j, ox = tx0 + frow;
        if (!PW && (oy >= a.OH || ox >= a.OW)) continue;
        const int p = PW ? p0 + (wn * TN + j) * 16 + frow : (n * a.OH + oy) * a.OW + ox;
        if (PW && p >= a.M) continue;
        float v[NV];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[i * 4 + r] = acc[i][j][r];
        epilogue_f32<NV>(a, v, cp, p, kb, n, p - n * ohw);
    }
}

}  // namespace

bool conv3x3_b3h_variant(int variant, int* bmk, int* th, int* tm, int* threads) {
    // 1: 128 ch x 8 rows, 8 waves; 2: 64 ch x 8 rows, 4 waves; 3: 64 ch x 8 rows, 8 waves; 4: 64 ch x 4 rows, 4 waves; 5: 128 ch x 4 rows, 8 waves
    // (a 64 ch x 16 rows form with 2 x 4 waves was measured too: 156-165 TF on VGG16's layers against 172-206 for variant 2)
    // 6..8: the pointwise (1x1) forms: 64 ch x 128 pixels 4 waves, 128 ch x 128 pixels 8 waves, 64 ch x 64 pixels 4 waves
    static const int tab[9][4] = {{0, 0, 0, 0}, {128, 8, 2, 512}, {64, 8, 2, 256}, {64, 8, 1, 512}, {64, 4, 2, 256}, {128, 4, 2, 512},
                                  {64, 8, 2, 256}, {128, 8, 2, 512}, {64, 4, 2, 256}};
    if (variant < 1 || variant > 8) return false;
    *bmk = tab[variant][0]; *th = tab[variant][1]; *tm = tab[variant][2]; *threads = tab[variant][3];
    return true;
}

// a.w: the fragment-ordered planes for the variant's TM (api_conv.hip: pack_b3h). Requires 3x3 / stride 1 / pad 1 / dilation 1,
// C % 32 == 0, NHWC f32 in and out, EPI_F32 without the pair / pooling epilogues.
hipError_t launch_conv3x3_b3h(int variant, const ConvKArgs& a, hipStream_t s) {
    int bmk, th, tm, thr;
    if (!conv3x3_b3h_variant(variant, &bmk, &th, &tm, &thr)) return hipErrorInvalidValue;
    const bool pw = variant >= 6;
    if (a.kh != (pw ? 1 : 3) || a.kw != a.kh || a.stride_h != 1 || a.stride_w != 1 || a.pad_h != (pw ? 0 : 1) || a.pad_w != a.pad_h || a.dil_h != 1 ||
        a.dil_w != 1 || (a.C & (pw ? 63 : 31)) || a.out_nchw || a.K2 || (a.pool_ow && (pw || ((a.OH | a.OW) & 1))))
        return hipErrorInvalidValue;
    ConvKArgs b = a;
    b.npx = pw ? (a.M + th * 16 - 1) / (th * 16) : a.N * ((a.OW + 15) / 16) * ((a.OH + th - 1) / th);
    b.nky = (a.K + bmk - 1) / bmk;
    b.mg_npx = magic_div(b.npx, (long long)b.npx * b.nky);
    dim3 grid(b.npx * b.nky), block(thr);
    switch (variant) {
    case 1: hipLaunchKernelGGL((conv3x3_b3h_kernel<4, 2, 8>), grid, block, 0, s, b); break;
    case 2: hipLaunchKernelGGL((conv3x3_b3h_kernel<2, 2, 8>), grid, block, 0, s, b); break;
    case 3: hipLaunchKernelGGL((conv3x3_b3h_kernel<4, 1, 8>), grid, block, 0, s, b); break;
    case 4: hipLaunchKernelGGL((conv3x3_b3h_kernel<2, 2, 4>), grid, block, 0, s, b); break;
    case 5: hipLaunchKernelGGL((conv3x3_b3h_kernel<4, 2, 4>), grid, block, 0, s, b); break;
    case 6: hipLaunchKernelGGL((conv3x3_b3h_kernel<2, 2, 8, 2, 1>), grid, block, 0, s, b); break;
    case 7: hipLaunchKernelGGL((conv3x3_b3h_kernel<4, 2, 8, 2, 1>), grid, block, 0, s, b); break;
    case 8: hipLaunchKernelGGL((conv3x3_b3h_kernel<2, 2, 4, 2, 1>), grid, block, 0, s, b); break;
    }
    return hipGetLastError();
}

}  // namespace saber_mi355x
