// anakin_amd/csrc/conv3x3_halo.h — INT8 3x3 / stride-1 convolution with an LDS-resident input halo (gfx950).
//
// The implicit-GEMM kernels re-gather every input pixel once per filter tap (9x for a 3x3). Here a
// workgroup owns a TH x 16 spatial tile of one image and 64 output channels: it loads the (TH+2) x 18
// input halo of a 64-channel chunk ONCE (all loads of a chunk in flight together: one exposed memory
// latency), then runs the 9 taps as 9 MFMA k-steps whose B fragments are the same LDS bytes read at a
// shifted pixel offset. Weights of the chunk ([64][9 taps][64 B]) sit beside it. For C > 64 the channel
// chunks are double-buffered (next chunk's global loads in flight while the current one is multiplied).
//
// Same arithmetic, fragment mapping, channel permutation and epilogues as conv_igemm_impl.h
// (v_mfma_i32_16x16x64_i8, u8 -> s8 by XOR 0x80 with the +128*sum(w) compensation, which also turns the
// zero padding of the halo into -128).  Role in the reference: the 3x3 layers of
// GemmX8S8S32XConv::sub_dispatch (gemm_x8s8s32x_conv.cpp:187-288), without its im2col.
#pragma once
#include "conv_igemm_impl.h"

namespace saber_mi355x {

// 64-byte-row swizzle (same as phys_chunk<4>): chunk q of row r lives at g(q) ^ ((r>>2)&3)
__device__ __forceinline__ int swz4(int row, int q) { return ((0x9C >> (2 * q)) & 3) ^ ((row >> 2) & 3); }

template <int TH, int EK, bool MULTI>
__global__ __launch_bounds__(256) void conv3x3_halo_kernel(const ConvKArgs a) {
    constexpr int TW = 16;
    constexpr int HW_ = TW + 2;                     // halo width in pixels
    constexpr int HP = (TH + 2) * HW_;              // halo pixels
    constexpr int XCH = HP * 4;                     // 16-byte chunks of the halo (64 B per pixel)
    constexpr int WCH = 64 * 36;                    // 64 rows x 9 taps x 4 chunks
    constexpr int XIT = (XCH + 255) / 256;
    constexpr int WIT = WCH / 256;                  // 9
    constexpr int TM = 2, TN = TH / 2, NV = 8;
    constexpr int NBUF = MULTI ? 2 : 1;
    constexpr int BUF = XCH + WCH;                  // chunks per buffer

    __shared__ v4i lds[NBUF][BUF];
    SABER_TL_DECL;
    SABER_TL(0);
    pin_hot_args(a);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int frow = lane & 15, fq = lane >> 4;

    // ---- which tile ------------------------------------------------------------------------------
    int ptile, tile_ky;
    xcd_tile(a, ptile, tile_ky);                    // a.npx = N * tiles_y * tiles_x
    const int tiles_x = (a.OW + TW - 1) / TW;
    const int tiles_y = (a.OH + TH - 1) / TH;
    const int per_img = tiles_x * tiles_y;
    const int n = ptile / per_img;
    const int trem = ptile - n * per_img;
    const int ty0 = (trem / tiles_x) * TH, tx0 = (trem % tiles_x) * TW;
    const int k_base = tile_ky * 64;

    const int kb = k_base + wm * 32 + fq * NV;

    // ---- staging assignments (fixed per thread) -----------------------------------------------------
    int x_off[XIT];                                 // element offset of the halo pixel (chunk q), or -1 if padding
    int x_dst[XIT];
#pragma unroll
    for (int it = 0; it < XIT; ++it) {
        const int idx = tid + it * 256;
        const int hp = idx >> 2, q = idx & 3;
        const int hy = hp / HW_, hx = hp - hy * HW_;
        const int iy = ty0 - a.pad_h + hy, ix = tx0 - a.pad_w + hx;
        const bool ok = (idx < XCH) && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
        x_off[it] = ok ? ((n * a.H + iy) * a.W + ix) * a.C + q * 16 : -1;
        x_dst[it] = idx < XCH ? hp * 4 + swz4(hp, q) : -1;
    }
    int w_off[WIT], w_dst[WIT];
#pragma unroll
    for (int it = 0; it < WIT; ++it) {
        const int idx = tid + it * 256;
        const int row = idx / 36, c36 = idx - row * 36;
        const int tap = c36 >> 2, q = c36 & 3;
        w_off[it] = (k_base + row) * a.Kg_pad + tap * a.C + q * 16;
        // permuted LDS row (MFMA tile (wm, tm) reads 16 consecutive rows), TM = 2: see conv_igemm_impl.h
        const int rr = row & 31, wmr = row >> 5;
        const int lrow = (wmr * 2 + ((rr >> 2) & 1)) * 16 + (rr >> 3) * 4 + (rr & 3);
        w_dst[it] = XCH + lrow * 36 + tap * 4 + swz4(lrow, q);
    }
    const unsigned xmask = a.in_u8 ? 0x80808080u : 0u;
    const char* xg = (const char*)a.x;
    const char* wg = (const char*)a.w;

    v4i xv[XIT], wv[WIT];
    auto load_chunk = [&](int cc) {
        const int coff = cc * 64;
#pragma unroll
        for (int it = 0; it < WIT; ++it) wv[it] = *(const v4i*)(wg + w_off[it] + coff);
#pragma unroll
        for (int it = 0; it < XIT; ++it) {
            v4i v = {0, 0, 0, 0};
            if (x_off[it] >= 0) v = *(const v4i*)(xg + x_off[it] + coff);
            v.x ^= xmask; v.y ^= xmask; v.z ^= xmask; v.w ^= xmask;
            xv[it] = v;
        }
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int it = 0; it < WIT; ++it) lds[buf][w_dst[it]] = wv[it];
#pragma unroll
        for (int it = 0; it < XIT; ++it)
            if (x_dst[it] >= 0) lds[buf][x_dst[it]] = xv[it];
    };

    v4i acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = v4i{0, 0, 0, 0};

    // A-fragment base (per MFMA tile): LDS row and its swizzled chunk
    int a_base[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int row = (wm * TM + i) * 16 + frow;
        a_base[i] = XCH + row * 36 + swz4(row, fq);
    }
    // B-fragment base halo pixel (tap (0,0)) of pixel group j
    int b_hp[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) b_hp[j] = (wn * TN + j) * HW_ + frow;

    const int nchunks = a.C >> 6;
    SABER_TL(1);
    load_chunk(0);
    ChanParams<NV> cp;   // per-channel constants: behind the first operand loads (cold argument fields), ahead of the loop
    load_chan_params<NV>(a, kb, cp);
    store_chunk(0);
    __syncthreads();
    SABER_TL(2);
    for (int cc = 0; cc < nchunks; ++cc) {
        const int buf = MULTI ? (cc & 1) : 0;
        if (MULTI && cc + 1 < nchunks) load_chunk(cc + 1);
        const v4i* L = lds[buf];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int ti = t / 3, tj = t - ti * 3;
            v4i af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = L[a_base[i] + t * 4];
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int hp = b_hp[j] + ti * HW_ + tj;
                bf[j] = L[hp * 4 + swz4(hp, fq)];
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = mma_step(af[i], bf[j], acc[i][j]);
        }
        if (MULTI && cc + 1 < nchunks) {
            store_chunk(buf ^ 1);
            __syncthreads();
        }
    }

    SABER_TL(3);
    // ---- epilogue ------------------------------------------------------------------------------------
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int oy = ty0 + wn * TN + j, ox = tx0 + frow;
        if (oy >= a.OH || ox >= a.OW) continue;
        const int p = (n * a.OH + oy) * a.OW + ox;
        int v[NV];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[i * 4 + r] = acc[i][j][r];
        if constexpr (EK == EK_GEN) {
            epilogue_i8<NV>(a, v, cp, p, kb);
        } else {
            if (kb < a.K) {
                epilogue_i8_fast<NV, EK>(a, v, cp, p, kb);   // K % 16 == 0 here (epilogue_kind)
            }
        }
    }
    SABER_TL(4);
    SABER_TL_FLUSH();
}

// th: 4 or 8 tile rows. Requires kh = kw = 3, stride 1, dilation 1, C % 64 == 0, group 1.
template <int EK>
static hipError_t launch_conv3x3_halo_inst(int th, const ConvKArgs& a, hipStream_t s) {
    ConvKArgs b = a;
    const int tiles = ((a.OW + 15) / 16) * ((a.OH + th - 1) / th);
    b.npx = a.N * tiles;
    b.nky = (a.K + 63) / 64;
    b.mg_npx = magic_div(b.npx, (long long)b.npx * b.nky);
    dim3 grid(b.npx * b.nky), block(256);
    const bool multi = a.C > 64;
    if (th == 4) {
        if (multi) hipLaunchKernelGGL((conv3x3_halo_kernel<4, EK, true>), grid, block, 0, s, b);
        else hipLaunchKernelGGL((conv3x3_halo_kernel<4, EK, false>), grid, block, 0, s, b);
    } else if (th == 8) {
        if (multi) hipLaunchKernelGGL((conv3x3_halo_kernel<8, EK, true>), grid, block, 0, s, b);
        else hipLaunchKernelGGL((conv3x3_halo_kernel<8, EK, false>), grid, block, 0, s, b);
    } else {
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace saber_mi355x
