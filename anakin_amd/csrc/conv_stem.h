// anakin_amd/csrc/conv_stem.h — the ResNet stem: INT8 7x7 / stride-2 convolution over a <=4-channel image
// with the input tile resident in LDS and the f32 -> s8 quantisation of the reference's
// "quantise on entry" (SaberConv2D<X86,AK_INT8>::dispatch -> reorder_nhwc_nchw, saber_conv.cpp:308,
// saber_util.h:759-781) fused into the staging (gfx950).
//
// The implicit-GEMM first-layer path gathers every input pixel ~12 times (49 taps / stride^2) from global
// memory, after a separate quantise kernel wrote the NHWC4 copy. Here a workgroup owns an 8 x 16 tile of
// output pixels of one image and 64 output channels: it reads the 21 x 38 input patch once (f32 NCHW
// planes, quantised on the fly: saturate(roundf(x * 1/scale)), packed as 4 bytes per pixel) into LDS
// together with the 16 KiB of weights, then runs 4 MFMA k-steps (two filter rows of 8x4 bytes each) whose
// B fragments are 16 contiguous LDS bytes at (2*py + i, 2*px + j). Same arithmetic and epilogues as
// conv_igemm_impl.h; weights use the first-layer repack [K][kh][8][4].
#pragma once
#include "epilogue_pack.h"

namespace saber_mi355x {

typedef int v2i __attribute__((ext_vector_type(2)));
typedef unsigned short us2_ __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pk_max_u16_(unsigned a, unsigned b) {   // two independent 16-bit unsigned maxima
    const us2_ r = __builtin_elementwise_max(__builtin_bit_cast(us2_, a), __builtin_bit_cast(us2_, b));
    return __builtin_bit_cast(unsigned, r);
}

// Input patch -> LDS as one dword per pixel (<= 4 channels), quantising f32 NCHW planes on the way
// (saturate(roundf(x * 1/scale)), saber_util.h:759-781). All of a thread's global loads are issued before the first
// conversion so that the patch costs one exposed memory latency, not one per pass.
template <bool F32IN, int NPIX, int ICP>
__device__ __forceinline__ void stage_input_patch(const ConvKArgs& a, int n, int iy0, int ix0, unsigned xmask, int tid,
                                                  unsigned* lds_x) {
    constexpr int NIT = (NPIX + 255) / 256;
    float fv[NIT][4];
    unsigned pv[NIT];
    bool ok[NIT];
    const size_t plane = (size_t)a.H * a.W;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int idx = tid + it * 256;
        const int r = idx / ICP, c = idx - r * ICP;
        const int iy = iy0 + r, ix = ix0 + c;
        ok[it] = idx < NPIX && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
        pv[it] = 0;
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) fv[it][ch] = 0.f;
        if (ok[it]) {
            if constexpr (F32IN) {
                const float* xp = (const float*)a.x + ((size_t)n * a.Cin * a.H + iy) * a.W + ix;
#pragma unroll
                for (int ch = 0; ch < 4; ++ch)
                    if (ch < a.Cin) fv[it][ch] = xp[ch * plane];
            } else {
                pv[it] = ((const unsigned*)a.x)[((size_t)n * a.H + iy) * a.W + ix];
            }
        }
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int idx = tid + it * 256;
        unsigned pk = pv[it];
        if constexpr (F32IN) {
            if (ok[it]) {
#pragma unroll
                for (int ch = 0; ch < 4; ++ch) {
                    if (ch < a.Cin) {
                        float v = __fmul_rn(fv[it][ch], a.qinv);
                        v = truncf(v + copysignf(0x1.fffffep-2f, v));          // roundf (conv_igemm_impl.h)
                        v = v < -128.f ? -128.f : (v > 127.f ? 127.f : v);      // saturate<int8_t>
                        pk |= ((unsigned)((int)v) & 0xffu) << (8 * ch);
                    }
                }
            }
        }
        if (idx < NPIX) lds_x[idx] = pk ^ xmask;   // padding becomes -128 for u8 inputs (compensated through comp)
    }
}

template <int EK, bool F32IN>
__global__ __launch_bounds__(256) void conv_stem7x7s2_kernel(const ConvKArgs a) {
    constexpr int TH = 8, TW = 16;
    constexpr int IR = (TH - 1) * 2 + 7 + 1;        // 22 input rows (one spare: the zero-weight 8th filter row)
    constexpr int ICP = 40;                         // input cols per LDS row ((TW-1)*2 + 8 = 38, padded to 40 -> 160 B)
    constexpr int TM = 2, TN = 4, NV = 8;
    constexpr int WCH = 64 * 16;                    // weight tile: 64 rows x 256 B

    __shared__ v4i lds_w[WCH];
    __shared__ unsigned lds_x[IR * ICP];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int frow = lane & 15, fq = lane >> 4;

    int ptile, tile_ky;
    xcd_tile(a, ptile, tile_ky);
    const int tiles_x = (a.OW + TW - 1) / TW, tiles_y = (a.OH + TH - 1) / TH;
    const int per_img = tiles_x * tiles_y;
    const int n = ptile / per_img;
    const int trem = ptile - n * per_img;
    const int ty0 = (trem / tiles_x) * TH, tx0 = (trem % tiles_x) * TW;
    const int k_base = tile_ky * 64;

    const int kb = k_base + wm * 32 + fq * NV;
    ChanParams<NV> cp;
    load_chan_params<NV>(a, kb, cp);

    // ---- weights -> LDS (rows permuted for 8-byte stores, chunks swizzled: CPR = 16 layout) ------------
    const v4i* w16 = (const v4i*)a.w;
    const int w_row_chunks = a.Kg_pad >> 4;
    v4i wv[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int idx = tid + it * 256;
        wv[it] = w16[(size_t)(k_base + (idx >> 4)) * w_row_chunks + (idx & 15)];
    }
    // ---- input patch -> LDS, quantising on the way ---------------------------------------------------
    const int iy0 = ty0 * 2 - a.pad_h, ix0 = tx0 * 2 - a.pad_w;
    const unsigned xmask = a.in_u8 ? 0x80808080u : 0u;
    stage_input_patch<F32IN, IR * ICP, ICP>(a, n, iy0, ix0, xmask, tid, lds_x);
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int idx = tid + it * 256;
        const int r = idx >> 4, q = idx & 15;
        const int rr = r & 31, wmr = r >> 5;
        const int lrow = (wmr * 2 + ((rr >> 2) & 1)) * 16 + (rr >> 3) * 4 + (rr & 3);
        lds_w[lrow * 16 + (q ^ (lrow & 15))] = wv[it];
    }
    __syncthreads();

    v4i acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = v4i{0, 0, 0, 0};

    int a_idx[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int row = (wm * TM + i) * 16 + frow;
        a_idx[i] = row * 16 + (fq ^ (row & 15));
    }
    // B fragment of k-step ks: filter row 2*ks + (fq>>1), taps 4*(fq&1) .. +3 -> 16 contiguous bytes
    int b_off[TN];   // dword index of (row 2*py + (fq>>1), col 2*px + 4*(fq&1)) at ks = 0
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int py = wn * TN + j;
        b_off[j] = (2 * py + (fq >> 1)) * ICP + 2 * frow + 4 * (fq & 1);
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        v4i af[TM], bf[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = lds_w[a_idx[i] ^ (ks << 2)];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const v2i* p = (const v2i*)&lds_x[b_off[j] + ks * 2 * ICP];   // 8-byte aligned (even dword index)
            const v2i lo = p[0], hi = p[1];
            bf[j] = v4i{lo.x, lo.y, hi.x, hi.y};
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = mma_step(af[i], bf[j], acc[i][j]);
    }

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
}

template <int EK>
static hipError_t launch_conv_stem_inst(int f32_in, const ConvKArgs& a, hipStream_t s) {
    ConvKArgs b = a;
    b.npx = a.N * ((a.OW + 15) / 16) * ((a.OH + 7) / 8);
    b.nky = (a.K + 63) / 64;
    b.mg_npx = magic_div(b.npx, (long long)b.npx * b.nky);
    dim3 grid(b.npx * b.nky), block(256);
    if (f32_in) hipLaunchKernelGGL((conv_stem7x7s2_kernel<EK, true>), grid, block, 0, s, b);
    else hipLaunchKernelGGL((conv_stem7x7s2_kernel<EK, false>), grid, block, 0, s, b);
    return hipGetLastError();
}


// ---------------------------------------------------------------------------------------------------------------
// SaberConv2DPooling<MI355X, AK_INT8> for the stem: the 7x7/2 convolution above followed by the 3x3 / stride-2 /
// pad-0 max pooling (ResNet conv1 + pool1), in one kernel; role in the reference: SaberConv2DPooling<X86,AK_INT8>
// (saber_conv_pooling.cpp:60-160: conv into an inner tensor, then pooling). A workgroup owns a 4 x 8 tile of POOLED
// pixels of one image and 64 channels: it computes the 9 x 17 convolution outputs under it (the one-pixel overlap
// with the neighbouring tiles is recomputed: +19 %), keeps them in LDS as bytes in unsigned order, and takes the
// window maxima from there. The 112x112x64 conv tensor is never written. Max pooling of the requantised bytes is
// exact, so the result equals pool(conv(x)) byte for byte; windows are clipped at the image border as
// SaberPooling does (ceil-mode output shape).
//
// TAIL (conv_stem_pool_pair_kernel): the two 1x1 / stride-1 convolutions that read the pooled tensor (ResNet res2a's `branch1`
// and `branch2a`, a sibling pair) run in the same launch on the workgroup's 32 pooled pixels: the pooled bytes go to LDS as the
// MFMA B operand (K = 64 = one i8 MFMA step), the pair's weights come straight from memory in A-fragment order (20 KB, packed by
// the host: saber_hip_conv2d_stem_pair_create), its per-channel constants by LDS-DMA at kernel entry. 20 units of 32 channels x 16
// pixels, five per wave; the epilogue is chain_out_pack = epilogue_i8_pair's arithmetic, so both outputs are the bits of the
// separate launch. The pooled tensor itself is written only when a.y is given.
template <bool F32IN, bool TAIL>
__device__ __forceinline__ void conv_stem_pool_body(const ConvKArgs& a, const StemPairTail& t) {
    constexpr int PH = 4, PW = 8;
    constexpr int CR = 2 * PH + 1, CC = 2 * PW + 1, NPX = CR * CC;      // 9 x 17 = 153 conv outputs
    constexpr int NG = (NPX + 15) / 16, TN = NG / 2;                    // 10 MFMA pixel groups, 5 per wave column
    constexpr int IR = (CR - 1) * 2 + 7 + 1, ICP = (CC - 1) * 2 + 8;    // 24 x 40 input pixels
    constexpr int TM = 2, NV = 8, WCH = 64 * 16;
    static_assert(NG % 2 == 0, "pixel groups split over two wave columns");

    __shared__ v4i lds_w[WCH];
    __shared__ __attribute__((aligned(16))) unsigned lds_x[IR * ICP];
    __shared__ uint2 lds_c[NG * 16 * 8];       // conv tile [pixel][64 channels], bytes in unsigned order
    constexpr int TPC = 256;                   // the pair's constants: 48 bytes per 4 channels, <= 320 channels (+ padding: 4 x 64 chunks)
    __shared__ v4i lds_tp[TAIL ? TPC : 1];
    SABER_TL_DECL;
    SABER_TL(0);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if constexpr (TAIL) lds_dma16((const v4i*)t.prm + tid, lds_tp + wave * 64);   // landed long before the tail (loads return in order)
    const int wm = wave >> 1, wn = wave & 1;
    const int frow = lane & 15, fq = lane >> 4;

    int ptile, tile_ky;
    xcd_tile(a, ptile, tile_ky);
    const int tiles_x = (a.pool_ow + PW - 1) / PW, tiles_y = (a.pool_oh + PH - 1) / PH;
    const int per_img = tiles_x * tiles_y;
    const int n = ptile / per_img;
    const int trem = ptile - n * per_img;
    const int py0 = (trem / tiles_x) * PH, px0 = (trem % tiles_x) * PW;   // pooled tile origin
    const int cy0 = py0 * 2, cx0 = px0 * 2;                               // conv-output origin (pool stride 2, pad 0)
    const int k_base = tile_ky * 64;

    const int kb = k_base + wm * 32 + fq * NV;
    ChanParams<NV> cp;
    load_chan_params<NV>(a, kb, cp);

    const v4i* w16 = (const v4i*)a.w;
    const int w_row_chunks = a.Kg_pad >> 4;
    v4i wv[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int idx = tid + it * 256;
        wv[it] = w16[(size_t)(k_base + (idx >> 4)) * w_row_chunks + (idx & 15)];
    }
    const int iy0 = cy0 * 2 - a.pad_h, ix0 = cx0 * 2 - a.pad_w;
    const unsigned xmask = a.in_u8 ? 0x80808080u : 0u;
    SABER_TL(1);
    stage_input_patch<F32IN, IR * ICP, ICP>(a, n, iy0, ix0, xmask, tid, lds_x);
    SABER_TL(2);
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int idx = tid + it * 256;
        const int r = idx >> 4, q = idx & 15;
        const int rr = r & 31, wmr = r >> 5;
        const int lrow = (wmr * 2 + ((rr >> 2) & 1)) * 16 + (rr >> 3) * 4 + (rr & 3);
        lds_w[lrow * 16 + (q ^ (lrow & 15))] = wv[it];
    }
    __syncthreads();

    v4i acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = v4i{0, 0, 0, 0};
    int a_idx[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int row = (wm * TM + i) * 16 + frow;
        a_idx[i] = row * 16 + (fq ^ (row & 15));
    }
    int b_off[TN];       // this lane's conv pixel of group j: flat index q -> (row q / CC, col q % CC) of the conv tile
    bool ok[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int q = (wn * TN + j) * 16 + frow;
        const int qq = q < NPX ? q : NPX - 1;
        const int r = qq / CC, c = qq - r * CC;
        b_off[j] = (2 * r + (fq >> 1)) * ICP + 2 * c + 4 * (fq & 1);
        ok[j] = q < NPX && cy0 + r < a.OH && cx0 + c < a.OW;
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        v4i af[TM], bf[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = lds_w[a_idx[i] ^ (ks << 2)];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const v2i* p = (const v2i*)&lds_x[b_off[j] + ks * 2 * ICP];
            const v2i lo = p[0], hi = p[1];
            bf[j] = v4i{lo.x, lo.y, hi.x, hi.y};
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = mma_step(af[i], bf[j], acc[i][j]);
    }

    SABER_TL(3);
    // conv epilogue -> LDS bytes in unsigned order (u8 as is, s8 + 128); positions outside the conv image hold 0,
    // the minimum, so a clipped window ignores them
    const bool u8 = a.out_dtype == DT_U8;
    const float lo_clamp = a.relu ? 0.f : -3.0e38f;
    const float off = u8 ? 0.f : 128.f;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int q = (wn * TN + j) * 16 + frow;
        unsigned pk[2] = {0u, 0u};
        if (ok[j]) {
#pragma unroll
            for (int v = 0; v < 2; ++v) {
                unsigned w = 0;
                float dq[4];
#pragma unroll
                for (int t = 0; t < 4; t += 2) {   // packed f32 add / mul on channel pairs (IEEE per component: same bits)
                    v2f d2 = {(float)(acc[v][j][t] + cp.comp[v * 4 + t]), (float)(acc[v][j][t + 1] + cp.comp[v * 4 + t + 1])};
                    d2 = d2 + v2f{cp.bias[v * 4 + t], cp.bias[v * 4 + t + 1]};
                    d2 = d2 * v2f{cp.scale[v * 4 + t], cp.scale[v * 4 + t + 1]};
                    dq[t] = d2.x;
                    dq[t + 1] = d2.y;
                }
                if (u8) {   // the saturating convert clamps at 0 itself: no max, no offset
#pragma unroll
                    for (int t = 0; t < 4; ++t) w = __builtin_amdgcn_cvt_pk_u8_f32(rintf(dq[t]), t, w);
                } else {
#pragma unroll
                    for (int t = 0; t < 4; ++t) w = __builtin_amdgcn_cvt_pk_u8_f32(fmaxf(rintf(dq[t]), lo_clamp) + off, t, w);
                }
                pk[v] = w;
            }
        }
        lds_c[q * 8 + wm * 4 + fq] = make_uint2(pk[0], pk[1]);
    }
    __syncthreads();

    SABER_TL(4);
    // the pair's weights, requested now (the accumulators are dead; any earlier and the kernel spills): wave w owns units u = w + 4 i,
    // unit = (32-channel group u >> 1, pixel group u & 1 - the same for all of a wave's units), 2 A fragments each
    constexpr int MAXU = 5;
    v4i taf[TAIL ? MAXU : 1][2];
    const int tng = TAIL ? (t.K1 + t.K2) >> 5 : 0;
    if constexpr (TAIL) {
#pragma unroll
        for (int i = 0; i < MAXU; ++i) {
            int g = (wave + 4 * i) >> 1;
            g = g < tng ? g : tng - 1;
#pragma unroll
            for (int mf = 0; mf < 2; ++mf) taf[i][mf] = ((const v4i*)t.w)[(g * 2 + mf) * 64 + lane];
        }
    }
    // 3x3 / stride 2 window maxima: one lane per (pooled pixel, 8 channels)
    const int pp = tid >> 3, cg = tid & 7;
    const int ppy = pp / PW, ppx = pp - ppy * PW;
    const int poy = py0 + ppy, pox = px0 + ppx;
    const int kc = k_base + cg * 8;
    if (poy < a.pool_oh && pox < a.pool_ow && kc < a.K) {
        unsigned ev[2] = {0, 0}, od[2] = {0, 0};
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const uint2 v = lds_c[((2 * ppy + dy) * CC + 2 * ppx + dx) * 8 + cg];
                ev[0] = pk_max_u16_(ev[0], v.x & 0x00ff00ffu);
                od[0] = pk_max_u16_(od[0], (v.x >> 8) & 0x00ff00ffu);
                ev[1] = pk_max_u16_(ev[1], v.y & 0x00ff00ffu);
                od[1] = pk_max_u16_(od[1], (v.y >> 8) & 0x00ff00ffu);
            }
        const unsigned flip = u8 ? 0u : 0x80808080u;
        const uint2 o = make_uint2((ev[0] | (od[0] << 8)) ^ flip, (ev[1] | (od[1] << 8)) ^ flip);
        if constexpr (TAIL) {
            // B operand of the pair: the pooled bytes as s8 (unsigned order ^ 0x80: u8 shifted by 128 - compensated through the
            // pair's comp - or the s8 value itself); pixel pitch 80 bytes: the 16 lanes of a fragment column hit 16 bank groups
            *(uint2*)&lds_x[pp * 20 + cg * 2] = make_uint2(o.x ^ flip ^ 0x80808080u, o.y ^ flip ^ 0x80808080u);
        }
        if (!TAIL || a.y) {
            uint8_t* y = (uint8_t*)a.y + (((size_t)n * a.pool_oh + poy) * a.pool_ow + pox) * a.K + kc;
            if (kc + 8 <= a.K) *(uint2*)y = o;
            else for (int t = 0; t < a.K - kc; ++t) y[t] = (uint8_t)((t < 4 ? o.x : o.y) >> (8 * (t & 3)));
        }
    }
    SABER_TL(5);
    if constexpr (TAIL) {
        __syncthreads();
        const int pg = wave & 1;
        const v4i bfr = *(const v4i*)&lds_x[(pg * 16 + frow) * 20 + fq * 4];
        const int tp = pg * 16 + frow;                        // this lane's pooled pixel of the tile
        const int toy = py0 + (tp >> 3), tox = px0 + (tp & 7);
        const bool tok = toy < a.pool_oh && tox < a.pool_ow;
        const size_t tpix = ((size_t)n * a.pool_oh + toy) * a.pool_ow + tox;
#pragma unroll
        for (int i = 0; i < MAXU; ++i) {
            const int g = (wave + 4 * i) >> 1;
            if (g >= tng) break;                              // wave-uniform
            const int kb = g * 32 + fq * 8;                   // 8 consecutive channels of the concatenated pair
            const v4i* pc = lds_tp + (kb >> 2) * 3;
            const v4i z = {0, 0, 0, 0};
            const v4i c0 = mma_step(taf[i][0], bfr, pc[2]), c1 = mma_step(taf[i][1], bfr, pc[5]);   // accumulate onto the compensation
            const bool second = kb >= t.K1;
            const bool ou8 = second ? t.u8_2 : t.u8_1;
            const float lo = (second ? t.relu2 : t.relu1) ? 0.f : -3.0e38f;
            const float off = ou8 ? 0.f : 128.f;
            const unsigned xm = ou8 ? 0u : 0x80808080u;
            const unsigned w0 = chain_out_pack(c0, z, __builtin_bit_cast(v4f, pc[1]), __builtin_bit_cast(v4f, pc[0]), lo, off, xm);
            const unsigned w1 = chain_out_pack(c1, z, __builtin_bit_cast(v4f, pc[4]), __builtin_bit_cast(v4f, pc[3]), lo, off, xm);
            if (tok) {
                uint8_t* y = second ? (uint8_t*)t.y2 + tpix * t.K2 + (kb - t.K1) : (uint8_t*)t.y1 + tpix * t.K1 + kb;
                *(uint2*)y = make_uint2(w0, w1);
            }
        }
        SABER_TL(6);
    }
    SABER_TL_FLUSH();
}

template <bool F32IN>
__global__ __launch_bounds__(256, 4) void conv_stem_pool_kernel(const ConvKArgs a) {
    conv_stem_pool_body<F32IN, false>(a, StemPairTail{});
}
template <bool F32IN>
__global__ __launch_bounds__(256, 4) void conv_stem_pool_pair_kernel(const StemPairKArgs ka) {
    conv_stem_pool_body<F32IN, true>(ka.c, ka.t);
}

static hipError_t launch_conv_stem_pool_inst(int f32_in, const ConvKArgs& a, hipStream_t s) {
    ConvKArgs b = a;
    b.npx = a.N * ((a.pool_ow + 7) / 8) * ((a.pool_oh + 3) / 4);
    b.nky = (a.K + 63) / 64;
    b.mg_npx = magic_div(b.npx, (long long)b.npx * b.nky);
    dim3 grid(b.npx * b.nky), block(256);
    if (f32_in) hipLaunchKernelGGL((conv_stem_pool_kernel<true>), grid, block, 0, s, b);
    else hipLaunchKernelGGL((conv_stem_pool_kernel<false>), grid, block, 0, s, b);
    return hipGetLastError();
}
static hipError_t launch_conv_stem_pool_pair_inst(int f32_in, const StemPairKArgs& ka, hipStream_t s) {
    StemPairKArgs b = ka;
    b.c.npx = ka.c.N * ((ka.c.pool_ow + 7) / 8) * ((ka.c.pool_oh + 3) / 4);
    b.c.nky = 1;                                   // K == 64 (checked by saber_hip_conv2d_stem_pair_create)
    b.c.mg_npx = magic_div(b.c.npx, (long long)b.c.npx);
    dim3 grid(b.c.npx), block(256);
    if (f32_in) hipLaunchKernelGGL((conv_stem_pool_pair_kernel<true>), grid, block, 0, s, b);
    else hipLaunchKernelGGL((conv_stem_pool_pair_kernel<false>), grid, block, 0, s, b);
    return hipGetLastError();
}

}  // namespace saber_mi355x
