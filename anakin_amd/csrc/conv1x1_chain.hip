// anakin_amd/csrc/conv1x1_chain.hip — two back-to-back 1x1 INT8 convolutions in one launch (gfx950).
//
// Role: the ResNet bottleneck boundary  branch2c (1x1, C1 -> 4*C1) + SaberEltwise sum + relu  ->  next block's
// branch2a (1x1, 4*C1 -> C1, relu). Both are per-pixel operators (reference: GemmX8S8S32XConv::dispatch,
// saber/funcs/impl/x86/gemm_x8s8s32x_conv.cpp:184-257, and SaberEltwise<X86,AK_INT8>, saber_eltwise_int8.cpp), so a
// workgroup that owns a tile of pixels can run the second conv on the first one's output without leaving the CU:
// one launch, one read of the tile instead of a write + re-read, and no second launch boundary (2.3-2.7 us each on this
// part, more than either kernel's arithmetic). Results are the same bits as the two separate launches: the second conv
// consumes exactly the s8 values the first one stores.
//
//   workgroup = NPX = 16*TN pixels x ALL channels of both convs, 4 waves;
//   wave w owns output channels [w*K1/4, (w+1)*K1/4) of the first conv and [w*K2/4, (w+1)*K2/4) of the second;
//   weights never touch LDS: the host packs both convs' weights into ONE stream per wave, already in MFMA A-operand
//   order and in the order the wave consumes them (1 KB = one k-step of 16 channels per "step"), so the inner loop is
//   `MFMA(ring[r], b, acc); ring[r] = stream[next]` on an R-deep register ring - a flat prefetch that runs across the
//   boundary between the two convs;
//   the B operand of the first conv (the pixel tile of x) sits in registers; its epilogue takes the residual tile from
//   LDS (brought in by LDS-DMA, XOR-swizzled), writes the s8 result over it, and after ONE barrier the same LDS tile is
//   (a) streamed to y1 with 16-byte coalesced stores and (b) the B operand of the second conv;
//   the per-channel constants {scale, bias', comp} of both convs are DMA'd into LDS once.
// Channel <-> MFMA row mapping (chosen by the host packing): in a group of 16*MFG channels handled by MFG accumulators,
// row rho of accumulator mf is channel  base + (rho >> 2) * 4*MFG + mf*4 + (rho & 3), so a lane (which holds rows
// 4*fq .. 4*fq+3 of every accumulator) ends up with 4*MFG CONSECUTIVE channels of its pixel: one 16-byte LDS / global
// access per pixel instead of four 4-byte ones.
#include "epilogue_pack.h"

namespace saber_mi355x {


// KS1 = C1/64, G1 = K1/256 (64-channel groups per wave, first 1x1 conv), MFG2 / G2: accumulators per group / groups per
// wave of the second 1x1 conv (K2 = 64*G2*MFG2), TN pixel fragments, R ring depth in steps (1 KB per wave each; divides
// the group lengths of both 1x1 convs).
// C3: the block's 3x3 conv (C1 -> C1, stride 1, pad 1, the `branch2b` in front of the first 1x1 conv) runs in the same
// launch as "phase 0": the workgroup's pixels are then a 2-D tile (TN rows x 16 columns) of one image, the input halo
// (TN+2) x 18 pixels comes into LDS by DMA, the 3x3 conv's weights lead the wave's stream ([tap][k-step][accumulator]),
// and its 8-bit output tile stays in LDS as the first 1x1 conv's B operand - that edge never reaches memory.
// NW: waves per workgroup (4, or 8 = two per SIMD: one wave's epilogue arithmetic overlaps the other's weight stream; G1 /
// G2 count the channel groups PER WAVE, so K1 = 64 * NW * G1)
// S0: stride of the leading 3x3 conv (1, or 2 = the form the reference's stride-up pass leaves in the last block of a stage:
// 3x3 / stride 2, then 1x1 + eltwise whose shortcut is the previous block's output sub-sampled by the same stride — the
// absorbed `<split>_pool`, ChainKArgs::res_sub). The halo is then (2 TN + 1) x 33 input pixels and tap (dy, dx) of output
// pixel (j, col) sits at halo pixel (2 j + dy, 2 col + dx).
template <int KS1, int G1, int MFG2, int G2, int TN, int R, bool C3, int SP, bool HAS2 = true, int NW = 4, int S0 = 1>
__global__ __launch_bounds__(64 * NW) void conv1x1_chain_kernel(const ChainKArgs a) {
    static_assert(NW == 4 || NW == 8, "4 or 8 waves");
    static_assert(S0 == 1 || (C3 && S0 == 2), "a strided head is led by its 3x3 conv");
    constexpr int KS2 = NW * G1;                      // K1 / 64
    constexpr int K1 = 64 * NW * G1, C1 = 64 * KS1, K2W = NW * G2 * 16 * MFG2, K2 = K2W * SP;   // K2W: this workgroup's share
    constexpr int NPX = 16 * TN;
    constexpr int MF0 = C3 ? C1 / (16 * NW) : 1;      // 3x3 conv: C1 / NW channels per wave = MF0 accumulators
    static_assert(!C3 || C1 % (16 * NW) == 0, "3x3 conv: at least 16 channels per wave");
    constexpr int T0 = C3 ? MF0 * 9 * KS1 : 0;        // steps of the 3x3 conv per wave
    constexpr int SG1 = KS1 * 4, SG2 = KS2 * MFG2;    // steps per channel group
    constexpr int T1 = G1 * SG1, T2 = HAS2 ? G2 * SG2 : 0;   // steps per wave (HAS2 = false: no second 1x1 conv)
    constexpr int OFF = T0 % R;                       // ring slot of the first 1x1 conv's first step
    constexpr int CPR = K1 / 16;                      // 16-byte chunks per tile row
    constexpr int CH1 = C1 / 16;                      // ... per pixel of the 3x3 conv's input / output
    // LDS pixel pitch of the halo and of the 3x3 conv's output tile: one chunk of padding per pixel makes the 16 lanes of a
    // fragment column (consecutive pixels, same chunk) hit 16 different 16-byte bank groups (pitch odd in chunks); with the
    // natural pitch (4 / 8 / 16 chunks) they collide 4- / 8- / 16-way (SQ_LDS_BANK_CONFLICT: 58 % of the LDS cycles at C = 128)
    constexpr int PCH = CH1 + 1;
    constexpr int HW = 16 * S0 + 3 - S0, HP = ((TN - 1) * S0 + 3) * HW;        // halo: 18 x (TN + 2) pixels at stride 1, 33 x (2 TN + 1) at 2
    constexpr int HCH = C3 ? (HP * PCH + 63) / 64 * 64 : 1;
    constexpr int P0C = C3 ? (C1 / 4 * 3 + 63) / 64 * 64 : 1;
    constexpr int P1C = K1 / 4 * 3, P2C = (K2 / 4 * 3 + 63) / 64 * 64;
    static_assert(SG1 % R == 0 && SG2 % R == 0, "ring depth must divide the group lengths");
    static_assert(CPR >= 16 && P1C % 64 == 0, "tile rows are swizzled on 16 chunks");
    static_assert(SP == 1 || (!C3 && (NPX * CPR / (64 * NW)) % SP == 0), "split second conv: 1x1 chains only");

    __shared__ v4i tile[NPX * CPR];
    __shared__ v4i prm1[P1C];
    __shared__ v4i prm2[P2C];
    __shared__ v4i halo[HCH];
    __shared__ v4i mid[C3 ? NPX * PCH : 1];
    __shared__ v4i prm0[P0C];
    SABER_TL_DECL;
    SABER_TL(0);
    asm volatile("" ::"s"(a.x), "s"(a.wstream), "s"(a.res), "s"(a.prm1), "s"(a.prm2), "s"(a.M), "s"(a.in_u8));
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fq = lane >> 4;
    // ---- this workgroup's pixels: a run of NPX pixels, or (C3) rows y0 .. y0+TN-1 x columns x0 .. x0+15 of image n
    // SP == 2: two workgroups per pixel tile; both run the first conv, each computes HALF of the second conv's output
    // channels and stores half of the first conv's tile (a workgroup's time is its weight stream: 1/4 less to pull)
    const int half = SP == 1 ? 0 : (int)(blockIdx.x % SP);
    const int p0 = (SP == 1 ? blockIdx.x : blockIdx.x / SP) * NPX;
    const int plast = a.M - 1;
    int n = 0, y0 = 0, x0 = 0;
    if constexpr (C3) {
        asm volatile("" ::"s"(a.prm0), "s"(a.zero), "s"(a.H), "s"(a.W), "s"(a.tiles_x), "s"(a.tiles_per_img), "s"(a.mg_tiles_x),
                     "s"(a.mg_tpi));
        const int t = blockIdx.x;
        n = a.mg_tpi ? (int)__umulhi((unsigned)t, a.mg_tpi) : t;
        const int rem = t - n * a.tiles_per_img;
        const int ty = a.mg_tiles_x ? (int)__umulhi((unsigned)rem, a.mg_tiles_x) : rem;
        y0 = ty * TN;
        x0 = (rem - ty * a.tiles_x) * 16;
    }
    auto pix = [&](int px, bool& ok) -> int {          // tile pixel -> index into the [M][channels] tensors (clamped)
        if constexpr (!C3) {
            const int p = p0 + px;
            ok = p < a.M;
            return p < plast ? p : plast;
        } else {
            int y = y0 + (px >> 4), x = x0 + (px & 15);
            ok = y < a.H && x < a.W;
            y = y < a.H ? y : a.H - 1;
            x = x < a.W ? x : a.W - 1;
            return (n * a.H + y) * a.W + x;
        }
    };

    // ---- first 1x1 conv's B operand: the pixel tile of x straight into registers (C3: produced by phase 0 instead)
    v4i bx[KS1][TN];
    if constexpr (!C3) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            bool ok;
            const int p = pix(j * 16 + frow, ok);      // pixels beyond the tensor repeat the last one (never stored)
            const char* xp = (const char*)a.x + (size_t)p * C1 + fq * 16;
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) bx[ks][j] = *(const v4i*)(xp + ks * 64);
        }
    }

    // ---- LDS by DMA: (C3) the 3x3 conv's input halo, the residual tile, the per-channel constants ----------------
    // tile[px][chunk ^ (px & 15)]: the epilogue's 16-lane column accesses and the second conv's operand reads hit 16
    // different bank groups. One instruction moves 64 chunks; lane L of instruction i lands in chunk i*64 + L.
    // Issued BEFORE the weight ring: vector-memory loads return in order, so by the time the first MFMA has its weights
    // this wave's DMA has landed.
    {
        if constexpr (C3) {
            const char* xg = (const char*)a.x;
            for (int i = wave; i < HCH / 64; i += NW) {
                const int L = i * 64 + lane;
                const int hp = L / PCH, cc = L - hp * PCH;            // cc == CH1: the padding chunk (fetches zeros)
                const int hy = hp / HW, hx = hp - hy * HW;
                const int gy = y0 * S0 - 1 + hy, gx = x0 * S0 - 1 + hx;
                const int H0 = S0 == 1 ? a.H : a.H0, W0 = S0 == 1 ? a.W : a.W0;     // the 3x3 conv's INPUT dims
                const bool in = hp < HP && cc < CH1 && (unsigned)gy < (unsigned)H0 && (unsigned)gx < (unsigned)W0;   // zero padding
                const char* src = in ? xg + ((size_t)((n * H0 + gy) * W0 + gx) * C1 + cc * 16) : (const char*)a.zero;
                lds_dma16(src, halo + i * 64);
            }
            for (int i = wave; i < P0C / 64; i += NW) lds_dma16((const v4i*)a.prm0 + i * 64 + lane, prm0 + i * 64);
        }
        const char* rg = (const char*)a.res;
        constexpr int NI = NPX * CPR / 64;
#pragma unroll
        for (int i0 = 0; i0 < NI; i0 += NW) {
            const int i = i0 + wave;
            const int L = i * 64 + lane;
            const int px = L / CPR, c = (L % CPR) ^ (px & 15);
            bool ok;
            int p = pix(px, ok);
            if constexpr (S0 > 1) {      // the shortcut is [N][res_H][res_W][K1]: pixel (y * res_sub, x * res_sub) of it
                const int yy = min(y0 + (px >> 4), a.H - 1), xx = min(x0 + (px & 15), a.W - 1);
                p = (n * a.res_H + yy * a.res_sub) * a.res_W + xx * a.res_sub;
            }
            lds_dma16(rg + (size_t)p * K1 + c * 16, tile + i * 64);
        }
        for (int i = wave; i < P1C / 64; i += NW) lds_dma16((const v4i*)a.prm1 + i * 64 + lane, prm1 + i * 64);
        if constexpr (HAS2)
            for (int i = wave; i < P2C / 64; i += NW) lds_dma16((const v4i*)a.prm2 + i * 64 + lane, prm2 + i * 64);
    }
    asm volatile("" ::: "memory");                     // keep the ring's loads behind the DMA in program order

    // ---- weight ring: the first R steps of this wave's stream ---------------------------------------------------
    const v4i* wsb = (const v4i*)a.wstream + (size_t)(half * NW + wave) * ((T0 + T1 + T2) * 64);
    v4i ring[R];
#pragma unroll
    for (int r = 0; r < R; ++r) ring[r] = wsb[r * 64 + lane];
    wsb += R * 64;                                     // step s of the current phase / group refills from wsb[s * 64 + lane]

    if constexpr (C3) {
        // ============= phase 0: the 3x3 conv on the halo, output tile -> LDS ========================================
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(R) : "memory");   // everything older than the ring: this wave's DMA
        __builtin_amdgcn_s_barrier();                              // ... and every other wave's
        SABER_TL(1);
        const int xm0 = a.in0_u8 ? (int)0x80808080u : 0;
        // accumulators start at the per-channel compensation (exact integer sum, any order): one add per output saved
        const int c0 = wave * (C1 / NW) + fq * (4 * MF0);
        const v4i* pp = prm0 + (c0 / 4) * 3;
        v4i acc[MF0][TN];
#pragma unroll
        for (int mf = 0; mf < MF0; ++mf)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[mf][j] = pp[mf * 3 + 2];
        const v4i* hb = halo + (frow * S0) * PCH + fq;
#pragma unroll
        for (int s = 0; s < T0; ++s) {                 // steps ordered [tap][k-step][accumulator]
            const int mf = s % MF0, ks = (s / MF0) % KS1, tap = s / (MF0 * KS1);
            const int dy = tap / 3, dx = tap % 3;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                v4i b = hb[((j * S0 + dy) * HW + dx) * PCH + ks * 4];
                b.x ^= xm0; b.y ^= xm0; b.z ^= xm0; b.w ^= xm0;
                acc[mf][j] = mma_step(ring[s % R], b, acc[mf][j]);
            }
            ring[s % R] = wsb[s * 64 + lane];
        }
        wsb += T0 * 64;
        // epilogue: lane = 4*MF0 consecutive channels c0.. of pixel j*16 + frow -> mid[px][C1] (bytes)
        const float lo0 = a.relu0 ? 0.f : -3.0e38f;
        const float off0 = a.in_u8 ? 0.f : 128.f;      // a.in_u8: dtype of the 3x3 conv's output = first 1x1 conv's input
        const unsigned xo0 = a.in_u8 ? 0u : 0x80808080u;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            unsigned o[MF0];
#pragma unroll
            for (int mf = 0; mf < MF0; ++mf)
                o[mf] = chain_out_pack(acc[mf][j], v4i{0, 0, 0, 0}, __builtin_bit_cast(v4f, pp[mf * 3 + 1]),
                                       __builtin_bit_cast(v4f, pp[mf * 3]), lo0, off0, xo0);
            char* mp = (char*)mid + (j * 16 + frow) * (PCH * 16) + c0;
            if constexpr (MF0 == 1) *(unsigned*)mp = o[0];
            else if constexpr (MF0 == 2) *(uint2*)mp = make_uint2(o[0], o[1]);
            else *(uint4*)mp = make_uint4(o[0], o[1], o[2], o[3]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) bx[ks][j] = mid[(j * 16 + frow) * PCH + ks * 4 + fq];
    }

    const int xmask = a.in_u8 ? (int)0x80808080u : 0;
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            bx[ks][j].x ^= xmask; bx[ks][j].y ^= xmask; bx[ks][j].z ^= xmask; bx[ks][j].w ^= xmask;
        }
    const float lo_s8 = a.relu1 ? 0.f : -128.f;
    const float res_lo = a.res_relu ? 0.f : -3.0e38f;
    if constexpr (!C3) SABER_TL(1);

    // ================= first 1x1 conv: groups of 64 channels, steps ordered [ks][mf] ===============================
#pragma unroll 1
    for (int g = 0; g < G1; ++g) {
        v4i acc[4][TN];
        const int cg = wave * (K1 / NW) + g * 64 + fq * 16;   // lane = 16 consecutive channels cg .. cg+15 of pixels j*16 + frow
        const v4i* pp = prm1 + (cg / 4) * 3;
#pragma unroll
        for (int mf = 0; mf < 4; ++mf)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if constexpr (C3) acc[mf][j] = pp[mf * 3 + 2];   // constants are in LDS since the barrier before phase 0
                else acc[mf][j] = v4i{0, 0, 0, 0};               // (1x1 chain: their DMA is still in flight; added in the epilogue)
            }
#pragma unroll
        for (int s = 0; s < SG1; ++s) {
            const int ri = (OFF + s) % R;
            const int ks = s / 4, mf = s % 4;
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[mf][j] = mma_step(ring[ri], bx[ks][j], acc[mf][j]);
            if (HAS2 || s + R < SG1 || g + 1 < G1) ring[ri] = wsb[s * 64 + lane];   // the stream runs on into the second conv's weights
        }
        wsb += SG1 * 64;
        if (!C3 && g == 0) {
            __builtin_amdgcn_s_barrier();              // every wave's DMA has landed (see above)
        }
        if (g == 0) SABER_TL(2);
        // epilogue
        v4f sc[4], bi[4];
        v4i co[4];
#pragma unroll
        for (int mf = 0; mf < 4; ++mf) {
            sc[mf] = __builtin_bit_cast(v4f, pp[mf * 3]);
            bi[mf] = __builtin_bit_cast(v4f, pp[mf * 3 + 1]);
            if constexpr (C3) co[mf] = v4i{0, 0, 0, 0};
            else co[mf] = pp[mf * 3 + 2];
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            v4i* tp = tile + (j * 16 + frow) * CPR + ((cg / 16) ^ frow);
            const v4i rs = *tp;
            v4i o;
            o.x = (int)chain_elt_pack(acc[0][j], co[0], bi[0], sc[0], (unsigned)rs.x, lo_s8, res_lo, a);
            o.y = (int)chain_elt_pack(acc[1][j], co[1], bi[1], sc[1], (unsigned)rs.y, lo_s8, res_lo, a);
            o.z = (int)chain_elt_pack(acc[2][j], co[2], bi[2], sc[2], (unsigned)rs.z, lo_s8, res_lo, a);
            o.w = (int)chain_elt_pack(acc[3][j], co[3], bi[3], sc[3], (unsigned)rs.w, lo_s8, res_lo, a);
            *tp = o;
        }
    }
    SABER_TL(3);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                      // the tile now holds the first conv's s8 output, all channels

    // ---- tile -> y1 (coalesced) and -> the second conv's B operand ---------------------------------------------------
    {
        char* yg = (char*)a.y1;
        constexpr int NIT = NPX * CPR / (64 * NW);
#pragma unroll
        for (int it = half; it < NIT; it += SP) {
            const int L = it * (64 * NW) + tid;
            const int px = L / CPR, c = (L % CPR) ^ (px & 15);
            bool ok;
            const int p = pix(px, ok);
            if (ok) *(v4i*)(yg + (size_t)p * K1 + c * 16) = tile[L];
        }
    }
    if constexpr (!HAS2) {
        SABER_TL(4);
        SABER_TL_FLUSH();
        return;
    }
    v4i b2[KS2][TN];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int ks = 0; ks < KS2; ++ks) b2[ks][j] = tile[(j * 16 + frow) * CPR + ((ks * 4 + fq) ^ frow)];
    SABER_TL(4);

    // ================= second 1x1 conv ==============================================================================
    // a.k2_split > 0: the "second conv" is a sibling PAIR (two 1x1 convs over the tile, rows [0, k2_split) -> y2, the rest -> y2b with
    // their own relu / output type; k2_split is a multiple of a group's 16 * MFG2 channels, so the side is wave-uniform per group)
#pragma unroll 1
    for (int g = 0; g < G2; ++g) {
        v4i acc[MFG2][TN];
        const int cg = half * K2W + wave * (K2W / NW) + g * (16 * MFG2) + fq * (4 * MFG2);
        const bool side_b = a.k2_split > 0 && wave * (K2W / NW) + g * (16 * MFG2) >= a.k2_split;
        const float lo2 = (side_b ? a.relu2b : a.relu2) ? 0.f : -3.0e38f;
        const float off2 = (side_b ? a.out_u8_2b : a.out_u8_2) ? 0.f : 128.f;
        const unsigned xm2 = (side_b ? a.out_u8_2b : a.out_u8_2) ? 0u : 0x80808080u;
        const int kout = a.k2_split > 0 ? (side_b ? K2 - a.k2_split : a.k2_split) : K2;
        char* yout = side_b ? (char*)a.y2b - a.k2_split : (char*)a.y2;      // (- k2_split: cg counts from the pair's first row)
        const v4i* pp = prm2 + (cg / 4) * 3;
#pragma unroll
        for (int mf = 0; mf < MFG2; ++mf)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[mf][j] = pp[mf * 3 + 2];   // start at the compensation (see phase 0)
#pragma unroll
        for (int s = 0; s < SG2; ++s) {
            const int ri = (OFF + s) % R;
            const int ks = s / MFG2, mf = s % MFG2;
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[mf][j] = mma_step(ring[ri], b2[ks][j], acc[mf][j]);
            if (s + R < SG2 || g + 1 < G2) ring[ri] = wsb[s * 64 + lane];   // the stream ends with the last group
        }
        wsb += SG2 * 64;
        unsigned o[MFG2];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
#pragma unroll
            for (int mf = 0; mf < MFG2; ++mf)
                o[mf] = chain_out_pack(acc[mf][j], v4i{0, 0, 0, 0}, __builtin_bit_cast(v4f, pp[mf * 3 + 1]),
                                       __builtin_bit_cast(v4f, pp[mf * 3]), lo2, off2, xm2);
            bool ok;
            const int p = pix(j * 16 + frow, ok);
            if (ok) {
                char* y = yout + (size_t)p * kout + cg;
                if constexpr (MFG2 == 1) *(unsigned*)y = o[0];
                else if constexpr (MFG2 == 2) *(uint2*)y = make_uint2(o[0], o[1]);
                else *(uint4*)y = make_uint4(o[0], o[1], o[2], o[3]);
            }
        }
    }
    SABER_TL(5);
    SABER_TL_FLUSH();
}

bool conv1x1_chain_ok(int c1, int k1, int k2) {
    return (c1 == 64 || c1 == 128 || c1 == 256 || c1 == 512) && k1 == 4 * c1 && k2 == c1;
}

int conv1x1_chain_tn(int c1, int m) {
    (void)m;
    switch (c1) {
    case 64: return 4;
    case 128: return 2;
    case 256: case 512: return 1;
    default: return 0;
    }
}

// with3x3: a.x is the 3x3 conv's input and the grid is tiles of tn rows x 16 columns (a.H, a.W, a.tiles_* set by api.hip)
// tile: pixel fragments per workgroup (1x1 chain: 16-pixel runs; with3x3: rows of a 16-column tile) | 8 when the second
// conv's output channels are split over two workgroups (wstream then holds [half][wave] streams, api.hip)
hipError_t launch_conv1x1_chain(const ChainKArgs& a, int c1, int k1, int k2, int tile, int with3x3, hipStream_t s) {
    const bool has2 = k2 != 0;          // k2 == 0: conv3x3 + first 1x1 conv only (with3x3 required)
    const bool pair2 = has2 && a.k2_split > 0;      // the strided head followed by the next stage's sibling pair: 64 -> 256 (+ sum) -> 512 | 128
    if (pair2 && !(a.s0 == 2 && with3x3 && c1 == 64 && k1 == 256 && k2 == 640 && a.k2_split % 32 == 0 && a.k2_split < 640)) return hipErrorInvalidValue;
    if (!conv1x1_chain_ok(c1, k1, (has2 && !pair2) ? k2 : c1) || a.M <= 0 || (!has2 && !with3x3)) return hipErrorInvalidValue;
    // C >= 256 codes: 1 = one fragment, 9 = second conv split over two workgroups, 11 = split + 8 waves per workgroup
    // C = 128 codes: 2 | 1 fragments, + 4 = 8 waves per workgroup
    const bool w8 = (c1 >= 256 && (tile & 7) == 3) || (c1 == 128 && (tile & 4));
    const int tn = c1 == 128 ? tile & 3 : (w8 ? 1 : tile & 7), sp = (tile & 8) ? 2 : 1;
    const dim3 block(w8 ? 512 : 256);
    const dim3 grid((with3x3 ? a.tiles_per_img * a.N : (a.M + 16 * tn - 1) / (16 * tn)) * sp);
#define SABER_CHAIN(KS1, G1, MFG2, G2, TN, R, C3, SP, ...) \
    hipLaunchKernelGGL((conv1x1_chain_kernel<KS1, G1, MFG2, G2, TN, R, C3, SP, ##__VA_ARGS__>), grid, block, 0, s, a)
    // ring depths: measured with scripts/probe/timeline_probe.hip (chain): deeper rings (32 / 64 steps, or the whole
    // stream in registers) only move the wait into the prologue - the stream is bound by the CU's vector-memory path
    // (~43 B/clk measured for these 1 KB-per-instruction loads), not by the latency of one round trip
    if (pair2) {
        switch (tile) {
        case 4: SABER_CHAIN(1, 1, 2, 5, 4, 4, true, 1, true, 4, 2); break;
        case 2: SABER_CHAIN(1, 1, 2, 5, 2, 4, true, 1, true, 4, 2); break;
        default: return hipErrorInvalidValue;
        }
        return hipGetLastError();
    }
    if (!has2 && a.s0 == 2) {      // strided head (the last block of a stage after the reference's stride-up)
        switch (c1 * 32 + tile) {
        case 64 * 32 + 4: SABER_CHAIN(1, 1, 1, 1, 4, 4, true, 1, false, 4, 2); break;
        case 64 * 32 + 2: SABER_CHAIN(1, 1, 1, 1, 2, 4, true, 1, false, 4, 2); break;
        case 128 * 32 + 2: SABER_CHAIN(2, 2, 2, 1, 2, 8, true, 1, false, 4, 2); break;
        case 128 * 32 + 1: SABER_CHAIN(2, 2, 2, 1, 1, 8, true, 1, false, 4, 2); break;
        case 256 * 32 + 1: SABER_CHAIN(4, 4, 4, 1, 1, 16, true, 1, false, 4, 2); break;
        case 256 * 32 + 3: SABER_CHAIN(4, 2, 2, 1, 1, 16, true, 1, false, 8, 2); break;     // 8 waves
        case 128 * 32 + 6: SABER_CHAIN(2, 1, 1, 1, 2, 8, true, 1, false, 8, 2); break;
        case 128 * 32 + 5: SABER_CHAIN(2, 1, 1, 1, 1, 8, true, 1, false, 8, 2); break;
        default: return hipErrorInvalidValue;
        }
        return hipGetLastError();
    }
    if (!has2) {
        switch (c1 * 32 + tile) {
        case 64 * 32 + 4: SABER_CHAIN(1, 1, 1, 1, 4, 4, true, 1, false); break;
        case 64 * 32 + 2: SABER_CHAIN(1, 1, 1, 1, 2, 4, true, 1, false); break;
        case 128 * 32 + 2: SABER_CHAIN(2, 2, 2, 1, 2, 8, true, 1, false); break;
        case 128 * 32 + 1: SABER_CHAIN(2, 2, 2, 1, 1, 8, true, 1, false); break;
        case 256 * 32 + 1: SABER_CHAIN(4, 4, 4, 1, 1, 16, true, 1, false); break;
        case 256 * 32 + 3: SABER_CHAIN(4, 2, 2, 1, 1, 16, true, 1, false, 8); break;        // 8 waves
        case 128 * 32 + 6: SABER_CHAIN(2, 1, 1, 1, 2, 8, true, 1, false, 8); break;
        case 128 * 32 + 5: SABER_CHAIN(2, 1, 1, 1, 1, 8, true, 1, false, 8); break;
        default: return hipErrorInvalidValue;
        }
        return hipGetLastError();
    }
    switch (c1 * 32 + tile * 2 + (with3x3 ? 1 : 0)) {
    case 64 * 32 + 4 * 2: SABER_CHAIN(1, 1, 1, 1, 4, 4, false, 1); break;
    case 64 * 32 + 2 * 2: SABER_CHAIN(1, 1, 1, 1, 2, 4, false, 1); break;
    case 128 * 32 + 2 * 2: SABER_CHAIN(2, 2, 2, 1, 2, 8, false, 1); break;
    case 128 * 32 + 1 * 2: SABER_CHAIN(2, 2, 2, 1, 1, 8, false, 1); break;
    case 256 * 32 + 1 * 2: SABER_CHAIN(4, 4, 4, 1, 1, 16, false, 1); break;
    case 256 * 32 + 9 * 2: SABER_CHAIN(4, 4, 2, 1, 1, 16, false, 2); break;     // second conv split over two workgroups
    case 256 * 32 + 11 * 2: SABER_CHAIN(4, 2, 1, 1, 1, 16, false, 2, true, 8); break;   // ... and 8 waves
    case 512 * 32 + 1 * 2: SABER_CHAIN(8, 8, 4, 2, 1, 16, false, 1); break;
    case 512 * 32 + 9 * 2: SABER_CHAIN(8, 8, 4, 1, 1, 16, false, 2); break;
    case 64 * 32 + 4 * 2 + 1: SABER_CHAIN(1, 1, 1, 1, 4, 4, true, 1); break;
    case 64 * 32 + 2 * 2 + 1: SABER_CHAIN(1, 1, 1, 1, 2, 4, true, 1); break;
    case 128 * 32 + 2 * 2 + 1: SABER_CHAIN(2, 2, 2, 1, 2, 8, true, 1); break;
    case 128 * 32 + 1 * 2 + 1: SABER_CHAIN(2, 2, 2, 1, 1, 8, true, 1); break;
    case 256 * 32 + 1 * 2 + 1: SABER_CHAIN(4, 4, 4, 1, 1, 16, true, 1); break;
    case 256 * 32 + 3 * 2 + 1: SABER_CHAIN(4, 2, 2, 1, 1, 16, true, 1, true, 8); break;   // 3x3-led, 8 waves: 32 / 128 / 32 channels per wave
    case 128 * 32 + 6 * 2: SABER_CHAIN(2, 1, 1, 1, 2, 8, false, 1, true, 8); break;      // 8 waves
    case 128 * 32 + 5 * 2: SABER_CHAIN(2, 1, 1, 1, 1, 8, false, 1, true, 8); break;
    case 128 * 32 + 6 * 2 + 1: SABER_CHAIN(2, 1, 1, 1, 2, 8, true, 1, true, 8); break;
    case 128 * 32 + 5 * 2 + 1: SABER_CHAIN(2, 1, 1, 1, 1, 8, true, 1, true, 8); break;
    default: return hipErrorInvalidValue;
    }
#undef SABER_CHAIN
    return hipGetLastError();
}

}  // namespace saber_mi355x
