// anakin_amd/csrc/conv3x3_img.h — INT8 3x3 / stride-1 convolution for the SMALL feature maps of a ResNet
// (28x28, 14x14, 7x7): image slabs resident in LDS, weights streamed straight into MFMA operand registers (gfx950).
//
// Why a third 3x3 kernel: at 14x14 / 7x7 the implicit-GEMM kernels are bound by the operand bytes each CU has to
// pull in (a 64x32 tile of res4's 3x3 re-gathers 221 KB per workgroup: every input pixel once per tap, every weight
// row once per pixel tile), and conv3x3_halo.h's fixed TH x 16 tile wastes half of a 14- or 7-wide row. Here a
// workgroup owns
//     IB images x RB output rows (all OW columns)  x  16 output channels,
// i.e. the smallest weight tile one MFMA takes (each weight byte is fetched by exactly ONE workgroup per slab, into
// registers, in the MFMA A-operand layout: no LDS round trip), while the slab's input halo
// [IB][RB+2][OW+2][C] is brought into LDS ONCE with global_load_lds (all requests of the workgroup in flight at
// the same time: one exposed memory latency) and every tap reads it at a shifted pixel offset (9x reuse). Pixels
// are packed 16 to an MFMA column group in linear (image, row, col) order — a lane computes its own halo
// address — so 14- and 7-wide rows waste nothing but the last partial group. res4 3x3 (14x14x256, batch 8):
// 73 KB per workgroup instead of 221 KB, 256 workgroups.
//
// The NW (4 or 8) waves split the reduction by 64-channel chunk (CW ways; each wave holds the 9 taps x NCH chunks of its
// weights in registers) and / or the pixel groups (PW = NW / CW ways); partial accumulators of the channel split
// are summed through LDS (integer sums: order independent, bit-exact). Same arithmetic, u8 handling (XOR 0x80 +
// comp) and epilogues as conv_igemm_impl.h. Reference role: the 3x3 layers of GemmX8S8S32XConv::sub_dispatch
// (gemm_x8s8s32x_conv.cpp:187-288), without its im2col.
#pragma once
#include "conv3x3_halo.h"

namespace saber_mi355x {

// LDS budget (16-byte chunks) by channel count / 64: halo [C/64][HPp][64 B], reused as the reduction scratch
template <int NCHUNK>
constexpr int img_lds_chunks() {
    return NCHUNK == 1 ? (24 * 1024) / 16 : (NCHUNK == 8 ? (96 * 1024) / 16 : (40 * 1024) / 16);
}
constexpr int IMG_TAB = 1024;  // upper bound on the halo pixels of a slab

// Measured with scripts/probe/timeline_probe.hip on res4's 3x3 at batch 8 (per-launch cost in a 40-launch chain; the
// implicit-GEMM kernel the autotuner used before: 7.5 us): first version (swizzled LDS, one ds_read ahead) 7.3 us =
// prologue 0.9 + DMA issue 1.0 + landing 0.45 + 63 ds_read/xor/MFMA steps 2.4 + reduce/epilogue 1.0 (+ launch);
// unswizzled LDS 6.3; LDS reads batched per tap 6.0. Tried and dropped: 8 waves per workgroup (7.9); register
// staging with the u8 XOR applied once at staging (6.0: what the MFMA loop gains the staging loses); row-linear
// register staging into a padded, conflict-free [pixel][C+16] image (6.7: the straight-line bound issues 20 loads per
// lane where 12 are needed).
template <int EK, int NW, int CW, int NCH, int GPW>
__global__ __launch_bounds__(64 * NW) void conv3x3_img_kernel(const ImgKArgs ia) {
    const ConvKArgs& a = ia.c;
    constexpr int PW = NW / CW;
    constexpr int NCHUNK = CW * NCH;
    constexpr int C = 64 * NCHUNK;
    constexpr int LCH = img_lds_chunks<NCHUNK>();
    constexpr int NPGMAX = PW * GPW;

    __shared__ v4i lds[LCH];
    SABER_TL_DECL;
    SABER_TL(0);
    pin_hot_args(a);
    asm volatile("" ::"s"(ia.ib), "s"(ia.rb), "s"(ia.nrs), "s"(ia.mg[0]), "s"(ia.mg[1]), "s"(ia.mg[2]), "s"(ia.mg[3]),
                 "s"(ia.mg[4]), "s"(ia.mg[5]), "s"(a.N), "s"(a.K));
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform (scalar)
    const int cw = wave % CW, pw = wave / CW;
    const int frow = lane & 15, fq = lane >> 4;

    // ---- which slab / which 16 output channels -----------------------------------------------------------------
    int slab, tile_ky;
    xcd_tile(a, slab, tile_ky);                       // a.npx = image groups x row slabs, a.nky = ceil(K / 16)
    const int IB = ia.ib, RB = ia.rb;
    const int nrs = ia.nrs;                           // row slabs per image = ceil(OH / RB)
    const int ig = nrs == 1 ? slab : (int)__umulhi((unsigned)slab, ia.mg[5]), rs = slab - ig * nrs;
    const int n0 = ig * IB, r0 = rs * RB;
    const int ibv = (a.N - n0) < IB ? (a.N - n0) : IB;
    const int rbv = (a.OH - r0) < RB ? (a.OH - r0) : RB;
    const int HWd = a.OW + 2;                         // halo width
    const int HPI = (RB + 2) * HWd;                   // halo pixels per image
    const int HP = IB * HPI;
    const int HPp = (HP + 15) & ~15;                  // per-chunk pitch: a DMA instruction covers 16 halo pixels
    const int NPX = ibv * rbv * a.OW;                 // valid output pixels of the slab
    const int NPG = (NPX + 15) >> 4;
    const int k0 = tile_ky * 16;
    const unsigned m_hpi = ia.mg[0], m_hwd = ia.mg[1], m_ow = ia.mg[2], m_rowpx = rbv == RB ? ia.mg[3] : ia.mg[4];
    const int rowpx = rbv * a.OW;

    // ---- weights: 9 taps x NCH chunks, global -> registers in the MFMA A layout (row = lane & 15, k-group = lane >> 4)
    v4i wf[NCH][9];
    {
        const char* wrow = (const char*)a.w + (size_t)(k0 + frow) * a.Kg_pad + fq * 16;
#pragma unroll
        for (int nc = 0; nc < NCH; ++nc)
#pragma unroll
            for (int t = 0; t < 9; ++t) wf[nc][t] = *(const v4i*)(wrow + t * C + (cw * NCH + nc) * 64);
    }
    SABER_TL(1);

    // ---- halo -> LDS by DMA: one instruction = 16 halo pixels x 64 B of one channel chunk; lane L lands in 16-byte slot
    // (L & 3) of pixel (L >> 2). The LDS image is NOT swizzled: the MFMA loop of this kernel is bound by instruction issue
    // (one wave per SIMD, ~12 instructions per MFMA with the swizzle arithmetic), not by LDS bandwidth - a plain
    // [pixel][64 B] layout makes a fragment address `lane base + wave-uniform tap offset` (one add per fragment instead of
    // six operations) and the 2..4-way bank conflicts of its ds_read_b128 (8-16 LDS cycles) hide behind the MFMA.
    // Out-of-image halo pixels (zero padding, rows of a partial slab) fetch the zero page.
    {
        const char* xg = (const char*)a.x;
        const char* zero = (const char*)a.zero;
        const int hl = lane >> 2, physq = lane & 3;
        for (int hpb = wave * 16; hpb < HPp; hpb += 16 * NW) {
            const int hp = hpb + hl;
            const int img = (int)__umulhi((unsigned)hp, m_hpi), rem = hp - img * HPI;
            const int hy = (int)__umulhi((unsigned)rem, m_hwd), hx = rem - hy * HWd;
            const int iy = r0 - a.pad_h + hy, ix = hx - a.pad_w;
            const bool ok = hp < HP && img < ibv && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            const char* src0 = xg + (size_t)((((n0 + img) * a.H + iy) * a.W + ix) * C + physq * 16);
#pragma unroll
            for (int cc = 0; cc < NCHUNK; ++cc) {
                const char* src = ok ? src0 + cc * 64 : zero;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(lds + (cc * HPp + hpb) * 4),
                                                 16, 0, 0);
            }
        }
    }
    SABER_TL(2);

    const int kb = k0 + fq * 4;
    ChanParams<4> cp;
    load_chan_params<4>(a, kb, cp);

    // ---- this wave's pixel groups: halo pixel of tap (0,0) per lane ---------------------------------------------
    const int ng = (NPG - pw * GPW) < 0 ? 0 : ((NPG - pw * GPW) > GPW ? GPW : (NPG - pw * GPW));
    int hp0[GPW];
#pragma unroll
    for (int j = 0; j < GPW; ++j) {
        int p = (pw * GPW + j) * 16 + frow;
        p = p < NPX ? p : NPX - 1;                   // lanes beyond the slab read a valid pixel; their results are dropped
        const int img = (int)__umulhi((unsigned)p, m_rowpx), rem = p - img * rowpx;
        const int row = (int)__umulhi((unsigned)rem, m_ow), col = rem - row * a.OW;
        hp0[j] = ((img * (RB + 2) + row) * HWd + col) * 4 + fq;   // 16-byte slot of (pixel of tap (0,0), k-group fq), chunk 0
    }
    v4i acc[GPW];
#pragma unroll
    for (int j = 0; j < GPW; ++j) acc[j] = v4i{0, 0, 0, 0};
    const int xmask = a.in_u8 ? (int)0x80808080u : 0;

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    SABER_TL(3);
    __syncthreads();                                  // every wave's DMA has landed
    SABER_TL(4);

#pragma unroll
    for (int nc = 0; nc < NCH; ++nc) {
        const int base = (cw * NCH + nc) * HPp;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int toff = (base + (t / 3) * HWd + (t % 3)) * 4;   // wave-uniform slot offset of this (chunk, tap)
            // straight-line over all GPW groups: groups beyond this wave's share (j >= ng) multiply a clamped, valid pixel
            // and are dropped afterwards (a wave-uniform `if (j < ng)` here makes hipcc shuffle every accumulator through
            // the branch: 150 moves per MFMA, measured 17 us for 63 MFMAs)
            // all GPW fragments of the tap are requested before the first is used: the LDS latency (~64-128 cycles) is paid
            // once per tap, not once per MFMA (hipcc keeps only one ds_read ahead when they are issued one by one)
            v4i bf[GPW];
#pragma unroll
            for (int j = 0; j < GPW; ++j) bf[j] = lds[hp0[j] + toff];
#pragma unroll
            for (int j = 0; j < GPW; ++j) {
                v4i b = bf[j];
                b.x ^= xmask; b.y ^= xmask; b.z ^= xmask; b.w ^= xmask;
                acc[j] = mma_step(wf[nc][t], b, acc[j]);
            }
        }
    }

    SABER_TL(5);
    // ---- epilogue of one pixel group: lane = channels kb..kb+3 of pixel g*16 + frow -------------------------------
    auto finish = [&](int g, const v4i& s) {
        const int pl = g * 16 + frow;
        if (pl >= NPX || kb >= a.K) return;
        const int img = (int)__umulhi((unsigned)pl, m_rowpx), rem = pl - img * rowpx;
        const int row = (int)__umulhi((unsigned)rem, m_ow), col = rem - row * a.OW;
        const int p = ((n0 + img) * a.OH + r0 + row) * a.OW + col;
        int v[4] = {s.x, s.y, s.z, s.w};
        if constexpr (EK == EK_GEN) {
            epilogue_i8<4>(a, v, cp, p, kb);
        } else {
            epilogue_i8_fast<4, EK>(a, v, cp, p, kb);   // K % 16 == 0 here (epilogue_kind)
        }
    };

    if constexpr (CW == 1) {
#pragma unroll
        for (int j = 0; j < GPW; ++j)
            if (j < ng) finish(pw * GPW + j, acc[j]);
    } else {
        // channel-split partial sums through LDS (the halo is dead once every wave has left the MFMA loop)
        __syncthreads();
#pragma unroll
        for (int j = 0; j < GPW; ++j) lds[(cw * NPGMAX + pw * GPW + j) * 64 + lane] = acc[j];
        __syncthreads();
        for (int g = wave; g < NPG; g += NW) {
            v4i s = lds[g * 64 + lane];
#pragma unroll
            for (int c = 1; c < CW; ++c) s += lds[(c * NPGMAX + g) * 64 + lane];
            finish(g, s);
        }
    }
    SABER_TL(6);
    SABER_TL_FLUSH();
}

// Host-side feasibility of (nw, ib, rb) for a layer; also returns the kernel shape. C in {64,128,256,512}.
struct ImgShape {
    int cw, nch, gpw;
};
static inline bool img_shape(int C, int OW, int OH, int N, int nw, int ib, int rb, ImgShape* out) {
    if (ib < 1 || rb < 1 || rb > OH || ib > N || OW < 2) return false;   // (OW >= 2: every magic divisor is >= 2)
    if (nw != 4) return false;   // (an 8-wave variant - two waves per SIMD - was measured slower on every layer: removed)
    int cw, nch, lbytes;
    switch (C) {
    case 64: cw = 1; nch = 1; lbytes = 24 * 1024; break;
    case 128: cw = 2; nch = 1; lbytes = 40 * 1024; break;
    case 256: cw = 4; nch = 1; lbytes = 40 * 1024; break;
    case 512: cw = 4; nch = 2; lbytes = 96 * 1024; break;
    default: return false;
    }
    const int pw = nw / cw;
    const int hp = ib * (rb + 2) * (OW + 2);
    const int hpp = (hp + 15) & ~15;
    if (hpp > IMG_TAB) return false;
    if ((size_t)(C / 64) * hpp * 64 > (size_t)lbytes) return false;   // halo [C/64][hpp][64 B]
    const int npg = (ib * rb * OW + 15) / 16;
    const int per = (npg + pw - 1) / pw;
    int gpw;
    if (nw == 4) {
        if (per > 7) return false;
        gpw = per <= 4 ? 4 : 7;
    } else {
        if (per > 4) return false;
        gpw = per <= 2 ? 2 : 4;
    }
    if (cw > 1 && (size_t)cw * pw * gpw * 1024 > (size_t)lbytes) return false;
    if (out) { out->cw = cw; out->nch = nch; out->gpw = gpw; }
    return true;
}

template <int EK>
static hipError_t launch_conv3x3_img_inst(const ConvKArgs& a, int nw, int ib, int rb, hipStream_t s) {
    ImgShape sh;
    if (!img_shape(a.C, a.OW, a.OH, a.N, nw, ib, rb, &sh)) return hipErrorInvalidValue;
    ImgKArgs b;
    b.c = a;
    b.ib = ib; b.rb = rb; b.nw = nw;
    b.nrs = (a.OH + rb - 1) / rb;
    b.c.npx = ((a.N + ib - 1) / ib) * b.nrs;
    b.c.nky = (a.K + 15) / 16;
    b.c.mg_npx = magic_div(b.c.npx, (long long)b.c.npx * b.c.nky);
    {   // ceil(2^32 / d): __umulhi(n, m) == n / d whenever n * d < 2^32 (here n < 2^16, d < 2^12) and d >= 2
        const int tail = a.OH % rb ? a.OH % rb : rb;
        const unsigned d[6] = {(unsigned)((rb + 2) * (a.OW + 2)), (unsigned)(a.OW + 2), (unsigned)a.OW, (unsigned)(rb * a.OW),
                               (unsigned)(tail * a.OW), (unsigned)b.nrs};
        for (int i = 0; i < 6; ++i) b.mg[i] = d[i] >= 2 ? (unsigned)((0x100000000ull + d[i] - 1) / d[i]) : 0u;
    }
    dim3 grid(b.c.npx * b.c.nky), block(64 * nw);
#define SABER_IMG_CASE(CW_, NCH_)                                                                                        \
    if (sh.cw == CW_ && sh.nch == NCH_) {                                                                                 \
        if (sh.gpw == 4) hipLaunchKernelGGL((conv3x3_img_kernel<EK, 4, CW_, NCH_, 4>), grid, block, 0, s, b);                   \
        else hipLaunchKernelGGL((conv3x3_img_kernel<EK, 4, CW_, NCH_, 7>), grid, block, 0, s, b);                               \
        return hipGetLastError();                                                                                         \
    }
    SABER_IMG_CASE(1, 1) SABER_IMG_CASE(2, 1) SABER_IMG_CASE(4, 1) SABER_IMG_CASE(4, 2)
#undef SABER_IMG_CASE
    return hipErrorInvalidValue;
}

}  // namespace saber_mi355x
