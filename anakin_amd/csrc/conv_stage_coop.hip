// anakin_amd/csrc/conv_stage_coop.hip - a RUN of ResNet res4 blocks (C = 256) as ONE persistent launch: for every block the 3x3-led
// chain of conv_chain_coop.hip (conv 3x3 C -> C, conv 1x1 C -> 4C + SaberEltwise sum + relu, next block's conv 1x1 4C -> C) with
// FOUR cooperating workgroups per tile of 2 rows x 16 columns, and between two blocks an XCD-local barrier instead of a kernel
// boundary.
//
// Why (round-3 verdict items 3 / 4; the in-kernel stamps of profiles/r04/timeline_coop*.txt). One chain launch at batch 8 is a
// LATENCY chain, not a throughput problem: 2.0 us from the kernel's start until the input halo is in LDS (the previous launch's output
// comes back from beyond the L2 after a kernel boundary), ~1 us per hand-off between the cooperating workgroups, 1.7 us of VALU work in
// the eltwise epilogue, and ~2.2 us between the end of one launch and the start of the next - for 3.5 GOP that the matrix cores would
// finish in 1.4 us. What a kernel boundary costs can be removed: with ALL tiles of an image on ONE XCD, the next block's halo (its
// neighbours' rows) is in that XCD's L2 when the image's workgroups have passed a barrier there, the block's own shortcut tile is still in
// LDS, and the next block's weights can be requested before the barrier.
//   workgroup b: XCD b % 8 (the placement api_conv.hip: xcd_round_robin verifies once per device; every workgroup also publishes its
//   XCC_ID and compares it with its image's), slot b / 8 -> image (slot / wpi) * 8 + XCD, tile and quarter from slot % wpi
//   (wpi = 4 x tiles per image = 28 at 14 x 14).  A single block (nblk = 1) may instead spread its tiles over all XCDs (per_image = 0):
//   the chain's tile code 15.
// Per block and workgroup (8 waves; q = its quarter of every convolution's output channels):
//   3x3 (64 channels)             wave = (n-tile w & 3) x (K half w >> 2), both tile rows per 1 KB weight fragment; halves summed in LDS
//   -> xch (global, through the L2) + LDS; ARRIVE 1; the quarter's own mid k-step; WAIT 1; the partners' k-steps (sc1 loads)
//   1x1 + eltwise (256 channels)  32 channels per wave; the shortcut tile is the previous block's output tile, still in LDS
//   -> y1 (the operator's own output tensor) + LDS; ARRIVE 2; own y1 k-steps; WAIT 2; the partners' 24 KB of the y1 tile by sc1 LDS-DMA
//      (ONE copy per workgroup: with per-wave loads the four n-tile waves of a K half each fetched the same rows, 96 KB per
//      workgroup through the L2 - 1.8 us of the four-workgroup chain's last phase)
//   1x1 (64 channels)             wave = n-tile x K half; halves summed in LDS -> y2
//   ARRIVE at the two edges of this tile row, WAIT there for the tile rows above and below (their y2 rows are the next block's halo);
//   the halo by sc1 LDS-DMA from y2
// Weight fragments: the 3x3's 18 + the first two of the 1x1 are requested one block AHEAD (after ARRIVE 2), the other 14 after
// ARRIVE 1 - nothing is in flight when a wave arrives (one vmcnt for loads and stores on gfx950). Integer sums throughout: the split
// and the order cannot change a bit. Counters are never reset: every barrier instance adds a multiple of 32.
#include "coop_sync.h"

namespace saber_mi355x {

template <int MAXB>
__global__ __launch_bounds__(512) void conv_stage4_c256_kernel(const Stage4KArgs<MAXB> ka) {
    constexpr int C1 = 256, K1 = 1024, K2 = 256, NW = 8;
    constexpr int F0 = 18, F1 = 8, F2 = 8;                   // 1 KB weight fragments per wave and phase
    constexpr int K0Q = C1 / 4, K1Q = K1 / 4, K2Q = K2 / 4;  // this workgroup's output channels per phase: 64, 256, 64
    constexpr int CH1 = C1 / 16, PCH = CH1 + 1, HW = 18, HP = 4 * HW;
    constexpr int HCH = (HP * PCH + 63) / 64 * 64;
    constexpr int CPRW = K1Q / 16;                           // 16-byte chunks per row of the quarter's shortcut / output tile: 16
    constexpr int P0C = (K0Q / 4 * 3 + 63) / 64 * 64, P1C = K1Q / 4 * 3, P2C = (K2Q / 4 * 3 + 63) / 64 * 64;
    constexpr int MPC = K0Q / 16 + 1;                        // LDS pitch (chunks) of this quarter's 3x3 output tile: 4 + 1 padding
    constexpr int YPC = (K1 - K1Q) / 16 + 1;                 // ... of the partners' part of the y1 tile: 48 + 1
    constexpr int YCH = (32 * YPC + 63) / 64 * 64;
    static_assert(P1C % 64 == 0 && (32 * CPRW) == 64 * NW, "DMA granularity");
    __shared__ v4i halo[HCH];
    __shared__ v4i tile[32 * CPRW];
    __shared__ v4i ptile[YCH];
    __shared__ v4i mid_own[32 * MPC];
    __shared__ v4i red[4 * 2 * 64];
    __shared__ v4i prm0[P0C];
    __shared__ v4i prm1[P1C];
    __shared__ v4i prm2[P2C];
    __shared__ int dma_off[((HCH / 64 + NW - 1) / NW + (YCH / 64 + NW - 1) / NW) * 512];
    SABER_TL_DECL;
    SABER_TL(0);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nt = wave & 3, kh = wave >> 2;
    const int frow = lane & 15, fq = lane >> 4;
    const int b = blockIdx.x;
    int n, rem, q;
    if (ka.per_image) {      // all workgroups of an image on one XCD
        const int slot = b >> 3;
        const int il = ka.mg_wpi ? (int)__umulhi((unsigned)slot, ka.mg_wpi) : slot;
        const int widx = slot - il * ka.tiles_per_img * 4;
        n = il * 8 + (b & 7);
        rem = widx >> 2;
        q = widx & 3;
        if (n >= ka.N) return;
    } else {                 // workgroups b, b + 8, b + 16, b + 24 (one XCD) are the quarters of tile (b / 32) * 8 + b % 8
        const int t = (b >> 5) * 8 + (b & 7);
        q = (b >> 3) & 3;
        if (t >= ka.N * ka.tiles_per_img) return;
        n = ka.mg_tpi ? (int)__umulhi((unsigned)t, ka.mg_tpi) : t;
        rem = t - n * ka.tiles_per_img;
    }
    const int t = n * ka.tiles_per_img + rem;                // global tile number: counters, exchange tile, XCC words
    const int ty = ka.mg_tiles_x ? (int)__umulhi((unsigned)rem, ka.mg_tiles_x) : rem;
    const int x0 = (rem - ty * ka.tiles_x) * 16, y0 = ty * 2;
    const int H = ka.H, W = ka.W;
    auto pix = [&](int m, bool& ok) -> int {                  // m = tile row * 16 + column
        int x = x0 + (m & 15), y = y0 + (m >> 4);
        ok = x < W && y < H;
        x = x < W ? x : W - 1;
        y = y < H ? y : H - 1;
        return (n * H + y) * W + x;
    };
    unsigned long long* const ctr = ka.grp_ctr + (size_t)t * 32;
    unsigned long long* const e_up = ka.img_ctr + ((size_t)n * (ka.tiles_per_img + 1) + ty) * 16;   // the edge above this tile row; + 16: below
    bool ok0, ok1, okt;
    const int p0 = pix(frow, ok0), p1 = pix(16 + frow, ok1);
    const int tpx = tid / CPRW, tc = (tid % CPRW) ^ (tpx & 15);          // this thread's chunk of the quarter's y1 tile
    const int pt = pix(tpx, okt);

    // This lane's source offsets of the two per-block DMA copies, computed ONCE (the instruction mix of the launch was 10 VALU per MFMA
    // instruction, a third of it the index arithmetic of these copies repeated every block): the input halo - 4 rows x 18 columns x 256
    // channels, -1 = the zero page (padding) - and the partners' part of the y1 tile (slot L = pixel * 49 + partner chunk, chunk 48 of
    // every pixel is padding).
    // (kept in LDS, one word per thread and copy instruction: in registers they pushed the kernel over 256 VGPRs)
    constexpr int HIT = (HCH / 64 + NW - 1) / NW, YIT = (YCH / 64 + NW - 1) / NW;
    int* const halo_off = dma_off + tid;               // [it * 512]
    int* const pt_off = dma_off + HIT * 512 + tid;
#pragma unroll
    for (int it = 0; it < HIT; ++it) {
        const int L = (wave + it * NW) * 64 + lane;
        const int hp = L / PCH, cc = L - hp * PCH;
        const int hy = hp / HW, hx = hp - hy * HW;
        const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
        const bool in = hp < HP && cc < CH1 && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
        halo_off[it * 512] = in ? ((n * H + gy) * W + gx) * C1 + cc * 16 : -1;
    }
#pragma unroll
    for (int it = 0; it < YIT; ++it) {
        const int L = (wave + it * NW) * 64 + lane;
        const int px = L / YPC, j = L - px * YPC;
        bool okp;
        const int pp_ = pix(px & 31, okp);
        const int pc = j < q * 16 ? j : j + 16;      // the partners' chunks in channel order, this quarter's 16 skipped
        pt_off[it * 512] = (px < 32 && j < YPC - 1) ? pp_ * K1 + pc * 16 : -1;
    }
    auto dma_halo = [&](const void* x, bool l2) {
        const char* xg = (const char*)x;
#pragma unroll
        for (int it = 0; it < HIT; ++it) {
            const int i = wave + it * NW;
            if (i >= HCH / 64) break;
            const int off = halo_off[it * 512];
            if (off < 0) lds_dma16(ka.zero, halo + i * 64);
            else if (l2) lds_dma16_l2(xg + off, halo + i * 64);
            else lds_dma16(xg + off, halo + i * 64);
        }
    };
    auto dma_prm = [&](const StageBlk& B) {
        if (wave < P0C / 64) lds_dma16((const v4i*)B.prm0 + q * (K0Q / 4 * 3) + wave * 64 + lane, prm0 + wave * 64);
        for (int i = wave; i < P1C / 64; i += NW) lds_dma16((const v4i*)B.prm1 + q * P1C + i * 64 + lane, prm1 + i * 64);
        if (wave < P2C / 64) lds_dma16((const v4i*)B.prm2 + q * (K2Q / 4 * 3) + wave * 64 + lane, prm2 + wave * 64);
    };
    auto stream_of = [&](const StageBlk& B) -> const v4i* {   // this (quarter, wave)'s fragments of block B, lane's 16 bytes
        return (const v4i*)B.wstream + (size_t)(q * NW + wave) * ((F0 + F1 + F2) * 64) + lane;
    };

    // ---- entry: the first block's halo, shortcut tile and constants by DMA; its first 20 weight fragments ------------------------
    dma_halo(ka.x, false);
    lds_dma16((const char*)ka.res + (size_t)pt * K1 + q * K1Q + tc * 16, tile + wave * 64);
    dma_prm(ka.blk[0]);
    asm volatile("" ::: "memory");
    v4i fr[F0 + 2];
    {
        const v4i* wsb = stream_of(ka.blk[0]);
#pragma unroll
        for (int r = 0; r < F0 + 2; ++r) {
            fr[r] = wsb[r * 64];
            asm volatile("" ::: "memory");                   // issue order = consumption order
        }
    }
    wait_vm_older_than<F0 + 2>();                             // everything older than the fragments: this wave's DMA
    __builtin_amdgcn_s_barrier();
    SABER_TL(1);

    StageBlk B = ka.blk[0];                                  // (by value: scalar registers for the whole block)
    for (int k = 0; k < ka.nblk; ++k) {
        const v4i z = {0, 0, 0, 0};
        // ================= phase 0: 3x3 conv, 16 mid channels x half of K per wave, both tile rows per fragment =====================
        {
            const int xm0 = B.in0_u8 ? (int)0x80808080u : 0;
            const int c0 = nt * 16 + fq * 4;                 // within this quarter
            const v4i* pp = prm0 + (c0 / 4) * 3;
            v4i acc0 = kh ? z : pp[2], acc1 = acc0;          // (the compensation enters once: an exact integer sum)
            const v4i* hb = halo + frow * PCH + fq + kh * 8;
#pragma unroll
            for (int s = 0; s < F0; ++s) {
                const int kl = s % 2, tap = s / 2;
                const int dy = tap / 3, dx = tap % 3;
                v4i b0 = hb[(dy * HW + dx) * PCH + kl * 4];
                v4i b1 = hb[((dy + 1) * HW + dx) * PCH + kl * 4];
                b0.x ^= xm0; b0.y ^= xm0; b0.z ^= xm0; b0.w ^= xm0;
                b1.x ^= xm0; b1.y ^= xm0; b1.z ^= xm0; b1.w ^= xm0;
                acc0 = mma_step(fr[s], b0, acc0);
                acc1 = mma_step(fr[s], b1, acc1);
            }
            if (kh) {
                red[(nt * 2 + 0) * 64 + lane] = acc0;
                red[(nt * 2 + 1) * 64 + lane] = acc1;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (!kh) {
                const v4i r0 = red[(nt * 2 + 0) * 64 + lane], r1 = red[(nt * 2 + 1) * 64 + lane];
                acc0.x += r0.x; acc0.y += r0.y; acc0.z += r0.z; acc0.w += r0.w;
                acc1.x += r1.x; acc1.y += r1.y; acc1.z += r1.z; acc1.w += r1.w;
                const float lo0 = B.relu0 ? 0.f : -3.0e38f;
                const float off0 = B.in_u8 ? 0.f : 128.f;
                const unsigned xo0 = B.in_u8 ? 0u : 0x80808080u;
                const unsigned o0 = chain_out_pack(acc0, z, __builtin_bit_cast(v4f, pp[1]), __builtin_bit_cast(v4f, pp[0]), lo0, off0, xo0);
                const unsigned o1 = chain_out_pack(acc1, z, __builtin_bit_cast(v4f, pp[1]), __builtin_bit_cast(v4f, pp[0]), lo0, off0, xo0);
                char* xg = (char*)ka.xch + ((size_t)t * 32 + frow) * C1 + q * K0Q + c0;         // for the partners (through the L2)
                *(unsigned*)xg = o0;
                *(unsigned*)(xg + 16 * C1) = o1;
                *(unsigned*)((char*)mid_own + frow * (MPC * 16) + c0) = o0;                      // for this workgroup (LDS)
                *(unsigned*)((char*)mid_own + (16 + frow) * (MPC * 16) + c0) = o1;
            }
        }
        if (k == 0 && tid == 0) {      // (stored HERE, not at entry: a store pending beside the loads makes the compiler wait for vmcnt(0))
            unsigned xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            ka.xcc[t * 32 + q] = xcc & 7u;
            if (rem == 0 && q == 0) ka.xcc[t * 32 + 4] = (xcc & 7u) + 1u;      // the image's first workgroup publishes its XCD (+ 1: 0 = not yet)
        }
        coop_arrive(ctr);
        // the rest of this block's stream, requested under the wait for the partners: 6 fragments of the first 1x1 conv, 8 of the second
        v4i fs[F1 - 2 + F2];
        {
            const v4i* wsb = stream_of(B);
#pragma unroll
            for (int r = 0; r < F1 - 2 + F2; ++r) {
                fs[r] = wsb[(F0 + 2 + r) * 64];
                asm volatile("" ::: "memory");
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                        // mid_own is complete
        SABER_TL(2);

        // ================= phase 1: 1x1 conv + eltwise, 32 channels per wave; the k-step over this quarter's mid channels first =======
        {
            const int xmask = B.in_u8 ? (int)0x80808080u : 0;
            const int cg = wave * 32 + fq * 8;               // within this quarter: 8 consecutive channels of pixel (row, frow)
            const v4i* pp = prm1 + (cg / 4) * 3;
            v4i acc[2][2];
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int mf = 0; mf < 2; ++mf) acc[m][mf] = pp[mf * 3 + 2];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                v4i bo = mid_own[(m * 16 + frow) * MPC + fq];
                bo.x ^= xmask; bo.y ^= xmask; bo.z ^= xmask; bo.w ^= xmask;
                acc[m][0] = mma_step(fr[F0], bo, acc[m][0]);
                acc[m][1] = mma_step(fr[F0 + 1], bo, acc[m][1]);
            }
            coop_wait<31ull>(ctr, ka.err);                   // the partners' quarters of the 3x3 tile are in the L2
            SABER_TL(3);
            const L2Reader xch_l2(ka.xch);
            v4i bp[3][2];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const unsigned mo = (unsigned)(((size_t)t * 32 + m * 16 + frow) * C1 + ((q + 1 + i) & 3) * 64 + fq * 16);
                    bp[i][m] = xch_l2.load16<SABER_COOP_AUX>(mo);
                }
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    v4i bv = bp[i][m];
                    bv.x ^= xmask; bv.y ^= xmask; bv.z ^= xmask; bv.w ^= xmask;
                    acc[m][0] = mma_step(fs[i * 2], bv, acc[m][0]);
                    acc[m][1] = mma_step(fs[i * 2 + 1], bv, acc[m][1]);
                }
            const float lo_s8 = B.relu1 ? 0.f : -128.f;
            const float res_lo = B.res_relu ? 0.f : -3.0e38f;
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                c2i* tp = (c2i*)((char*)tile + ((m * 16 + frow) * CPRW + ((cg / 16) ^ frow)) * 16 + (fq & 1) * 8);
                const c2i rs = *tp;
                c2i o;
                o.x = (int)chain_elt_pack(acc[m][0], z, __builtin_bit_cast(v4f, pp[1]), __builtin_bit_cast(v4f, pp[0]), (unsigned)rs.x, lo_s8, res_lo, B);
                o.y = (int)chain_elt_pack(acc[m][1], z, __builtin_bit_cast(v4f, pp[4]), __builtin_bit_cast(v4f, pp[3]), (unsigned)rs.y, lo_s8, res_lo, B);
                *tp = o;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (okt) *(v4i*)((char*)ka.y1[k] + (size_t)pt * K1 + q * K1Q + tc * 16) = tile[tid];      // this quarter's 32 x 256 tile -> y1, coalesced
        coop_arrive(ctr + 16);
        SABER_TL(4);
        const StageBlk Bn = ka.blk[k + 1 < ka.nblk ? k + 1 : k];      // the next block's constants, one phase ahead of their use
        if (k + 1 < ka.nblk) {         // the next block's first 20 fragments: one block ahead, behind the arrival
            const v4i* wsb = stream_of(Bn);
#pragma unroll
            for (int r = 0; r < F0 + 2; ++r) {
                fr[r] = wsb[r * 64];
                asm volatile("" ::: "memory");
            }
        }

        // ================= phase 2: second 1x1 conv, 16 channels x half of K per wave; this quarter's y1 channels (in LDS) first ========
        {
            const int c2 = nt * 16 + fq * 4;                 // within this quarter
            const v4i* pp = prm2 + (c2 / 4) * 3;
            v4i acc0 = kh ? z : pp[2], acc1 = acc0;
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {                 // k-steps kh * 2 + jj of this quarter's own 256 channels
                const int ch = ((kh * 2 + jj) * 4 + fq) ^ frow;
                acc0 = mma_step(fs[F1 - 2 + jj], tile[frow * CPRW + ch], acc0);
                acc1 = mma_step(fs[F1 - 2 + jj], tile[(16 + frow) * CPRW + ch], acc1);
            }
            coop_wait<31ull>(ctr + 16, ka.err);              // the partners' quarters of the y1 tile are in the L2
            SABER_TL(5);
            // ... into LDS, once per workgroup (offsets: pt_off above)
#pragma unroll
            for (int it = 0; it < YIT; ++it) {
                const int i = wave + it * NW;
                if (i >= YCH / 64) break;
                const int off = pt_off[it * 512];
                if (off >= 0) lds_dma16_l2((const char*)ka.y1[k] + off, ptile + i * 64);
                else lds_dma16(ka.zero, ptile + i * 64);
            }
            wait_vm_older_than<0>();
            __builtin_amdgcn_s_barrier();
#pragma unroll
            for (int jj = 0; jj < 6; ++jj) {                 // k-steps (q * 4 + 4 + kh * 6 + jj) % 16 of the y1 rows: the partners' channels
                const int ks = (q * 4 + 4 + kh * 6 + jj) & 15;
                const int pc = ks * 4 + fq;
                const int j = pc < q * 16 ? pc : pc - 16;
                acc0 = mma_step(fs[F1 + jj], ptile[frow * YPC + j], acc0);
                acc1 = mma_step(fs[F1 + jj], ptile[(16 + frow) * YPC + j], acc1);
            }
            if (kh) {
                red[(nt * 2 + 0) * 64 + lane] = acc0;
                red[(nt * 2 + 1) * 64 + lane] = acc1;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (!kh) {
                const v4i r0 = red[(nt * 2 + 0) * 64 + lane], r1 = red[(nt * 2 + 1) * 64 + lane];
                acc0.x += r0.x; acc0.y += r0.y; acc0.z += r0.z; acc0.w += r0.w;
                acc1.x += r1.x; acc1.y += r1.y; acc1.z += r1.z; acc1.w += r1.w;
                const float lo2 = B.relu2 ? 0.f : -3.0e38f;
                const float off2 = B.out_u8_2 ? 0.f : 128.f;
                const unsigned xm2 = B.out_u8_2 ? 0u : 0x80808080u;
                const unsigned o0 = chain_out_pack(acc0, z, __builtin_bit_cast(v4f, pp[1]), __builtin_bit_cast(v4f, pp[0]), lo2, off2, xm2);
                const unsigned o1 = chain_out_pack(acc1, z, __builtin_bit_cast(v4f, pp[1]), __builtin_bit_cast(v4f, pp[0]), lo2, off2, xm2);
                if (ok0) *(unsigned*)((char*)ka.y2[k] + (size_t)p0 * K2 + q * K2Q + c2) = o0;
                if (ok1) *(unsigned*)((char*)ka.y2[k] + (size_t)p1 * K2 + q * K2Q + c2) = o1;
            }
        }
        SABER_TL(6);
        if (k + 1 < ka.nblk) {
            // ============= between two blocks: the next halo = the y2 rows of this tile and of the tiles above and below it ==============
            // One counter per EDGE between two tile rows of the image (tiles_x = 1): every workgroup of the two tiles at an edge arrives there
            // when its y2 stores are in the L2 (8 arrivals; 4 at the image's top and bottom edge, where one tile takes part), and its
            // wave 0 waits at the tile's two edges (both polled in ONE loop) - a barrier among participants only (the counters are never reset: multiples of 8 / 4),
            // no image-wide skew. (One arrival and one poller per WORKGROUP: with every wave arriving and polling, 64 waves spinning on
            // the line the atomics go to made the five-block launch 71 us instead of 54.) Then the halo by sc1 DMA.
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                                // (every wave is also done with this block's constants)
            // The counter of an edge alternates with the block's PARITY (two words of the edge's line): arrival is unconditional and comes before
            // the wait, so with ONE word a tile row running a block ahead - the dependency structure allows rows two apart to be two blocks
            // apart - could re-arrive at a shared edge before this row had sampled it at its multiple of 8, and the mask test would never pass
            // (round-4 advisor finding). A row cannot reach block k + 2's arrival before its neighbour has PASSED block k's wait (its block
            // k + 1 needs the neighbour's block-k + 1 rows), so the word of block k's parity stays at its multiple until everybody has seen it.
            unsigned long long* const e_k = e_up + (k & 1);
            if (wave == 0 && lane == 0) {
                (void)__hip_atomic_fetch_add(e_k, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                (void)__hip_atomic_fetch_add(e_k + 16, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            dma_prm(Bn);                                                 // (the next block's: nothing to wait for)
            if (wave == 0) coop_wait_mask2(e_k, ty > 0 ? 7ull : 3ull, e_k + 16, ty + 1 < ka.tiles_per_img ? 7ull : 3ull, ka.err);
            __builtin_amdgcn_s_barrier();
            dma_halo(ka.y2[k], true);
            wait_vm_older_than<0>();
            __builtin_amdgcn_s_barrier();
            B = Bn;
            SABER_TL(7);
        }
    }
    if (tid == 0 && ka.err) {      // the quarters of this tile - and, with an image per XCD, its image's first tile - ran on this XCD?
        const L2Reader xr(ka.xcc);
        const v4i xc = xr.load16((unsigned)t * 128u);
        bool same = xc.x == xc.y && xc.x == xc.z && xc.x == xc.w;
        if (ka.per_image && ka.nblk > 1) {      // (0: the image's first workgroup has not published yet - nothing to compare)
            const unsigned first = (unsigned)xr.load16((unsigned)(n * ka.tiles_per_img) * 128u + 16u).x;
            same = same && (!first || first == (unsigned)xc.x + 1u);
        }
        if (!same) __hip_atomic_fetch_add(ka.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    SABER_TL_FLUSH();
}

// ------------------------------------------------------------------------------------------------------------------------------
// The same persistent launch for the res3 stage (C = 128, 28 x 28): ONE workgroup per tile of 2 rows x 16 columns. At C = 128 a block's
// weights are 277 KB - what a QUARTER of a C = 256 block is - so a workgroup computes all channels of its tile itself and the fragment
// counts per wave are the same 18 + 8 + 8: no channel split, no hand-off inside a block (the 3x3 conv's tile and the y1 tile stay in
// LDS), only the edge barrier between two blocks. Waves: 3x3 = n-tile w (both tile rows per fragment), 1x1 + eltwise = 64 channels
// per wave (4 accumulators x 2 tile rows), last 1x1 = n-tile w. 28 tiles per image (14 tile rows x 2 column tiles) = 28 of an XCD's
// 32 CUs; an edge between two tile rows has 4 arriving workgroups (2 at the image's top and bottom).
template <int MAXB>
__global__ __launch_bounds__(512) void conv_stage1_c128_kernel(const Stage4KArgs<MAXB> ka) {
    constexpr int C1 = 128, K1 = 512, K2 = 128, NW = 8;
    constexpr int F0 = 18, F1 = 8, F2 = 8;                   // 1 KB weight fragments per wave and phase
    constexpr int CH1 = C1 / 16, PCH = CH1 + 1, HW = 18, HP = 4 * HW;
    constexpr int HCH = (HP * PCH + 63) / 64 * 64;
    constexpr int CPRW = K1 / 16;                            // 16-byte chunks per row of the shortcut / output tile: 32
    constexpr int P0C = (C1 / 4 * 3 + 63) / 64 * 64, P1C = K1 / 4 * 3, P2C = (K2 / 4 * 3 + 63) / 64 * 64;
    constexpr int MPC = C1 / 16 + 1;                         // LDS pitch (chunks) of the 3x3 output tile: 8 + 1 padding
    static_assert(P1C % 64 == 0 && (32 * CPRW) % (64 * NW) == 0, "DMA granularity");
    __shared__ v4i halo[HCH];
    __shared__ v4i tile[32 * CPRW];
    __shared__ v4i mid[32 * MPC];
    __shared__ v4i prm0[P0C];
    __shared__ v4i prm1[P1C];
    __shared__ v4i prm2[P2C];
    SABER_TL_DECL;
    SABER_TL(0);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fq = lane >> 4;
    const int b = blockIdx.x;
    const int slot = b >> 3;                                 // all workgroups of an image on one XCD: image (slot / tiles) * 8 + b % 8
    const int il = ka.mg_tpi ? (int)__umulhi((unsigned)slot, ka.mg_tpi) : slot;
    const int rem = slot - il * ka.tiles_per_img;
    const int n = il * 8 + (b & 7);
    if (n >= ka.N) return;
    const int t = n * ka.tiles_per_img + rem;
    const int ty = ka.mg_tiles_x ? (int)__umulhi((unsigned)rem, ka.mg_tiles_x) : rem;
    const int x0 = (rem - ty * ka.tiles_x) * 16, y0 = ty * 2;
    const int tile_rows = ka.tiles_per_img / ka.tiles_x;
    const int H = ka.H, W = ka.W;
    auto pix = [&](int m, bool& ok) -> int {                  // m = tile row * 16 + column
        int x = x0 + (m & 15), y = y0 + (m >> 4);
        ok = x < W && y < H;
        x = x < W ? x : W - 1;
        y = y < H ? y : H - 1;
        return (n * H + y) * W + x;
    };
    unsigned long long* const e_up = ka.img_ctr + ((size_t)n * (ka.tiles_per_img + 1) + ty) * 16;   // the edge above this tile row; + 16: below
    bool ok0, ok1;
    const int p0 = pix(frow, ok0), p1 = pix(16 + frow, ok1);

    auto dma_halo = [&](const void* x, bool l2) {            // 4 rows x 18 columns x 128 channels, zero page for the padding
        const char* xg = (const char*)x;
        for (int i = wave; i < HCH / 64; i += NW) {
            const int L = i * 64 + lane;
            const int hp = L / PCH, cc = L - hp * PCH;
            const int hy = hp / HW, hx = hp - hy * HW;
            const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
            const bool in = hp < HP && cc < CH1 && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
            const char* src = in ? xg + ((size_t)((n * H + gy) * W + gx) * C1 + cc * 16) : (const char*)ka.zero;
            if (l2 && in) lds_dma16_l2(src, halo + i * 64);
            else lds_dma16(src, halo + i * 64);
        }
    };
    auto dma_prm = [&](const StageBlk& B) {
        for (int i = wave; i < P0C / 64; i += NW) lds_dma16((const v4i*)B.prm0 + i * 64 + lane, prm0 + i * 64);
        for (int i = wave; i < P1C / 64; i += NW) lds_dma16((const v4i*)B.prm1 + i * 64 + lane, prm1 + i * 64);
        for (int i = wave; i < P2C / 64; i += NW) lds_dma16((const v4i*)B.prm2 + i * 64 + lane, prm2 + i * 64);
    };
    auto stream_of = [&](const StageBlk& B) -> const v4i* {   // this wave's fragments of block B, lane's 16 bytes
        return (const v4i*)B.wstream + (size_t)wave * ((F0 + F1 + F2) * 64) + lane;
    };

    // ---- entry: the first block's halo, shortcut tile and constants by DMA; its first 20 weight fragments ------------------------
    dma_halo(ka.x, false);
    for (int i = wave; i < 32 * CPRW / 64; i += NW) {
        const int L = i * 64 + lane;
        const int px = L / CPRW, c = (L % CPRW) ^ (px & 15);
        bool okp;
        const int pp_ = pix(px, okp);
        lds_dma16((const char*)ka.res + (size_t)pp_ * K1 + c * 16, tile + i * 64);
    }
    dma_prm(ka.blk[0]);
    asm volatile("" ::: "memory");
    v4i fr[F0 + 2];
    {
        const v4i* wsb = stream_of(ka.blk[0]);
#pragma unroll
        for (int r = 0; r < F0 + 2; ++r) {
            fr[r] = wsb[r * 64];
            asm volatile("" ::: "memory");                   // issue order = consumption order
        }
    }
    wait_vm_older_than<F0 + 2>();                             // everything older than the fragments: this wave's DMA
    __builtin_amdgcn_s_barrier();
    SABER_TL(1);
    unsigned my_xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(my_xcc));
    my_xcc = (my_xcc & 7u) + 1u;
    if (rem == 0 && tid == 0) ka.xcc[(size_t)t * 32 + 4] = my_xcc;      // the image's first tile publishes its XCD (+ 1: 0 = not yet)

    StageBlk B = ka.blk[0];
    for (int k = 0; k < ka.nblk; ++k) {
        const v4i z = {0, 0, 0, 0};
        v4i fs[F1 - 2 + F2];                                 // the rest of this block's stream: lands under the 3x3 conv
        {
            const v4i* wsb = stream_of(B);
#pragma unroll
            for (int r = 0; r < F1 - 2 + F2; ++r) {
                fs[r] = wsb[(F0 + 2 + r) * 64];
                asm volatile("" ::: "memory");
            }
        }
        // ================= phase 0: 3x3 conv, 16 mid channels per wave, both tile rows per fragment ==================================
        {
            const int xm0 = B.in0_u8 ? (int)0x80808080u : 0;
            const int c0 = wave * 16 + fq * 4;
            const v4i* pp = prm0 + (c0 / 4) * 3;
            v4i acc0 = pp[2], acc1 = acc0;                   // starts at the compensation (exact integer sum)
            const v4i* hb = halo + frow * PCH + fq;
#pragma unroll
            for (int s = 0; s < F0; ++s) {
                const int kl = s % 2, tap = s / 2;
                const int dy = tap / 3, dx = tap % 3;
                v4i b0 = hb[(dy * HW + dx) * PCH + kl * 4];
                v4i b1 = hb[((dy + 1) * HW + dx) * PCH + kl * 4];
                b0.x ^= xm0; b0.y ^= xm0; b0.z ^= xm0; b0.w ^= xm0;
                b1.x ^= xm0; b1.y ^= xm0; b1.z ^= xm0; b1.w ^= xm0;
                acc0 = mma_step(fr[s], b0, acc0);
                acc1 = mma_step(fr[s], b1, acc1);
            }
            const float lo0 = B.relu0 ? 0.f : -3.0e38f;
            const float off0 = B.in_u8 ? 0.f : 128.f;
            const unsigned xo0 = B.in_u8 ? 0u : 0x80808080u;
            const unsigned o0 = chain_out_pack(acc0, z, __builtin_bit_cast(v4f, pp[1]), __builtin_bit_cast(v4f, pp[0]), lo0, off0, xo0);
            const unsigned o1 = chain_out_pack(acc1, z, __builtin_bit_cast(v4f, pp[1]), __builtin_bit_cast(v4f, pp[0]), lo0, off0, xo0);
            *(unsigned*)((char*)mid + frow * (MPC * 16) + c0) = o0;
            *(unsigned*)((char*)mid + (16 + frow) * (MPC * 16) + c0) = o1;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                        // the 3x3 conv's tile is complete
        SABER_TL(2);

        // ================= phase 1: 1x1 conv + eltwise, 64 channels per wave ==========================================================
        {
            const int xmask = B.in_u8 ? (int)0x80808080u : 0;
            const int cg = wave * 64 + fq * 16;              // 16 consecutive channels of pixel (row, frow)
            const v4i* pp = prm1 + (cg / 4) * 3;
            v4i acc[2][4];
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int mf = 0; mf < 4; ++mf) acc[m][mf] = pp[mf * 3 + 2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    v4i bo = mid[(m * 16 + frow) * MPC + ks * 4 + fq];
                    bo.x ^= xmask; bo.y ^= xmask; bo.z ^= xmask; bo.w ^= xmask;
#pragma unroll
                    for (int mf = 0; mf < 4; ++mf) {
                        const int f = ks * 4 + mf;           // fragments 0, 1 came with the previous block's prefetch
                        acc[m][mf] = mma_step(f < 2 ? fr[F0 + f] : fs[f - 2], bo, acc[m][mf]);
                    }
                }
            const float lo_s8 = B.relu1 ? 0.f : -128.f;
            const float res_lo = B.res_relu ? 0.f : -3.0e38f;
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                v4i* tp = tile + (m * 16 + frow) * CPRW + ((cg / 16) ^ frow);
                const v4i rs = *tp;
                v4i o;
                o.x = (int)chain_elt_pack(acc[m][0], z, __builtin_bit_cast(v4f, pp[1]), __builtin_bit_cast(v4f, pp[0]), (unsigned)rs.x, lo_s8, res_lo, B);
                o.y = (int)chain_elt_pack(acc[m][1], z, __builtin_bit_cast(v4f, pp[4]), __builtin_bit_cast(v4f, pp[3]), (unsigned)rs.y, lo_s8, res_lo, B);
                o.z = (int)chain_elt_pack(acc[m][2], z, __builtin_bit_cast(v4f, pp[7]), __builtin_bit_cast(v4f, pp[6]), (unsigned)rs.z, lo_s8, res_lo, B);
                o.w = (int)chain_elt_pack(acc[m][3], z, __builtin_bit_cast(v4f, pp[10]), __builtin_bit_cast(v4f, pp[9]), (unsigned)rs.w, lo_s8, res_lo, B);
                *tp = o;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int j = 0; j < 32 * CPRW / 512; ++j) {          // the 32 x 512 tile -> y1, coalesced
            const int L = tid + j * 512;
            const int px = L / CPRW, c = (L % CPRW) ^ (px & 15);
            bool okp;
            const int pp_ = pix(px, okp);
            if (okp) *(v4i*)((char*)ka.y1[k] + (size_t)pp_ * K1 + c * 16) = tile[L];
        }
        SABER_TL(4);
        const StageBlk Bn = ka.blk[k + 1 < ka.nblk ? k + 1 : k];      // the next block's constants, one phase ahead of their use
        if (k + 1 < ka.nblk) {         // the next block's first 20 fragments: one block ahead
            const v4i* wsb = stream_of(Bn);
#pragma unroll
            for (int r = 0; r < F0 + 2; ++r) {
                fr[r] = wsb[r * 64];
                asm volatile("" ::: "memory");
            }
        }

        // ================= phase 2: second 1x1 conv, 16 channels per wave, its operand is the y1 tile in LDS ===========================
        {
            const int c2 = wave * 16 + fq * 4;
            const v4i* pp = prm2 + (c2 / 4) * 3;
            v4i acc0 = pp[2], acc1 = acc0;
#pragma unroll
            for (int ks = 0; ks < F2; ++ks) {
                const int ch = (ks * 4 + fq) ^ frow;
                acc0 = mma_step(fs[F1 - 2 + ks], tile[frow * CPRW + ch], acc0);
                acc1 = mma_step(fs[F1 - 2 + ks], tile[(16 + frow) * CPRW + ch], acc1);
            }
            const float lo2 = B.relu2 ? 0.f : -3.0e38f;
            const float off2 = B.out_u8_2 ? 0.f : 128.f;
            const unsigned xm2 = B.out_u8_2 ? 0u : 0x80808080u;
            const unsigned o0 = chain_out_pack(acc0, z, __builtin_bit_cast(v4f, pp[1]), __builtin_bit_cast(v4f, pp[0]), lo2, off2, xm2);
            const unsigned o1 = chain_out_pack(acc1, z, __builtin_bit_cast(v4f, pp[1]), __builtin_bit_cast(v4f, pp[0]), lo2, off2, xm2);
            if (ok0) *(unsigned*)((char*)ka.y2[k] + (size_t)p0 * K2 + c2) = o0;
            if (ok1) *(unsigned*)((char*)ka.y2[k] + (size_t)p1 * K2 + c2) = o1;
        }
        SABER_TL(6);
        if (k + 1 < ka.nblk) {
            // ============= between two blocks: the edge barriers of conv_stage4_c256_kernel, tiles_x workgroups per tile row ================
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                                // (every wave is also done with this block's constants)
            unsigned long long* const e_k = e_up + (k & 1);              // (one word per block parity: see conv_stage4_c256_kernel)
            if (wave == 0 && lane == 0) {
                (void)__hip_atomic_fetch_add(e_k, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                (void)__hip_atomic_fetch_add(e_k + 16, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            dma_prm(Bn);
            if (wave == 0) {
                const unsigned long long one = (unsigned long long)ka.tiles_x;     // arrivals at an edge: tiles_x per tile row
                coop_wait_mask2(e_k, (ty > 0 ? 2 * one : one) - 1, e_k + 16, (ty + 1 < tile_rows ? 2 * one : one) - 1, ka.err);
            }
            __builtin_amdgcn_s_barrier();
            dma_halo(ka.y2[k], true);
            wait_vm_older_than<0>();
            __builtin_amdgcn_s_barrier();
            B = Bn;
            SABER_TL(7);
        }
    }
    if (tid == 0 && ka.err) {      // this tile ran on its image's XCD? (0: the first tile has not published yet - nothing to compare)
        const unsigned first = (unsigned)L2Reader(ka.xcc).load16((unsigned)(n * ka.tiles_per_img) * 128u + 16u).x;
        if (first && first != my_xcc) __hip_atomic_fetch_add(ka.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    SABER_TL_FLUSH();
}

template <int MAXB>
static hipError_t launch_stage1(const Stage4KArgs<MAXB>& ka, hipStream_t s) {
    if (ka.nblk <= 1 || ka.nblk > MAXB || ka.N <= 0 || !ka.blk || !ka.xcc || !ka.per_image || !ka.img_ctr ||
        (ka.tiles_x != 1 && ka.tiles_x != 2 && ka.tiles_x != 4))
        return hipErrorInvalidValue;
    const dim3 grid((ka.N + 7) / 8 * ka.tiles_per_img * 8), block(512);
    hipLaunchKernelGGL(conv_stage1_c128_kernel<MAXB>, grid, block, 0, s, ka);
    return hipGetLastError();
}
hipError_t launch_conv_stage1_c128(const Stage4KArgs<STAGE4_SHORT>& a, hipStream_t s) { return launch_stage1(a, s); }
hipError_t launch_conv_stage1_c128(const Stage4KArgs<STAGE4_LONG>& a, hipStream_t s) { return launch_stage1(a, s); }
template <int MAXB>
static hipError_t launch_stage4(const Stage4KArgs<MAXB>& ka, hipStream_t s) {
    if (ka.nblk <= 0 || ka.nblk > MAXB || ka.N <= 0 || !ka.blk || !ka.grp_ctr || !ka.xch || !ka.xcc ||
        (ka.nblk > 1 && (!ka.per_image || !ka.img_ctr || ka.tiles_x != 1)))
        return hipErrorInvalidValue;
    const int tiles = ka.N * ka.tiles_per_img;
    const dim3 grid(ka.per_image ? (ka.N + 7) / 8 * ka.tiles_per_img * 4 * 8 : (tiles + 7) / 8 * 32), block(512);
    hipLaunchKernelGGL(conv_stage4_c256_kernel<MAXB>, grid, block, 0, s, ka);
    return hipGetLastError();
}
hipError_t launch_conv_stage4(const Stage4KArgs<STAGE4_SHORT>& a, hipStream_t s) { return launch_stage4(a, s); }
hipError_t launch_conv_stage4(const Stage4KArgs<STAGE4_LONG>& a, hipStream_t s) { return launch_stage4(a, s); }

}  // namespace saber_mi355x
