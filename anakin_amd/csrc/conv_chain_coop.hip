// anakin_amd/csrc/conv_chain_coop.hip - the 3x3-led chain of a ResNet res4 block (conv 3x3 C -> C, conv 1x1 C -> 4C + SaberEltwise sum +
// relu, next block's conv 1x1 4C -> C; C = 256) with the WEIGHT STREAM SPLIT over two cooperating workgroups per pixel tile.
//
// Why (round-3 verdict, item 3): conv1x1_chain_kernel gives every pixel tile to ONE workgroup, which therefore
// streams the block's whole 1.1 MB of weights through one CU's vector-memory path - measured 122 GB/s per CU = 51 of the path's
// 64 B/clk with eight waves, so the launch's 11.5 us IS its weight stream, while 112 of 256 CUs have a workgroup at all. The
// only lever left is fewer weight bytes per CU: two workgroups share a tile, each computes HALF of every convolution's output
// channels (integer sums: the bits cannot change), 557 KB per CU, 224 CUs busy.
//   phase 0   3x3: mid channels [h * 128, + 128)                         -> xch[tile][pixel][256] (global, through the L2) + LDS
//   arrive 1 ... phase 1's k-steps over THIS half's mid channels (LDS) ... wait 1 ... the partner's half (sc1 loads from the L2)
//   phase 1   1x1 + eltwise: channels [h * 512, + 512), residual half     -> y1 (the operator's own output tensor) + LDS
//   arrive 2 ... phase 2's k-steps over this half's y1 channels (LDS) ... wait 2 ... the partner's half (sc1 loads)
//   phase 2   1x1: channels [h * 128, + 128)                              -> y2
// (each half's weight stream is packed with its own k-steps first; integer sums: the order cannot change a bit)
// The two workgroups of a tile are 8 apart in the grid = the same XCD (the placement api_conv.hip: xcd_round_robin verifies once
// per device; each half also publishes its XCC_ID and the pair compares them), so the hand-off needs no L2 write-back: plain
// stores, s_waitcnt vmcnt(0), one relaxed WORKGROUP-scope atomic on the pair's counter (it executes in this XCD's L2; an agent-scope
// one is performed beyond it: 1.6 us per arrival), a spin on SCALAR loads (they do not queue behind the wave's weight loads), then
// sc1 LOADS of the partner's data (they miss the L1 and hit the L2) - NOT `buffer_inv sc1` + plain loads as in round 3: that
// device-scope invalidate also drops the XCD's L2 (coop_sync.h). The barrier itself is the protocol profiles/r03/boundary_probe.txt
// measured at 0.9 us for 32 workgroups. Every tile's counters (and XCC words) sit in 128-byte lines of their own - tiles t .. t + 7 of a group run on
// eight different XCDs, whose L2s are not coherent with each other: with the counters packed, eight L2s fought over one line and the
// launch took 37 us instead of 12. Counters are never reset: an arrival adds 1, the first of a pair waits for the value to become even again.
// Both workgroups of a pair are dispatched back to back on one XCD, so a waiting workgroup's partner always gets a slot; a spin
// that still exceeds ~20 ms (a foreign kernel holding every CU) gives up, counts itself in the host-visible error word and lets
// the launch finish with garbage - the host turns that into an error status on the next run and stops selecting this form.
// Everything else (weights straight into MFMA A registers from a per-wave stream in consumption order, LDS-DMA halo with a
// padded pixel pitch, XOR-swizzled residual tile, the requantising epilogues of epilogue_pack.h) is conv1x1_chain.hip's.
#include "coop_sync.h"

namespace saber_mi355x {

// C1 = 256, K1 = 1024, K2 = 256, 16-pixel tiles (one row x 16 columns of one image), 8 waves, 2 workgroups per tile.
__global__ __launch_bounds__(512) void conv_chain_coop_c256_kernel(const CoopKArgs ka) {
    const ChainKArgs& a = ka.c;
    constexpr int C1 = 256, K1 = 1024, K2 = 256, NW = 8, R = 16;
    constexpr int KS1 = C1 / 64, KS2 = K1 / 64;              // 64-byte k-steps of the two 1x1 convs' reductions
    constexpr int K1W = K1 / 2, K0W = C1 / 2, K2W = K2 / 2;  // this workgroup's output channels per phase
    constexpr int T0 = 9 * KS1, T1 = KS1 * 4, T2 = KS2;      // steps (1 KB of weights each) per wave and phase: 36 + 16 + 16
    constexpr int CH1 = C1 / 16, PCH = CH1 + 1, HW = 18, HP = 3 * HW;
    constexpr int HCH = (HP * PCH + 63) / 64 * 64;
    constexpr int CPRW = K1W / 16;                           // 16-byte chunks per row of the residual / output half tile: 32
    constexpr int P0C = (K0W / 4 * 3 + 63) / 64 * 64, P1C = K1W / 4 * 3, P2C = (K2W / 4 * 3 + 63) / 64 * 64;
    static_assert(P1C % 64 == 0 && (16 * CPRW) % 64 == 0, "DMA granularity");
    static_assert(T2 == R, "the second 1x1 conv's steps are all in the ring when it starts");

    constexpr int MPC = K0W / 16 + 1;                        // LDS pitch (chunks) of this half's 3x3 output tile: 8 + 1 padding
    __shared__ v4i halo[HCH];
    __shared__ v4i tile[16 * CPRW];
    __shared__ v4i mid_own[16 * MPC];
    __shared__ v4i prm0[P0C];
    __shared__ v4i prm1[P1C];
    __shared__ v4i prm2[P2C];
    SABER_TL_DECL;
    SABER_TL(0);
    asm volatile("" ::"s"(a.x), "s"(a.wstream), "s"(a.res), "s"(a.prm0), "s"(a.prm1), "s"(a.prm2), "s"(a.zero), "s"(a.H), "s"(a.W),
                 "s"(a.tiles_x), "s"(a.tiles_per_img), "s"(a.mg_tiles_x), "s"(a.mg_tpi), "s"(ka.n_tiles));
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fq = lane >> 4;
    // workgroups b and b + 8 (the same XCD) are the two halves of tile (b / 16) * 8 + b % 8
    const int b = blockIdx.x;
    const int half = (b >> 3) & 1;
    const int t = (b >> 4) * 8 + (b & 7);
    if (t >= ka.n_tiles) return;                              // (both halves of a pair take this exit together)
    const int n = a.mg_tpi ? (int)__umulhi((unsigned)t, a.mg_tpi) : t;
    const int rem = t - n * a.tiles_per_img;
    const int y0 = a.mg_tiles_x ? (int)__umulhi((unsigned)rem, a.mg_tiles_x) : rem;
    const int x0 = (rem - y0 * a.tiles_x) * 16;
    auto pix = [&](int px, bool& ok) -> int {
        int x = x0 + px;
        ok = x < a.W;
        x = x < a.W ? x : a.W - 1;
        return (n * a.H + y0) * a.W + x;
    };

    // ---- LDS by DMA: the 3x3 conv's input halo (zero page for the padding), this half's residual rows and constants ------------
    {
        const char* xg = (const char*)a.x;
        for (int i = wave; i < HCH / 64; i += NW) {
            const int L = i * 64 + lane;
            const int hp = L / PCH, cc = L - hp * PCH;
            const int hy = hp / HW, hx = hp - hy * HW;
            const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
            const bool in = hp < HP && cc < CH1 && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
            const char* src = in ? xg + ((size_t)((n * a.H + gy) * a.W + gx) * C1 + cc * 16) : (const char*)a.zero;
            lds_dma16(src, halo + i * 64);
        }
        const char* rg = (const char*)a.res + half * K1W;
        {
            const int L = wave * 64 + lane;                  // 16 x 32 chunks = 8 instructions, one per wave
            const int px = L / CPRW, c = (L % CPRW) ^ (px & 15);
            bool ok;
            const int p = pix(px, ok);
            lds_dma16(rg + (size_t)p * K1 + c * 16, tile + wave * 64);
        }
        if (wave < P0C / 64) lds_dma16((const v4i*)a.prm0 + half * (K0W / 4 * 3) + wave * 64 + lane, prm0 + wave * 64);
        for (int i = wave; i < P1C / 64; i += NW) lds_dma16((const v4i*)a.prm1 + half * P1C + i * 64 + lane, prm1 + i * 64);
        if (wave < P2C / 64) lds_dma16((const v4i*)a.prm2 + half * (K2W / 4 * 3) + wave * 64 + lane, prm2 + wave * 64);
    }
    asm volatile("" ::: "memory");
    // ---- weight ring: this (half, wave)'s stream, [3x3: tap][k-step] | [1x1: k-step][accumulator] | [1x1: k-step] -------------
    const v4i* wsb = (const v4i*)a.wstream + (size_t)(half * NW + wave) * ((T0 + T1 + T2) * 64);
    v4i ring[R];
#pragma unroll
    for (int r = 0; r < R; ++r) ring[r] = wsb[r * 64 + lane];
    wsb += R * 64;
    wait_vm_older_than<R>();                                 // everything older than the ring: this wave's DMA
    __builtin_amdgcn_s_barrier();
    SABER_TL(1);

    // ================= phase 0: 3x3 conv, 16 mid channels per wave ==============================================================
    {
        const int xm0 = a.in0_u8 ? (int)0x80808080u : 0;
        const int c0 = wave * 16 + fq * 4;                   // within this half
        const v4i* pp = prm0 + (c0 / 4) * 3;
        v4i acc = pp[2];                                     // starts at the compensation (exact integer sum)
        const v4i* hb = halo + frow * PCH + fq;
#pragma unroll
        for (int s = 0; s < T0; ++s) {
            const int ks = s % KS1, tap = s / KS1;
            const int dy = tap / 3, dx = tap % 3;
            v4i bv = hb[(dy * HW + dx) * PCH + ks * 4];
            bv.x ^= xm0; bv.y ^= xm0; bv.z ^= xm0; bv.w ^= xm0;
            acc = mma_step(ring[s % R], bv, acc);
            ring[s % R] = wsb[s * 64 + lane];
        }
        wsb += T0 * 64;
        const float lo0 = a.relu0 ? 0.f : -3.0e38f;
        const float off0 = a.in_u8 ? 0.f : 128.f;
        const unsigned xo0 = a.in_u8 ? 0u : 0x80808080u;
        const unsigned o = chain_out_pack(acc, v4i{0, 0, 0, 0}, __builtin_bit_cast(v4f, pp[1]), __builtin_bit_cast(v4f, pp[0]), lo0, off0, xo0);
        *(unsigned*)((char*)ka.coop_xch + ((size_t)t * 16 + frow) * C1 + half * K0W + c0) = o;     // for the partner (through the L2)
        *(unsigned*)((char*)mid_own + frow * (MPC * 16) + c0) = o;                                   // for this workgroup (LDS)
    }
    if (tid == 0) {      // (stored HERE, not at entry: a store pending beside the ring's loads makes the compiler wait for vmcnt(0))
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        ka.coop_xcc[t * 32 + half] = xcc & 7u;
    }
    coop_arrive(ka.coop_ctr + t * 32);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                            // mid_own is complete
    SABER_TL(2);

    // ================= phase 1: 1x1 conv + eltwise, 64 channels per wave; k-steps of this half's mid channels first ===============
    {
        const int xmask = a.in_u8 ? (int)0x80808080u : 0;
        const int cg = wave * 64 + fq * 16;                  // within this half: 16 consecutive channels of pixel frow
        const v4i* pp = prm1 + (cg / 4) * 3;
        v4i acc[4];
#pragma unroll
        for (int mf = 0; mf < 4; ++mf) acc[mf] = pp[mf * 3 + 2];
        v4i bo[KS1 / 2];
#pragma unroll
        for (int k = 0; k < KS1 / 2; ++k) {
            bo[k] = mid_own[frow * MPC + k * 4 + fq];
            bo[k].x ^= xmask; bo[k].y ^= xmask; bo[k].z ^= xmask; bo[k].w ^= xmask;
        }
#pragma unroll
        for (int s = 0; s < T1 / 2; ++s) {                   // the stream holds this half's k-steps first (api_chain.hip: ks0)
            const int ri = (T0 + s) % R, k = s / 4, mf = s % 4;
            acc[mf] = mma_step(ring[ri], bo[k], acc[mf]);
            ring[ri] = wsb[s * 64 + lane];
        }
        coop_wait(ka.coop_ctr + t * 32, ka.coop_err);        // the partner's half of the 3x3 tile is in the L2
        SABER_TL(3);
        const L2Reader xch_l2(ka.coop_xch);
        const unsigned mo = (unsigned)(((size_t)t * 16 + frow) * C1 + (1 - half) * K0W + fq * 16);
        v4i bp[KS1 / 2];
#pragma unroll
        for (int k = 0; k < KS1 / 2; ++k) {
            bp[k] = xch_l2.load16<SABER_COOP_AUX>(mo + k * 64);
            bp[k].x ^= xmask; bp[k].y ^= xmask; bp[k].z ^= xmask; bp[k].w ^= xmask;
        }
#pragma unroll
        for (int s = T1 / 2; s < T1; ++s) {
            const int ri = (T0 + s) % R, k = s / 4 - KS1 / 2, mf = s % 4;
            acc[mf] = mma_step(ring[ri], bp[k], acc[mf]);
            ring[ri] = wsb[s * 64 + lane];
        }
        wsb += T1 * 64;
        const float lo_s8 = a.relu1 ? 0.f : -128.f;
        const float res_lo = a.res_relu ? 0.f : -3.0e38f;
        v4i* tp = tile + frow * CPRW + ((cg / 16) ^ frow);
        const v4i rs = *tp;
        v4i o;
        o.x = (int)chain_elt_pack(acc[0], v4i{0, 0, 0, 0}, __builtin_bit_cast(v4f, pp[1]), __builtin_bit_cast(v4f, pp[0]), (unsigned)rs.x, lo_s8, res_lo, a);
        o.y = (int)chain_elt_pack(acc[1], v4i{0, 0, 0, 0}, __builtin_bit_cast(v4f, pp[4]), __builtin_bit_cast(v4f, pp[3]), (unsigned)rs.y, lo_s8, res_lo, a);
        o.z = (int)chain_elt_pack(acc[2], v4i{0, 0, 0, 0}, __builtin_bit_cast(v4f, pp[7]), __builtin_bit_cast(v4f, pp[6]), (unsigned)rs.z, lo_s8, res_lo, a);
        o.w = (int)chain_elt_pack(acc[3], v4i{0, 0, 0, 0}, __builtin_bit_cast(v4f, pp[10]), __builtin_bit_cast(v4f, pp[9]), (unsigned)rs.w, lo_s8, res_lo, a);
        *tp = o;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    {   // this half's 16 x 512 tile -> y1, coalesced
        const int L = tid;
        const int px = L / CPRW, c = (L % CPRW) ^ (px & 15);
        bool ok;
        const int p = pix(px, ok);
        if (ok) *(v4i*)((char*)a.y1 + (size_t)p * K1 + half * K1W + c * 16) = tile[L];
    }
    coop_arrive(ka.coop_ctr + t * 32 + 16);
    SABER_TL(4);

    // ================= phase 2: second 1x1 conv, 16 channels per wave; k-steps of this half's y1 channels (still in LDS) first =====
    {
        bool ok;
        const int p = pix(frow, ok);
        const int c2 = wave * 16 + fq * 4;                   // within this half
        const v4i* pp = prm2 + (c2 / 4) * 3;
        v4i acc = pp[2];
        v4i bo[KS2 / 2];
#pragma unroll
        for (int k = 0; k < KS2 / 2; ++k) bo[k] = tile[frow * CPRW + ((k * 4 + fq) ^ frow)];
#pragma unroll
        for (int s = 0; s < T2 / 2; ++s) {
            const int ri = (T0 + T1 + s) % R;
            acc = mma_step(ring[ri], bo[s], acc);            // (T2 == R: the ring already holds the whole phase, no refills)
        }
        coop_wait(ka.coop_ctr + t * 32 + 16, ka.coop_err);   // the partner's half of the y1 tile is in the L2
        SABER_TL(5);
        const L2Reader y1_l2(a.y1);
        const unsigned yo = (unsigned)((size_t)p * K1 + (1 - half) * K1W + fq * 16);
        v4i bp[KS2 / 2];
#pragma unroll
        for (int k = 0; k < KS2 / 2; ++k) bp[k] = y1_l2.load16<SABER_COOP_AUX>(yo + k * 64);
#pragma unroll
        for (int s = T2 / 2; s < T2; ++s) {
            const int ri = (T0 + T1 + s) % R;
            acc = mma_step(ring[ri], bp[s - T2 / 2], acc);
        }
        const float lo2 = a.relu2 ? 0.f : -3.0e38f;
        const float off2 = a.out_u8_2 ? 0.f : 128.f;
        const unsigned xm2 = a.out_u8_2 ? 0u : 0x80808080u;
        const unsigned o = chain_out_pack(acc, v4i{0, 0, 0, 0}, __builtin_bit_cast(v4f, pp[1]), __builtin_bit_cast(v4f, pp[0]), lo2, off2, xm2);
        if (ok) *(unsigned*)((char*)a.y2 + (size_t)p * K2 + half * K2W + c2) = o;
    }
    if (tid == 0 && ka.coop_err) {      // both halves ran on one XCD? (off every wave's critical path: after the last store)
        const v4i xc = L2Reader(ka.coop_xcc).load16((unsigned)t * 128u);
        if (xc.x != xc.y) __hip_atomic_fetch_add(ka.coop_err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    SABER_TL(6);
    SABER_TL_FLUSH();
}

hipError_t launch_conv_chain_coop(const CoopKArgs& ka, hipStream_t s) {
    if (ka.n_tiles <= 0 || !ka.coop_ctr || !ka.coop_xch || !ka.coop_xcc) return hipErrorInvalidValue;
    const dim3 grid((ka.n_tiles + 7) / 8 * 16), block(512);
    hipLaunchKernelGGL(conv_chain_coop_c256_kernel, grid, block, 0, s, ka);
    return hipGetLastError();
}

}  // namespace saber_mi355x
