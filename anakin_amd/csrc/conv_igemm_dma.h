// anakin_amd/csrc/conv_igemm_dma.h — implicit-GEMM convolution with LDS-DMA staging (gfx950).
//
// Same GEMM view, fragment layout, LDS swizzle and epilogues as conv_igemm_impl.h, but both operands
// are staged with `global_load_lds_dwordx4` (16 B per lane written straight into LDS, no VGPR round
// trip) into a RING of NS stages, so NS-1 stages of global loads are in flight while one stage is
// consumed. The deep-K layers of ResNet (stage 4/5: few output tiles, 4..18 stages of reduction) are
// latency-bound with one stage in flight; with the ring the K loop costs ~one memory latency.
//
//  * LDS-DMA writes lane L's 16 bytes at  wave-uniform base + L*16, so the LDS image is lane-linear:
//    the swizzle is applied on the SOURCE side (each lane fetches the logical chunk whose physical slot
//    it owns) — cdna_hip_programming.md §5.4 rule 21.
//  * Out-of-image taps / rows beyond the tile cannot be predicated off (an inactive lane would leave
//    stale LDS bytes): they fetch from a 16-byte zero page instead.
//  * u8 activations: the XOR 0x80 (u8 -> s8 shift) is applied to the pixel fragments after ds_read.
//  * Ordering: counted `s_waitcnt vmcnt(N)` (this wave's DMA for the stage has landed) -> raw
//    `s_barrier` (every wave's has) -> ds_read. Never __syncthreads() inside the loop (it would drain
//    the DMA queue), never an ordinary global load inside the loop (hipcc would wait vmcnt(0) for it).
#pragma once
#include "conv_igemm_impl.h"

namespace saber_mi355x {

template <int BYTES>
constexpr int ring_stages() {   // ring depth from a ~96 KiB LDS budget, 2..8
    constexpr int n = (96 * 1024) / BYTES;
    return n > 8 ? 8 : (n < 2 ? 2 : n);
}

#define SABER_WAIT_VMCNT_CASE(n) \
    case n: asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory"); break;

// waits until at most `cnt` LDS-DMA instructions of this wave are outstanding (cnt is wave-uniform)
__device__ __forceinline__ void wait_vmcnt(int cnt) {
    switch (cnt) {
        SABER_WAIT_VMCNT_CASE(0) SABER_WAIT_VMCNT_CASE(1) SABER_WAIT_VMCNT_CASE(2) SABER_WAIT_VMCNT_CASE(3)
        SABER_WAIT_VMCNT_CASE(4) SABER_WAIT_VMCNT_CASE(5) SABER_WAIT_VMCNT_CASE(6) SABER_WAIT_VMCNT_CASE(7)
        SABER_WAIT_VMCNT_CASE(8) SABER_WAIT_VMCNT_CASE(9) SABER_WAIT_VMCNT_CASE(10) SABER_WAIT_VMCNT_CASE(11)
        SABER_WAIT_VMCNT_CASE(12) SABER_WAIT_VMCNT_CASE(13) SABER_WAIT_VMCNT_CASE(14) SABER_WAIT_VMCNT_CASE(15)
        SABER_WAIT_VMCNT_CASE(16) SABER_WAIT_VMCNT_CASE(17) SABER_WAIT_VMCNT_CASE(18) SABER_WAIT_VMCNT_CASE(19)
        SABER_WAIT_VMCNT_CASE(20) SABER_WAIT_VMCNT_CASE(21) SABER_WAIT_VMCNT_CASE(22) SABER_WAIT_VMCNT_CASE(23)
        SABER_WAIT_VMCNT_CASE(24) SABER_WAIT_VMCNT_CASE(25) SABER_WAIT_VMCNT_CASE(26) SABER_WAIT_VMCNT_CASE(27)
        SABER_WAIT_VMCNT_CASE(28) SABER_WAIT_VMCNT_CASE(29) SABER_WAIT_VMCNT_CASE(30) SABER_WAIT_VMCNT_CASE(31)
        SABER_WAIT_VMCNT_CASE(32) SABER_WAIT_VMCNT_CASE(36) SABER_WAIT_VMCNT_CASE(40) SABER_WAIT_VMCNT_CASE(42)
        SABER_WAIT_VMCNT_CASE(48) SABER_WAIT_VMCNT_CASE(56)
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

// MODE 0: int8 (C % 16 == 0); MODE 2: f32 (C % 4 == 0).
// WG wave groups of 4 waves (256 threads) each: every group multiplies the same output tile against its
// own KS k-steps of a stage (intra-block split-K) so that each SIMD hosts WG waves whose wait / issue
// phases overlap; the partial accumulators are summed through LDS before the epilogue.
template <int MODE, int TM, int TN, int KS, int EK, int WG>
__global__ __launch_bounds__(256 * WG) void conv_igemm_dma_kernel(const ConvKArgs a) {
    constexpr bool F32 = (MODE == 2);
    constexpr int ES = F32 ? 4 : 1;
    constexpr int EC = 16 / ES;
    constexpr int NT = 256 * WG;         // threads per block
    constexpr int CPR = 4 * KS * WG;     // 16-byte chunks per row per stage
    constexpr int ESTAGE = CPR * EC;
    constexpr int RPP = NT / CPR;        // rows covered by one DMA pass of the block
    constexpr int BMK = 2 * TM * 16;
    constexpr int BNP = 2 * TN * 16;
    constexpr int WIT = (BMK + RPP - 1) / RPP;
    constexpr int XIT = (BNP + RPP - 1) / RPP;
    constexpr int WROWS = WIT * RPP;     // tile rows rounded up to whole DMA passes (pad rows fetch zeros)
    constexpr int XROWS = XIT * RPP;
    constexpr int STAGE = (WROWS + XROWS) * CPR;        // 16-byte chunks per stage
    constexpr int NS = ring_stages<STAGE * 16>();
    constexpr int LPS = WIT + XIT;       // DMA instructions per wave per stage
    constexpr int NV = TM * 4;
    using acc_t = typename std::conditional<F32, v4f, v4i>::type;

    __shared__ v4i lds[NS][STAGE];
    SABER_TL_DECL;
    SABER_TL(0);
    pin_hot_args(a);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;           // 0 .. 4*WG-1
    const int grp = wave >> 2;           // wave group = which k-steps of a stage
    const int wm = (wave >> 1) & 1, wn = wave & 1;
    int tile_px, tile_ky;
    xcd_tile(a, tile_px, tile_ky);
    const int pix_base = tile_px * BNP;
    const int k_base = tile_ky * BMK;
    const int pc = tid % CPR;            // physical chunk column this thread's DMA lands in
    const int lr = tid / CPR;            // first LDS row it fills
    // logical chunk column whose physical home (in row lr + it*RPP, any it) is pc: inverse of phys_chunk
    int lq;
    if constexpr (CPR == 4) lq = (0x78 >> (2 * (pc ^ ((lr >> 2) & 3)))) & 3;
    else if constexpr (CPR == 8) lq = pc ^ ((lr >> 1) & 7);
    else lq = pc ^ (lr & 15);            // CPR >= 16 (phys_chunk XORs the low 4 bits of the chunk index)

    // ---- per-row gather state ---------------------------------------------------------------------
    int x_base[XIT], x_ih0[XIT], x_iw0[XIT];
    bool x_ok[XIT];
    const int ohw = a.OH * a.OW;
#pragma unroll
    for (int it = 0; it < XIT; ++it) {
        const int r = lr + it * RPP;
        const int p = pix_base + r;
        x_ok[it] = (r < BNP) && (p < a.M);
        const int pp = x_ok[it] ? p : 0;
        int n, oh, ow;
        pixel_decode(a, pp, n, oh, ow);
        x_base[it] = n * a.H * a.W * a.C;
        x_ih0[it] = oh * a.stride_h - a.pad_h;
        x_iw0[it] = ow * a.stride_w - a.pad_w;
    }
    const char* w_src[WIT];              // weight row base pointers (nullptr -> zero page)
    const int w_row_bytes = a.Kg_pad * ES;
#pragma unroll
    for (int it = 0; it < WIT; ++it) {
        const int rho = lr + it * RPP;   // LDS row -> weight row of the block tile (inverse of the store permutation)
        const int tile16 = rho >> 4;
        const int r = (tile16 / TM) * (TM * 16) + ((rho >> 2) & 3) * (TM * 4) + (tile16 % TM) * 4 + (rho & 3);
        w_src[it] = (rho < BMK) ? (const char*)a.w + (size_t)(k_base + r) * w_row_bytes + (size_t)lq * 16 : nullptr;
    }
    int cur_c, cur_i, cur_j;
    {
        const int kk0 = lq * EC;
        const int tap = kk0 / a.C;
        cur_c = kk0 - tap * a.C;
        cur_i = tap / a.kw;
        cur_j = tap - cur_i * a.kw;
    }
    const char* zero = (const char*)a.zero;

    auto issue_stage = [&](int s) {
        v4i* stage = lds[s % NS];
#pragma unroll
        for (int it = 0; it < WIT; ++it) {
            const char* src = w_src[it] ? w_src[it] + (size_t)s * (CPR * 16) : zero;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(stage + it * NT + wave * 64),
                                             16, 0, 0);
        }
        const bool tap_ok = cur_i < a.kh;
#pragma unroll
        for (int it = 0; it < XIT; ++it) {
            const int ih = x_ih0[it] + cur_i * a.dil_h;
            const int iw = x_iw0[it] + cur_j * a.dil_w;
            const bool ok = x_ok[it] && tap_ok && (ih >= 0) && (ih < a.H) && (iw >= 0) && (iw < a.W);
            const char* src = ok ? (const char*)a.x + ((size_t)x_base[it] + (size_t)(ih * a.W + iw) * a.C + cur_c) * ES
                                 : zero;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(stage + WROWS * CPR + it * NT + wave * 64),
                                             16, 0, 0);
        }
        cur_c += ESTAGE;
        while (cur_c >= a.C) {
            cur_c -= a.C;
            if (++cur_j == a.kw) { cur_j = 0; ++cur_i; }
        }
    };

    const int frow = lane & 15, fq = lane >> 4;
    const int kb = k_base + wm * (TM * 16) + fq * NV;

    acc_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = acc_t{0, 0, 0, 0};

    const int steps = a.steps;
    const int pre = steps < NS - 1 ? steps : NS - 1;
    SABER_TL(1);
    for (int s = 0; s < pre; ++s) issue_stage(s);
    SABER_TL(2);
    // per-channel epilogue constants, requested behind the ring prefetch (their pointers are in the cold part of the
    // argument block; see conv_igemm_impl.h). These ordinary loads sit between the prefetched stages and the later refills
    // in the in-order VMEM queue, so the counted waits below (which let only the YOUNGEST n requests stay outstanding) can
    // only wait longer than needed at the first stages, never too little.
    ChanParams<NV> cp;
    load_chan_params<NV>(a, kb, cp);

    const unsigned xmask = (!F32 && a.in_u8) ? 0x80808080u : 0u;
    // fragment chunk indices for this wave group's k-step 0; phys_chunk(row, c ^ (ks*4)) == phys_chunk(row, c) ^ (ks*4)
    // because the group base (grp*KS*4, a multiple of 16 when WG > 1) and fq do not touch chunk bits 2..3
    int a_idx[TM], b_idx[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int row = (wm * TM + i) * 16 + frow;
        a_idx[i] = row * CPR + phys_chunk<CPR>(row, grp * KS * 4 + fq);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int row = (wn * TN + j) * 16 + frow;
        b_idx[j] = (WROWS + row) * CPR + phys_chunk<CPR>(row, grp * KS * 4 + fq);
    }
    for (int s = 0; s < steps; ++s) {
        // stages s+1 .. min(s+NS-2, steps-1) may stay in flight
        const int ahead = (steps - 1 - s) < (NS - 2) ? (steps - 1 - s) : (NS - 2);
        wait_vmcnt(ahead * LPS);
        __builtin_amdgcn_s_barrier();    // stage s landed for every wave; everyone finished reading stage s-1
        if (s + NS - 1 < steps) issue_stage(s + NS - 1);   // refills the buffer consumed in iteration s-1
        const v4i* stage = lds[s % NS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            v4i af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = stage[a_idx[i] ^ (ks << 2)];
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                v4i v = stage[b_idx[j] ^ (ks << 2)];
                if (!F32) { v.x ^= xmask; v.y ^= xmask; v.z ^= xmask; v.w ^= xmask; }
                bf[j] = v;
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = mma_step(af[i], bf[j], acc[i][j]);
        }
    }

    SABER_TL(3);
    // ---- intra-block split-K: groups 1..WG-1 hand their partial accumulators to group 0 via LDS ------
    if constexpr (WG > 1) {
        __syncthreads();                 // all DMA consumed (every wait above ended at vmcnt(0)), ring is free
        acc_t* red = (acc_t*)&lds[0][0];
        constexpr int PER_GRP = 256 * TM * TN;
        if (grp > 0) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) red[(grp - 1) * PER_GRP + (i * TN + j) * 256 + (tid & 255)] = acc[i][j];
        }
        __syncthreads();
        if (grp > 0) return;
        SABER_TL(5);
#pragma unroll
        for (int g = 1; g < WG; ++g)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] += red[(g - 1) * PER_GRP + (i * TN + j) * 256 + tid];
    }

    // ---- epilogue (identical to the register-staged kernel) ---------------------------------------
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int p = pix_base + (wn * TN + j) * 16 + frow;
        if constexpr (F32) {
            float v[NV];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[i * 4 + r] = acc[i][j][r];
            int n = 0, sp = 0;
            if (a.out_nchw || a.res_mode == RES_SUM_INPLACE) fast_divmod(p < a.M ? p : 0, ohw, a.inv_ohw, n, sp);
            if (a.pool_ow) epilogue_f32_pool2<NV>(a, v, cp, p, kb, lane);
            else if (a.K2 > 0) epilogue_f32_pair<NV>(a, v, cp, p, kb);
            else epilogue_f32<NV>(a, v, cp, p, kb, n, sp);
        } else {
            int v[NV];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[i * 4 + r] = acc[i][j][r];
            if constexpr (EK == EK_GEN) {
                epilogue_i8<NV>(a, v, cp, p, kb);
            } else if constexpr (EK == EK_PAIR) {
                if (p < a.M) epilogue_i8_pair<NV>(a, v, cp, p, kb);
            } else {
                // conv_igemm.hip:epilogue_kind sends K % 16 != 0 to EK_GEN, so every lane group here is whole
                if (p < a.M && kb < a.K) {
                    epilogue_i8_fast<NV, EK>(a, v, cp, p, kb);
                }
            }
        }
    }
    SABER_TL(4);
    SABER_TL_FLUSH();
}

template <int MODE, int KS, int EK, int WG>
static hipError_t launch_dma_mode(int tile, const ConvKArgs& a, hipStream_t s) {
    int bmk, bnp;
    tile_dims(tile, &bmk, &bnp);
    ConvKArgs b = a;
    b.npx = (a.M + bnp - 1) / bnp;
    b.nky = (a.K + bmk - 1) / bmk;
    b.mg_npx = magic_div(b.npx, (long long)b.npx * b.nky);
    dim3 grid(b.npx * b.nky);
    dim3 block(256 * WG);
    switch (tile) {
    case TILE_32x32: hipLaunchKernelGGL((conv_igemm_dma_kernel<MODE, 1, 1, KS, EK, WG>), grid, block, 0, s, b); break;
    case TILE_64x32:   // 4 wave groups would need > 160 KiB of LDS
        if constexpr (WG <= 2) { hipLaunchKernelGGL((conv_igemm_dma_kernel<MODE, 2, 1, KS, EK, WG>), grid, block, 0, s, b); break; }
        return hipErrorInvalidValue;
    case TILE_64x64:
        if constexpr (WG <= 2) { hipLaunchKernelGGL((conv_igemm_dma_kernel<MODE, 2, 2, KS, EK, WG>), grid, block, 0, s, b); break; }
        return hipErrorInvalidValue;
    case TILE_128x64:
        if constexpr (WG == 1) { hipLaunchKernelGGL((conv_igemm_dma_kernel<MODE, 4, 2, KS, EK, 1>), grid, block, 0, s, b); break; }
        return hipErrorInvalidValue;
    case TILE_64x128:
        if constexpr (WG == 1) { hipLaunchKernelGGL((conv_igemm_dma_kernel<MODE, 2, 4, KS, EK, 1>), grid, block, 0, s, b); break; }
        return hipErrorInvalidValue;
    case TILE_128x128:
        if constexpr (WG == 1) { hipLaunchKernelGGL((conv_igemm_dma_kernel<MODE, 4, 4, KS, EK, 1>), grid, block, 0, s, b); break; }
        return hipErrorInvalidValue;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// wg: wave groups (1, 2 or 4). wg > 1 exists for stage depth 4 on the 32x32 / 64x32 / 64x64 tiles.
template <int MODE, int EK>
static hipError_t launch_igemm_dma_inst(int tile, int ks, int wg, const ConvKArgs& a, hipStream_t s) {
    if (wg == 2) return ks == 4 ? launch_dma_mode<MODE, 4, EK, 2>(tile, a, s) : hipErrorInvalidValue;
    if (wg == 4) return ks == 4 ? launch_dma_mode<MODE, 4, EK, 4>(tile, a, s) : hipErrorInvalidValue;
    switch (ks) {
    case 1: return launch_dma_mode<MODE, 1, EK, 1>(tile, a, s);
    case 2: return launch_dma_mode<MODE, 2, EK, 1>(tile, a, s);
    case 4: return launch_dma_mode<MODE, 4, EK, 1>(tile, a, s);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace saber_mi355x
