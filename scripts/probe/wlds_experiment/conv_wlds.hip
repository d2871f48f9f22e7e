// anakin_amd/csrc/conv_wlds.hip - FP32 3x3 (stride 1 / pad 1) and 1x1 (stride 1) convolution on the bf16 matrix cores with the WEIGHT planes
// shared through LDS and the ACTIVATIONS straight in registers: the form between the implicit-GEMM kernel (both operands through LDS, one
// barrier and 72 KB of LDS reads per 32-deep stage: LDS-bound) and conv1x1_pwk.hip (nothing shared: bound by the bytes a CU can pull).
//
// Role: SaberConv2D<AK_FLOAT> / SaberConvEltwise<AK_FLOAT> on ResNet's 3x3 layers of res3 / res4 / res5 and its deep pointwise layers (the
// reference: SaberConv2D<X86, AK_FLOAT> -> jit / gemm kernels, saber/funcs/impl/x86/saber_conv.cpp:37-165).
//
// Shape of the work: a workgroup owns 64 output channels x 64 P pixels (a run of CONSECUTIVE pixels of the [N H W] list) and a slice of
// the reduction; a STAGE is one (32-channel chunk, filter tap) pair.
//   * Weights: the stage's 4 x 3 fragments (16 channels x 32 deep x three bf16 planes, 1 KB each, already in MFMA A-operand order:
//     saber_hip_conv::d_w3h1, the planes of conv3x3_b3h.hip) go global -> LDS by DMA, wave w copies tile w's three planes; a ring of
//     DX + 1 stages; every wave reads all twelve (12 KB of LDS reads per wave and stage for 48 MFMAs - the implicit-GEMM kernel needs
//     9 KB per 12).
//   * Activations: wave w owns pixel groups w P .. w P + P - 1 (16 pixels each); a lane requests its pixel's eight channels of the chunk
//     at the tap's offset (buffer loads: outside the image the offset is beyond the buffer = zeros, no branch, no zero page), DX stages
//     ahead, and splits them into the three bf16 planes in registers (x = h + m + l exactly). A tap re-reads what its neighbours read:
//     L1 hits.
//   * One barrier per stage (the ring slot of the stage before is free, every wave's DMA of this stage has landed).
//   * Few pixels (res4: 1 568, res5: 392 at batch 8) = few tiles: the reduction is split over 2^ksplit_sh workgroups of one XCD exactly as
//     in conv_igemm_impl.h (partials through that XCD's L2, the last arrival sums in split order: deterministic).
// Arithmetic: six plane products per stage in mma_step3's order (small terms first), term-major over the wave's 4 x P accumulators; stages
// ascending (chunk-major, taps row-major inside a chunk). Epilogue = epilogue_f32: d = acc; [d += y_old]; d += bias; relu / leaky.
// Differs from the other FP32 kernels in accumulation order only: inside the 1e-4 FP32 tolerance.
#include "coop_sync.h"
#include "conv_igemm_impl.h"

namespace saber_mi355x {

template <int KS, int P, int DX, int MINB, int NFIX>
__global__ __launch_bounds__(256, MINB) void conv_wlds_kernel(const ConvKArgs a) {
    constexpr int TM = 4, T = KS * KS;
    constexpr int R = DX + 1;                        // ring slots (LDS) and activation register buffers
    constexpr int L = 3 + 2 * P;                     // vector-memory instructions a wave issues per stage
    __shared__ v4i lds[R][TM * 3][64];               // 12 KB per stage
    SABER_TL_DECL;
    SABER_TL(0);
    const int M = a.M, K = a.K, C = a.C, W = a.W, H = a.H;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fq = lane >> 4;
    int ptile, tky, split = 0, tile_L = 0;
    if (a.ksplit_sh > 0) {
        if (!xcd_tile_split(a, ptile, tky, split, tile_L)) return;
    } else xcd_tile(a, ptile, tky);
    const int kbase = tky * 64;
    const int NS = C >> 5, nst_all = NS * T;
    // this workgroup's slice of the stages
    int q0 = 0, q1 = nst_all;
    if (a.ksplit_sh > 0) {
        const int spp = (nst_all + (1 << a.ksplit_sh) - 1) >> a.ksplit_sh;
        q0 = split * spp;
        q1 = q0 + spp < nst_all ? q0 + spp : nst_all;
    }
    const int n = q1 > q0 ? q1 - q0 : 0;

    // ---- this lane's pixels: byte offset of (pixel, channel 8 fq) and the taps that fall inside the image ----
    constexpr unsigned OOB = 0x7ffffff0u;            // beyond the buffer: the load returns zeros
    int xoff[P];
    unsigned tmask[P];
#pragma unroll
    for (int g = 0; g < P; ++g) {
        const int p = ptile * (64 * P) + (wave * P + g) * 16 + frow;
        const bool ok = p < M;
        xoff[g] = ((ok ? p : 0) * C + fq * 8) * 4;
        unsigned m = 0;
        if (KS == 1) m = ok ? 1u : 0u;
        else {
            int nimg, rem, y, x;
            fast_divmod(ok ? p : 0, H * W, a.inv_ohw, nimg, rem);
            fast_divmod(rem, W, a.inv_ow, y, x);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int iy = y + t / 3 - 1, ix = x + t % 3 - 1;
                if (ok && iy >= 0 && iy < H && ix >= 0 && ix < W) m |= 1u << t;
            }
        }
        tmask[g] = m;
    }
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), 0, (unsigned)M * (unsigned)C * 4u, 0x00020000);
    // weights: tile (tky * 4 + wave), stage q, plane pl at  ((tile * nst_all + q) * 3 + pl) KB; this lane's 16 bytes of each fragment
    const char* const wsrc = (const char*)a.w + ((size_t)(tky * TM + wave) * nst_all + q0) * 3072 + lane * 16;

    v4f acc[TM][P];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int g = 0; g < P; ++g) acc[i][g] = v4f{0.f, 0.f, 0.f, 0.f};

    // stage i of this workgroup (0 .. n - 1) = global stage q0 + i = (chunk, tap)
    auto issue = [&](int i, int slot, v4i (&xv)[P][2]) {
        const char* ws = wsrc + (size_t)i * 3072;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ws + pl * 1024),
                                             (__attribute__((address_space(3))) void*)&lds[slot][wave * 3 + pl][0], 16, 0, 0);
        const int q = q0 + i;
        const int cc = KS == 1 ? q : q / 9, t = KS == 1 ? 0 : q - cc * 9;      // (scalar: q is wave-uniform)
        const int toff = KS == 1 ? 0 : (((t / 3) - 1) * W + (t % 3) - 1) * C * 4;
#pragma unroll
        for (int g = 0; g < P; ++g) {
            const unsigned vo = ((tmask[g] >> t) & 1u) ? (unsigned)(xoff[g] + toff) : OOB;
            xv[g][0] = __builtin_amdgcn_raw_buffer_load_b128(xrs, (int)vo, cc * 128, 0);
            xv[g][1] = __builtin_amdgcn_raw_buffer_load_b128(xrs, (int)vo + 16, cc * 128, 0);
        }
    };
    // The stage's twelve fragments, LDS -> registers, by inline assembly: an LDS read the compiler can SEE while a DMA is pending gets an
    // s_waitcnt vmcnt(0) in front of it ("pending flat": the DMA might write what is read) - every stage would wait for the loads of the
    // stages behind it. The reads are issued here and waited for in read_w_wait (after the next stage's requests have been issued).
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) v4i*)&lds[0][0][0] + lane * 16;
    auto read_w = [&](int slot, v4i (&wv)[TM][3]) {
        const unsigned ad = lds0 + slot * (TM * 3 * 1024);
        asm volatile("ds_read_b128 %0, %12\n\tds_read_b128 %1, %12 offset:1024\n\tds_read_b128 %2, %12 offset:2048\n\t"
                     "ds_read_b128 %3, %12 offset:3072\n\tds_read_b128 %4, %12 offset:4096\n\tds_read_b128 %5, %12 offset:5120\n\t"
                     "ds_read_b128 %6, %12 offset:6144\n\tds_read_b128 %7, %12 offset:7168\n\tds_read_b128 %8, %12 offset:8192\n\t"
                     "ds_read_b128 %9, %12 offset:9216\n\tds_read_b128 %10, %12 offset:10240\n\tds_read_b128 %11, %12 offset:11264"
                     : "=&v"(wv[0][0]), "=&v"(wv[0][1]), "=&v"(wv[0][2]), "=&v"(wv[1][0]), "=&v"(wv[1][1]), "=&v"(wv[1][2]),
                       "=&v"(wv[2][0]), "=&v"(wv[2][1]), "=&v"(wv[2][2]), "=&v"(wv[3][0]), "=&v"(wv[3][1]), "=&v"(wv[3][2])
                     : "v"(ad)
                     : "memory");
    };
    auto read_w_wait = [&](v4i (&wv)[TM][3]) {
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(wv[0][0]), "+v"(wv[0][1]), "+v"(wv[0][2]), "+v"(wv[1][0]), "+v"(wv[1][1]), "+v"(wv[1][2]),
                       "+v"(wv[2][0]), "+v"(wv[2][1]), "+v"(wv[2][2]), "+v"(wv[3][0]), "+v"(wv[3][1]), "+v"(wv[3][2])
                     :
                     : "memory");
    };
    auto split_planes = [&](const v4i (&xv)[P][2], v4i (&bp)[P][3]) {
#pragma unroll
        for (int g = 0; g < P; ++g) {
            const v4f f0 = __builtin_bit_cast(v4f, xv[g][0]), f1 = __builtin_bit_cast(v4f, xv[g][1]);
            unsigned h[4], m[4], l[4];
            split3_pair(f0.x, f0.y, h[0], m[0], l[0]);
            split3_pair(f0.z, f0.w, h[1], m[1], l[1]);
            split3_pair(f1.x, f1.y, h[2], m[2], l[2]);
            split3_pair(f1.z, f1.w, h[3], m[3], l[3]);
            bp[g][0] = v4i{(int)h[0], (int)h[1], (int)h[2], (int)h[3]};
            bp[g][1] = v4i{(int)m[0], (int)m[1], (int)m[2], (int)m[3]};
            bp[g][2] = v4i{(int)l[0], (int)l[1], (int)l[2], (int)l[3]};
        }
    };
    auto mma = [&](const v4i (&wv)[TM][3], const v4i (&bp)[P][3]) {
        constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};      // (weights, activations): small terms first
#pragma unroll
        for (int tt = 0; tt < 6; ++tt)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int g = 0; g < P; ++g)
                    acc[i][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, wv[i][PA[tt]]),
                                                                        __builtin_bit_cast(v8bf, bp[g][PB[tt]]), acc[i][g], 0, 0, 0);
    };
    auto combine = [&](const v4i (&wv)[TM][3], const v4i (&xv)[P][2]) {
        v4i bp[P][3];
        split_planes(xv, bp);
        mma(wv, bp);
    };

    v4i xv[R][P][2];
    v4i wv[TM][3];
    // Ring: stage i's weights in slot i % R, its activations in register buffer i % R, DX stages in flight. Per stage: wait for this wave's
    // loads of the stage (counted: everything this wave issued later may still be in flight), barrier (the slot of stage i - 1 is free,
    // everybody's DMA of stage i has landed), issue stage i + DX, combine stage i. When nothing is left to issue the loop falls into a
    // drain that only combines (wait counts shrink with the loads behind).
    if constexpr (NFIX != 0) {
        // exactly NFIX stages (launcher): straight-line code, SOFTWARE-PIPELINED inside the wave. (Straight-line: in a loop the compiler's
        // wait-count pass merges the back edge conservatively and, with a DMA pending, turns every wait it needs itself into s_waitcnt
        // vmcnt(0).) The first version did per stage: wait, barrier, LDS reads, issue, wait for the LDS reads, split the planes (~90 VALU
        // instructions), 48 MFMAs - one after the other at ONE wave per SIMD: 0.57 us per stage for 0.32 us of MFMAs. Now stage i's operands are
        // in registers when its MFMAs start, and UNDER them the wave reads stage i + 1's fragments from LDS and splits stage i + 1's planes.
        v4i wv2[2][TM][3], bp2[2][P][3];
#pragma unroll
        for (int j = 0; j < DX; ++j) issue(j, j, xv[j]);
        __builtin_amdgcn_sched_barrier(0);
        wait_vm_older_than<(DX - 1) * L>();                      // stage 0 has landed (this wave's part)
        __builtin_amdgcn_s_barrier();
        read_w(0, wv2[0]);
        __builtin_amdgcn_sched_barrier(0);
        if (DX < NFIX) issue(DX, DX % R, xv[DX % R]);
        __builtin_amdgcn_sched_barrier(0);
        read_w_wait(wv2[0]);
        split_planes(xv[0], bp2[0]);
        __builtin_amdgcn_sched_barrier(0);
        SABER_TL(1);
#pragma unroll
        for (int i = 0; i < NFIX; ++i) {
            if (i == 6) SABER_TL(2);
            if (i == 12) SABER_TL(3);
            if (i + 1 < NFIX) {
                // stage i + 1: landed for this wave (the stages requested after it may be in flight), then for all (barrier: also, everybody has
                // stage i's fragments in registers - its ring slot and activation buffer are free for stage i + 1 + DX)
                constexpr int behind = 0;
                if (NFIX - 2 - i >= DX - 1) wait_vm_older_than<(DX - 1) * L>();
                else if (NFIX - 2 - i == 2) wait_vm_older_than<2 * L>();
                else if (NFIX - 2 - i == 1) wait_vm_older_than<L>();
                else wait_vm_older_than<behind>();
                __builtin_amdgcn_s_barrier();
                read_w((i + 1) % R, wv2[(i + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
                if (i + 1 + DX < NFIX) issue(i + 1 + DX, (i + 1 + DX) % R, xv[(i + 1 + DX) % R]);
                __builtin_amdgcn_sched_barrier(0);
                split_planes(xv[(i + 1) % R], bp2[(i + 1) & 1]);        // (one scheduling region with the MFMAs below: the VALU work runs under them)
                mma(wv2[i & 1], bp2[i & 1]);
                __builtin_amdgcn_sched_barrier(0);
                read_w_wait(wv2[(i + 1) & 1]);
            } else {
                mma(wv2[i & 1], bp2[i & 1]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
#pragma unroll
    for (int j = 0; j < DX; ++j)
        if (j < n) issue(j, j, xv[j]);
    __builtin_amdgcn_sched_barrier(0);
    int left = n;
    for (;;) {
#pragma unroll
        for (int j = 0; j < R; ++j) {
            if (left <= DX) goto drain;
            wait_vm_older_than<(DX - 1) * L>();
            __builtin_amdgcn_s_barrier();            // (raw: __syncthreads() carries a fence = s_waitcnt vmcnt(0), the whole ring landed)
            read_w(j, wv);                           // (LDS reads BEFORE the next DMA is issued: with a DMA pending the compiler's wait-count
            __builtin_amdgcn_sched_barrier(0);       //  pass puts s_waitcnt vmcnt(0) in front of every LDS read - "pending flat")
            issue(n - left + DX, (j + DX) % R, xv[(j + DX) % R]);
            __builtin_amdgcn_sched_barrier(0);
            read_w_wait(wv);
            combine(wv, xv[j]);
            __builtin_amdgcn_sched_barrier(0);
            --left;
        }
    }
drain:
    {
        const int ph = (n - left) % R;               // `left` <= DX stages sit in consecutive slots starting at ph
#pragma unroll
        for (int j = 0; j < R; ++j) {
            if (ph == j) {
#pragma unroll
                for (int d = 0; d < DX; ++d) {
                    if (d < left) {
                        if (left - 1 - d >= 2) wait_vm_older_than<2 * L>();
                        else if (left - 1 - d == 1) wait_vm_older_than<L>();
                        else wait_vm_older_than<0>();
                        __builtin_amdgcn_s_barrier();
                        read_w((j + d) % R, wv);
                        read_w_wait(wv);
                        combine(wv, xv[(j + d) % R]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        }
    }
    }

    SABER_TL(4);
    if (a.ksplit_sh > 0) {
        // ---- split-K: partial accumulators -> this XCD's L2; the last arrival sums them in split order (conv_igemm_impl.h) ----
        const int S = 1 << a.ksplit_sh;
        v4f* pw = (v4f*)a.part + ((size_t)(tile_L * S + split) * 4 + wave) * (TM * P * 64) + lane;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int g = 0; g < P; ++g) pw[(i * P + g) * 64] = acc[i][g];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __shared__ unsigned s_old;
        __syncthreads();
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 7u;
        if (tid == 0) s_old = __hip_atomic_fetch_add(a.part_ctr + tile_L, 1u << (4u * xcc), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const unsigned arrived = (((s_old & 0x0f0f0f0fu) + ((s_old >> 4) & 0x0f0f0f0fu)) * 0x01010101u) >> 24;
        if (arrived != (unsigned)(S - 1)) {
            SABER_TL_FLUSH();
            return;
        }
        SABER_TL(5);
        if (tid == 0) a.part_ctr[tile_L] = 0u;       // re-armed for the next launch
        const L2Reader part_l2(a.part);
        const unsigned pr0 = (unsigned)(((size_t)(tile_L * S) * 4 + wave) * (TM * P * 64) + lane) * 16u;      // byte offset
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int g = 0; g < P; ++g) acc[i][g] = __builtin_bit_cast(v4f, part_l2.load16(pr0 + (i * P + g) * 1024u));
        for (int s2 = 1; s2 < S; ++s2) {
            const unsigned prs = pr0 + (unsigned)s2 * (4 * (TM * P * 64) * 16u);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int g = 0; g < P; ++g) acc[i][g] = acc[i][g] + __builtin_bit_cast(v4f, part_l2.load16(prs + (i * P + g) * 1024u));
        }
        if (s_old != (unsigned)(S - 1) << (4u * xcc)) {      // a split ran on another XCD: poison + count (the host switches the split off)
            if (tid == 0 && a.part_err) __hip_atomic_fetch_add(a.part_err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            const float nan = __builtin_nanf("");
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int g = 0; g < P; ++g) acc[i][g] = v4f{nan, nan, nan, nan};
        }
    }

    SABER_TL(6);
    // ---- epilogue: lane (pixel frow of group g, quarter fq) owns channels kbase + 16 i + 4 fq .. + 3 ----
    const int ohw = H * W;
#pragma unroll
    for (int g = 0; g < P; ++g) {
        const int p = ptile * (64 * P) + (wave * P + g) * 16 + frow;
        int nimg = 0, sp = 0;
        if (a.res_mode == RES_SUM_INPLACE) fast_divmod(p < M ? p : 0, ohw, a.inv_ohw, nimg, sp);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int kb = kbase + 16 * i + 4 * fq;
            if (p >= M || kb >= K) continue;
            ChanParams<4> cp;
            load_chan_params<4>(a, kb, cp);
            const float v[4] = {acc[i][g][0], acc[i][g][1], acc[i][g][2], acc[i][g][3]};
            epilogue_f32<4>(a, v, cp, p, kb, nimg, sp);
        }
    }
    SABER_TL(7);
    SABER_TL_FLUSH();
}

// Variant v = 1 .. 3 -> (pixel groups per wave, stages in flight, workgroups per CU)
bool conv_wlds_variant(int v, int* p, int* dx, int* minb) {
    static const int Tb[3][3] = {{2, 3, 2}, {2, 4, 1}, {4, 2, 1}};
    if (v < 1 || v > 3) return false;
    *p = Tb[v - 1][0]; *dx = Tb[v - 1][1]; *minb = Tb[v - 1][2];
    return true;
}
// 3x3 / stride 1 / pad 1 or 1x1 / stride 1 / pad 0, NHWC f32, C % 32 == 0 (1x1: C % 64 == 0, the condition d_w3h1 is packed under), K % 64 == 0
bool conv_wlds_ok(int ks, int m, int c, int k) {
    return (ks == 1 || ks == 3) && c >= 64 && c % (ks == 1 ? 64 : 32) == 0 && k >= 64 && k % 64 == 0 && m >= 1 &&
           (long long)m * (c > k ? c : k) * 4 < 0x7fffffe0ll;
}

template <int KS, int P, int DX, int MINB>
static hipError_t launch_wlds(ConvKArgs& k, hipStream_t s) {
    const int nst = (k.C >> 5) * KS * KS, spp = (nst + (1 << k.ksplit_sh) - 1) >> k.ksplit_sh;
    const bool fix18 = spp == 18 && (spp << k.ksplit_sh) == nst;      // every workgroup has exactly 18 stages: the straight-line form
    k.npx = (k.M + 64 * P - 1) / (64 * P);
    k.nky = k.K / 64;
    k.mg_npx = magic_div(k.npx, (long long)k.npx * k.nky);
    const int T = k.npx * k.nky;
    const dim3 grid((unsigned)(k.ksplit_sh > 0 ? (8 * ((T + 7) / 8)) << k.ksplit_sh : T));
    if (fix18) hipLaunchKernelGGL((conv_wlds_kernel<KS, P, DX, MINB, 18>), grid, dim3(256), 0, s, k);
    else hipLaunchKernelGGL((conv_wlds_kernel<KS, P, DX, MINB, 0>), grid, dim3(256), 0, s, k);
    return hipGetLastError();
}

// a.w: the fragment-ordered planes d_w3h1 ([16-row tile][chunk][tap][plane][lane] x 16 B)
hipError_t launch_conv_wlds(int variant, const ConvKArgs& a, hipStream_t s) {
    int p, dx, minb;
    const int ks = a.kh;
    if (!conv_wlds_variant(variant, &p, &dx, &minb) || a.kh != a.kw || !conv_wlds_ok(ks, a.M, a.C, a.K) || a.stride_h != 1 || a.stride_w != 1 ||
        a.pad_h != (ks - 1) / 2 || a.pad_w != (ks - 1) / 2 || a.dil_h != 1 || a.dil_w != 1 || a.out_nchw || a.K2 || a.pool_ow ||
        (a.res_mode != RES_NONE && a.res_mode != RES_SUM_INPLACE) || a.ksplit_sh > 3)
        return hipErrorInvalidValue;
    ConvKArgs k = a;
    if (ks == 3) {
        switch (variant) {
        case 1: return launch_wlds<3, 2, 3, 2>(k, s);
        case 2: return launch_wlds<3, 2, 4, 1>(k, s);
        default: return launch_wlds<3, 4, 2, 1>(k, s);
        }
    }
    switch (variant) {
    case 1: return launch_wlds<1, 2, 3, 2>(k, s);
    case 2: return launch_wlds<1, 2, 4, 1>(k, s);
    default: return launch_wlds<1, 4, 2, 1>(k, s);
    }
}

}  // namespace saber_mi355x
