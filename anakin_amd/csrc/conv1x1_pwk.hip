// anakin_amd/csrc/conv1x1_pwk.hip - FP32 pointwise (1x1 / stride 1) convolution with MANY input channels (C = 128 .. 2048) on the bf16
// matrix cores without LDS staging and without a barrier in the reduction loop: the four waves of a workgroup split the REDUCTION.
//
// Role: SaberConv2D<AK_FLOAT> / SaberConvEltwise<AK_FLOAT> on ResNet's deep pointwise layers - `branch2a` of res2b .. res5c (256 -> 64 ...
// 2048 -> 512) and `branch2c` (+ in-place sum + relu) of res4 / res5 (256 -> 1024, 512 -> 2048): the role of the reference's SaberConv1X1
// (saber/funcs/impl/x86/saber_conv_eltwise.cpp:40-151; NV: conv_gemm_k1s1p0, third-party/sass/include/sass_funcs.h:699-781).
//
// Why: these layers take 15 - 19 us each at batch 8 for 2 - 5 us of matrix-core work. The kernels that run them keep one
// 32-deep slab of the reduction in flight per workgroup: load -> split into planes -> LDS -> barrier -> MFMA, 0.47 - 0.56 us per slab for
// 0.16 us of MFMAs, 32 - 64 slabs in a row - a latency chain whose link is one global-memory round trip. Here a wave owns every FOURTH slab
// of its workgroup's tile (TM x 16 output channels x P x 16 pixels) and keeps D slabs in flight on its own: the slab's weight fragments
// arrive straight in MFMA A-operand registers (the fragment-ordered planes of conv1x1_pw.hip, saber_hip_conv::d_wpw: one 1 KB load per
// fragment and wave), the slab's activations in the lanes that split them into the three bf16 planes in registers (x = h + m + l exactly),
// six plane products per slab in mma_step3's order, term-major over the wave's TM x P accumulators. Nothing is shared, so nothing waits for
// another wave until the four partial accumulators meet in LDS ONCE at the end (summed in wave order: deterministic), wave (t mod 4)
// finishes tile t (residual and bias requested before the reduction loop starts).
// Epilogue = epilogue_f32 of conv_igemm_impl.h: d = acc; [d += y_old]; d += bias; relu / leaky. The accumulation order over input channels
// differs from the implicit-GEMM kernels' (and from MKL's): inside the 1e-4 FP32 tolerance, like every FP32 kernel here.
//
// Measured (ResNet50 FP32, batch 8, cold operands, profiles/r05/pwk_autotune.txt): 13.0 - 14.8 us where the incumbents took 15.0 - 18.0
// (1024 -> 256 at 14 x 14: 17.4 -> 13.0; 512 -> 2048 + sum at 7 x 7: 17.3 -> 13.4; 2048 -> 512 at 7 x 7: 18.6 -> 16.2), 19 of the
// network's 21 deep pointwise layers select it, 8 350 -> 8 740 - 8 900 images/s. What bounds it is NOT the latency chain any more: one, two
// or three slabs in flight per wave time the same (with exact wait counts: straight-line code per C, below). A slab is 16 KB per wave
// through the CU's vector-memory path (TM x 3 KB of weight planes + P x 2 KB of activations, nothing shared between the waves) and a
// workgroup moves 0.5 - 1 MB that way at 60 - 75 GB/s per CU; fewer bytes per MFMA needs 64 x 64 per WAVE (more registers than a wave
// has beside D slabs in flight) or weights shared through LDS (the barrier again), more CUs need a reduction split across workgroups.
// Two things the compiler does to such a loop, both found in the ISA: (i) with 64-bit global addresses the register allocator recycles
// the destination of a PENDING load as an address temporary and the loop header waits for the whole ring (buffer loads: per-lane
// offsets set once + a scalar offset per slab); (ii) the wait-count pass merges a loop's back edge conservatively - the first slab of
// every round waited for ALL slabs in flight - hence the fully unrolled forms for C = 256 / 512 / 1024 / 2048.
#include "conv_igemm_impl.h"

namespace saber_mi355x {

template <int TM, int P, int D, bool SUM, int MINB, int NSW>
__global__ __launch_bounds__(256, MINB) void conv1x1_pwk_kernel(const ConvKArgs a) {
    constexpr int R = D + 1;                         // register buffers: D slabs in flight behind the one being combined
    constexpr int NT = TM * P, FT = NT / 4;          // output tiles of the workgroup; tiles a wave finishes
    static_assert(NT % 4 == 0 && (P & (P - 1)) == 0, "tiles");
    extern __shared__ v4f pwk_red[];                 // [4 waves][NT][64 lanes]
    const int M = a.M, K = a.K, C = a.C;
    const float* __restrict__ x = (const float*)a.x;
    float* y = (float*)a.y;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int frow = lane & 15, fq = lane >> 4;
    int ptile, tky;
    xcd_tile(a, ptile, tky);
    const int p0 = ptile * (16 * P), kbase = tky * (16 * TM);
    const int NS = C >> 5, nsw = NSW ? NSW : NS >> 2;      // slabs; slabs per wave (launcher: C % 128 == 0; NSW != 0: C == 128 NSW)

    // what this wave finishes at the end: tiles t = wave + 4 j; their residual values and bias are requested now
    float4 rs[FT], bs[FT];
    int fo[FT];
#pragma unroll
    for (int j = 0; j < FT; ++j) {
        const int t = wave + 4 * j, i = t / P, g = t % P;
        const int p = p0 + 16 * g + frow, kb = kbase + 16 * i + 4 * fq;
        const int pc = p < M ? p : M - 1;
        fo[j] = p < M ? pc * K + kb : -1;
        bs[j] = a.bias ? *(const float4*)(a.bias + kb) : make_float4(0.f, 0.f, 0.f, 0.f);
        if (SUM) rs[j] = *(const float4*)(y + (size_t)pc * K + kb);
        else rs[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    int xo[P];
#pragma unroll
    for (int g = 0; g < P; ++g) {
        const int p = p0 + 16 * g + frow;
        xo[g] = ((p < M ? p : M - 1) * C + fq * 4) * 4;      // bytes
    }
    v4f acc[TM][P];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int g = 0; g < P; ++g) acc[i][g] = v4f{0.f, 0.f, 0.f, 0.f};

    // Buffer loads: per-lane offsets that never change (VGPRs set once) + a per-slab scalar offset + an immediate - no vector address
    // arithmetic in the loop (with 64-bit global addresses the allocator recycled a pending load's destination as an address temporary and
    // the loop header drained the whole ring: s_waitcnt vmcnt(0) every R slabs)
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.w), 0, 0xffffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), 0, 0xffffffff, 0x00020000);
    const int wlo = lane * 16;
    const int wtile = tky * TM * NS * 3072;          // bytes to this workgroup's first 16-channel tile (launcher: K x C x 6 < 2^31)
    auto request = [&](int si, v4i (&wv)[TM][3], v4i (&xv)[P][2]) {
        const int s = wave + 4 * si;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int so = wtile + (i * NS + s) * 3072;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) wv[i][pl] = __builtin_amdgcn_raw_buffer_load_b128(wrs, wlo + pl * 1024, so, 0);
        }
#pragma unroll
        for (int g = 0; g < P; ++g) {
            xv[g][0] = __builtin_amdgcn_raw_buffer_load_b128(xrs, xo[g], s * 128, 0);
            xv[g][1] = __builtin_amdgcn_raw_buffer_load_b128(xrs, xo[g] + 64, s * 128, 0);
        }
    };
    auto split = [&](const v4i (&xv)[P][2], v4i (&bp)[P][3]) {
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
        // term-major: a plane product runs over the TM x P accumulators before the next product, so consecutive MFMAs never touch the
        // same accumulator (accumulator-major, each waits out its predecessor: conv_igemm_impl.h)
        constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};      // (weights, activations): small terms first
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int g = 0; g < P; ++g)
                    acc[i][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, wv[i][PA[t]]),
                                                                        __builtin_bit_cast(v8bf, bp[g][PB[t]]), acc[i][g], 0, 0, 0);
    };
    v4i wv[R][TM][3];
    v4i xv[R][P][2];
    v4i bp[P][3];
    // D slabs in flight (launcher: D <= slabs per wave).
    if constexpr (NSW != 0) {
        // C = 256 / 512 / 1024 / 2048: straight-line code, so every s_waitcnt counts exactly the loads behind the slab it needs (in a loop
        // the compiler's wait-count pass merges the back edge conservatively: the ring drained once per R slabs)
#pragma unroll
        for (int j = 0; j < D; ++j) request(j, wv[j], xv[j]);
        __builtin_amdgcn_sched_barrier(0);           // (left alone the scheduler sinks every request to just above its first use)
#pragma unroll
        for (int si = 0; si < NSW; ++si) {
            if (si + D < NSW) request(si + D, wv[(si + D) % R], xv[(si + D) % R]);
            __builtin_amdgcn_sched_barrier(0);
            split(xv[si % R], bp);
            mma(wv[si % R], bp);
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
        // any other C: the loop body requests slab si + D and combines slab si; when there is nothing left to request it falls into a
        // drain that only combines (a request past the end would be a wasted 16 KB per wave; a BRANCH around the request inside the loop
        // makes the compiler wait for every load at the next use)
#pragma unroll
        for (int j = 0; j < D; ++j) request(j, wv[j], xv[j]);
        __builtin_amdgcn_sched_barrier(0);
        int left = nsw;                              // slabs not yet combined
        for (;;) {
#pragma unroll
            for (int j = 0; j < R; ++j) {
                if (left <= D) goto drain;
                request(nsw - left + D, wv[(j + D) % R], xv[(j + D) % R]);
                __builtin_amdgcn_sched_barrier(0);
                split(xv[j], bp);
                mma(wv[j], bp);
                __builtin_amdgcn_sched_barrier(0);
                --left;
            }
        }
drain:
        {
            // `left` <= D slabs sit in consecutive buffers starting at phase ph = (nsw - left) % R
            const int ph = (nsw - left) % R;
#pragma unroll
            for (int j = 0; j < R; ++j) {
                if (ph == j) {
#pragma unroll
                    for (int d = 0; d < D; ++d) {
                        if (d < left) {
                            split(xv[(j + d) % R], bp);
                            mma(wv[(j + d) % R], bp);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                }
            }
        }
    }

    // ---- the four partial sums meet in LDS: [wave][tile][lane]; wave (t mod 4) sums tile t in wave order and finishes it ----
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int g = 0; g < P; ++g) pwk_red[((wave * NT) + i * P + g) * 64 + lane] = acc[i][g];
    __syncthreads();
    const bool relu = a.relu != 0;
    const float slope = a.neg_slope;
#pragma unroll
    for (int j = 0; j < FT; ++j) {
        const int t = wave + 4 * j;
        v4f s = pwk_red[(0 * NT + t) * 64 + lane];
#pragma unroll
        for (int w = 1; w < 4; ++w) s += pwk_red[(w * NT + t) * 64 + lane];
        float o[4] = {s[0], s[1], s[2], s[3]};
        const float r4[4] = {rs[j].x, rs[j].y, rs[j].z, rs[j].w};
        const float b4[4] = {bs[j].x, bs[j].y, bs[j].z, bs[j].w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float d = o[r];
            if (SUM) d = __fadd_rn(d, r4[r]);
            d = __fadd_rn(d, b4[r]);
            if (relu) d = d > 0.f ? d : (slope == 0.f ? 0.f : __fmul_rn(d, slope));
            o[r] = d;
        }
        if (fo[j] >= 0) *(float4*)(y + fo[j]) = make_float4(o[0], o[1], o[2], o[3]);
    }
}

// Variant v = 1 .. 4 -> (16-channel tiles, 16-pixel groups, slabs in flight, workgroups per CU)
bool conv1x1_pwk_variant(int v, int* tm, int* p, int* d, int* minb) {
    static const int T[4][4] = {{4, 2, 1, 2}, {2, 2, 3, 2}, {4, 2, 2, 1}, {4, 4, 2, 1}};
    if (v < 1 || v > 4) return false;
    *tm = T[v - 1][0]; *p = T[v - 1][1]; *d = T[v - 1][2]; *minb = T[v - 1][3];
    return true;
}
// C % 128 == 0 (every wave has at least one slab), K % 64 == 0; byte offsets are 32-bit
bool conv1x1_pwk_ok(int m, int c, int k) {
    return c >= 128 && c % 128 == 0 && k >= 64 && k % 64 == 0 && m >= 1 && (long long)m * (c > k ? c : k) * 4 < 0x7fffffffll && (long long)k * c * 6 < 0x7fffffffll;
}

template <int TM, int P, int D, int MINB, int NSW>
static void launch_pwk_n(const ConvKArgs& k, dim3 grid, size_t lds, hipStream_t s) {
    if (k.res_mode == RES_SUM_INPLACE) hipLaunchKernelGGL((conv1x1_pwk_kernel<TM, P, D, true, MINB, NSW>), grid, dim3(256), lds, s, k);
    else hipLaunchKernelGGL((conv1x1_pwk_kernel<TM, P, D, false, MINB, NSW>), grid, dim3(256), lds, s, k);
}
template <int TM, int P, int D, int MINB>
static hipError_t launch_pwk(ConvKArgs& k, hipStream_t s) {
    k.npx = (k.M + 16 * P - 1) / (16 * P);
    k.nky = k.K / (16 * TM);
    k.mg_npx = magic_div(k.npx, (long long)k.npx * k.nky);
    const int nsw = k.C >> 7;
    if (nsw < D) return hipErrorInvalidValue;
    const dim3 grid((unsigned)(k.npx * k.nky));
    const size_t lds = (size_t)4 * TM * P * 64 * sizeof(v4f);
    if (nsw == 2 && D <= 2) launch_pwk_n<TM, P, (D <= 2 ? D : 1), MINB, 2>(k, grid, lds, s);      // (C = 256: ResNet's res2 / res4 layers)
    else if (nsw == 4) launch_pwk_n<TM, P, D, MINB, 4>(k, grid, lds, s);
    else if (nsw == 8) launch_pwk_n<TM, P, D, MINB, 8>(k, grid, lds, s);
    else if (nsw == 16) launch_pwk_n<TM, P, D, MINB, 16>(k, grid, lds, s);
    else launch_pwk_n<TM, P, D, MINB, 0>(k, grid, lds, s);
    return hipGetLastError();
}

// 1x1 / stride 1 / pad 0, NHWC f32 in and out, no pair / pooling epilogue; a.w: the fragment-ordered planes (d_wpw)
hipError_t launch_conv1x1_pwk(int variant, const ConvKArgs& a, hipStream_t s) {
    int tm, p, d, minb;
    if (!conv1x1_pwk_variant(variant, &tm, &p, &d, &minb) || !conv1x1_pwk_ok(a.M, a.C, a.K) || a.kh != 1 || a.kw != 1 || a.stride_h != 1 ||
        a.stride_w != 1 || a.pad_h || a.pad_w || a.out_nchw || a.K2 || a.pool_ow || (a.res_mode != RES_NONE && a.res_mode != RES_SUM_INPLACE))
        return hipErrorInvalidValue;
    ConvKArgs k = a;
    switch (variant) {
    case 1: return launch_pwk<4, 2, 1, 2>(k, s);
    case 2: return launch_pwk<2, 2, 3, 2>(k, s);
    case 3: return launch_pwk<4, 2, 2, 1>(k, s);
    default: return launch_pwk<4, 4, 2, 1>(k, s);
    }
}

}  // namespace saber_mi355x
