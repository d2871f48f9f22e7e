// anakin_amd/csrc/fc_small.hip — INT8 fully-connected layer for SMALL batches (m <= 16 rows), gfx950.
//
// Role: SaberFc / VenderFc<X86,AK_INT8> at inference batch sizes (reference: PackedMKLInt8Gemm::dispatch,
// mkl_packed_int8_gemm.cpp:46-97; VenderFc u8 path, vender_fc.cpp:284-335). Through the implicit-GEMM conv kernel a
// [8 x 2048] x [2048 x 1000] fc gets 16 workgroups that each walk a 4..8-stage reduction (7.8 us at batch 8): the
// layer is a 2 MB weight stream, so it wants every CU pulling a slice of the weights at once. Here
//   workgroup = 16 output channels, its 4 waves split the reduction 4 ways;
//   a wave loads its weight slice [16 rows][k/4] and the matching activation slice STRAIGHT into MFMA operand
//   registers (A: row = lane & 15 = output channel, B: row = lane & 15 = batch row, k-group = lane >> 4; all loads of
//   the wave in flight together: one exposed memory latency), runs v_mfma_i32_16x16x64_i8 over them and the four
//   partial accumulators are summed through LDS (integer sums: exact, order independent).
// Same epilogue arithmetic as the conv kernel's EPI_I8_FC_S8 / EPI_I8_FC_U8 branches (conv_igemm_impl.h).
#include "conv_igemm_impl.h"

namespace saber_mi355x {

// KSW: 64-byte k-steps per wave (reduction = 4 waves x KSW x 64, zero padded weights beyond Kg)
// SM (round 5, round-4 verdict item 2 ii): the Softmax operator that follows the fc (saber_softmax.cpp role; softmax_f32_kernel's
// arithmetic per row: max, exp(x - max), sum, divide) in the SAME launch - the fc's 63 workgroups each hold 16 of an image's 1000 logits,
// so the workgroup that arrives LAST on a device-wide counter normalises the rows (one wave per row, 16 logits per lane, wave shuffles).
// Round 1 tried this with an agent-scope RELEASE fence per workgroup (an L2 write-back: 33.9 us against 8.7 for the two launches); here
// the hand-off is the one the guide measures at ~1 us: the logits leave as agent-scope (write-through) stores, `s_waitcnt vmcnt(0)`,
// one returning agent-scope atomic on the counter, and the last workgroup reads them back with SYSTEM-scope loads (sc0 sc1: served from
// beyond its XCD's L2, whatever that L2 holds of the lines its own 64-byte stores touched). The counter is put back to zero by the last
// workgroup. The logits tensor is written as before (it is an edge of the op list); a row's sum runs lane-major instead of
// softmax_f32_kernel's thread-major order: same values within the 1e-4 the softmax output is held to everywhere.
struct FcSoftmaxTail {
    float* prob;        // [M][K]
    unsigned* ctr;      // one word, zero between launches
};
// FENCED (round 6, round-5 advisor): the same hand-off in the textbook form - plain visibility through an agent-scope RELEASE fence before
// the arrival and an agent-scope ACQUIRE fence in the last workgroup (L2 write-back + L1 invalidate: the 1.7 - 6.5 us rows of the guide's
// table) - selected by SABER_HIP_FC_SOFTMAX_FENCED=1 as the A/B and fall-back for a runtime / part where the write-through form's timing
// assumptions (sc1 stores acknowledged at the device's coherence point before `s_waitcnt vmcnt(0)` returns) should not hold. The default
// form is the guide's "sc1 payload -> asm vmcnt(0) -> device-scope counter, sc0 sc1 loads on the reader" (MI355X_MICROARCH.md, valid forms).
// ONE launch of an fc object may be in flight at a time (the counter word belongs to the object): the reference's contract - an impl
// instance is never called concurrently (threading) - and a Net's / plan's launches are stream-ordered.
template <int KSW, bool SM, bool FENCED = false>
__device__ __forceinline__ void fc_i8_small_body(const ConvKArgs& a, const FcSoftmaxTail& t) {
    __shared__ v4i red[3][64];
    __shared__ unsigned last_flag;
    SABER_TL_DECL;
    SABER_TL(0);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fq = lane >> 4;
    const int n0 = blockIdx.x * 16;
    const int m = frow < a.M ? frow : a.M - 1;            // rows beyond the batch re-read the last one (results dropped)
    const char* wp = (const char*)a.w + (size_t)(n0 + frow) * a.Kg_pad + (size_t)wave * (KSW * 64) + fq * 16;
    const char* xp = (const char*)a.x + (size_t)m * a.C + (size_t)wave * (KSW * 64) + fq * 16;
    v4i wf[KSW], xf[KSW];
#pragma unroll
    for (int s = 0; s < KSW; ++s) wf[s] = *(const v4i*)(wp + s * 64);
#pragma unroll
    for (int s = 0; s < KSW; ++s) {
        // the activation row is only C bytes long: k-steps beyond it multiply zero weights, fetch the zero page
        const bool in = (wave * KSW + s) * 64 + fq * 16 < a.C;
        xf[s] = *(const v4i*)(in ? xp + s * 64 : (const char*)a.zero);
    }
    const int kb = n0 + fq * 4;
    ChanParams<4> cp;
    load_chan_params<4>(a, kb, cp);
    const int xmask = a.in_u8 ? (int)0x80808080u : 0;
    SABER_TL(1);
    v4i acc = {0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s < KSW; ++s) {
        v4i b = xf[s];
        b.x ^= xmask; b.y ^= xmask; b.z ^= xmask; b.w ^= xmask;
        acc = mma_step(wf[s], b, acc);
    }
    SABER_TL(2);
    if (wave > 0) red[wave - 1][lane] = acc;
    __syncthreads();
    SABER_TL(3);
    if (wave == 0) {
        acc += red[0][lane];
        acc += red[1][lane];
        acc += red[2][lane];
        // lane: output channels kb..kb+3 of batch row frow
        if (frow < a.M && kb < a.K) {
            float out[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int v = acc[r] + cp.comp[r];
                const float d = (float)v;
                if (a.epi == EPI_I8_FC_S8) out[r] = __fadd_rn(__fmul_rn(d, cp.scale[r]), cp.bias[r]);
                else out[r] = (cp.scale[r] == 1.f) ? d : __fmul_rn(cp.scale[r], d);     // EPI_I8_FC_U8
            }
            float* y = (float*)a.y + (size_t)frow * a.K + kb;
            if constexpr (SM) {
                for (int r = 0; r < 4; ++r)
                    if (kb + r < a.K) __hip_atomic_store(y + r, out[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // write-through
            } else if (kb + 4 <= a.K && (a.K & 3) == 0) {
                *(float4*)y = make_float4(out[0], out[1], out[2], out[3]);
            } else {
                for (int r = 0; r < 4; ++r)
                    if (kb + r < a.K) y[r] = out[r];
            }
        }
    }
    SABER_TL(4);
    if constexpr (!SM) {
        SABER_TL_FLUSH();
        return;
    } else {
        if constexpr (FENCED) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                                          // every wave's logits are issued and acknowledged
            if (tid == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the compiler may drop the fence's own wait: guide, compiler hazard)
                last_flag = __hip_atomic_fetch_add(t.ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1 ? 1u : 0u;
                if (last_flag) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
        } else if (wave == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this workgroup's logits are at the device's coherence point
            if (lane == 0) last_flag = __hip_atomic_fetch_add(t.ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1 ? 1u : 0u;
        }
        __syncthreads();
        if (!last_flag) {
            SABER_TL_FLUSH();
            return;
        }
        SABER_TL(5);
        if (tid == 0) __hip_atomic_store(t.ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // for the next launch
        constexpr int PER = 16;                                   // logits per lane: rows of up to 1024 (checked by the launcher)
        // a wave normalises rows wave, wave + 4 (, + 8, + 12): two rows at a time, BOTH rows' 32 loads in flight before the first use - the
        // tail is one memory round trip per pair (the first version walked its rows one after the other: 4.5 us for two rows per wave,
        // profiles/r05/timeline_tail.txt)
        for (int row0 = wave; row0 < a.M; row0 += 8) {
            const int row1 = row0 + 4;
            const bool has1 = row1 < a.M;
            const float* y0 = (const float*)a.y + (size_t)row0 * a.K;
            const float* y1 = (const float*)a.y + (size_t)(has1 ? row1 : row0) * a.K;
            float v0[PER], v1[PER];
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const int c = lane + 64 * i;
                const int cc = c < a.K ? c : a.K - 1;             // (unconditional loads: all in flight together)
                v0[i] = __hip_atomic_load(y0 + cc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                v1[i] = __hip_atomic_load(y1 + cc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            auto normalise = [&](float (&v)[PER], int row) {
                float mx = -3.4e38f;
#pragma unroll
                for (int i = 0; i < PER; ++i) mx = lane + 64 * i < a.K ? fmaxf(mx, v[i]) : mx;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
                float sum = 0.f;
#pragma unroll
                for (int i = 0; i < PER; ++i) {
                    v[i] = lane + 64 * i < a.K ? expf(v[i] - mx) : 0.f;
                    sum += v[i];
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
                float* pr = t.prob + (size_t)row * a.K;
#pragma unroll
                for (int i = 0; i < PER; ++i)
                    if (lane + 64 * i < a.K) pr[lane + 64 * i] = v[i] / sum;
            };
            normalise(v0, row0);
            if (has1) normalise(v1, row1);
        }
        SABER_TL(6);
        SABER_TL_FLUSH();
    }
}
template <int KSW>
__global__ __launch_bounds__(256) void fc_i8_small_kernel(const ConvKArgs a) {
    fc_i8_small_body<KSW, false>(a, FcSoftmaxTail{nullptr, nullptr});
}
template <int KSW, bool FENCED>
__global__ __launch_bounds__(256) void fc_i8_small_softmax_kernel(const ConvKArgs a, const FcSoftmaxTail t) {
    fc_i8_small_body<KSW, true, FENCED>(a, t);
}

// FP32 fc (VenderFc<X86,AK_FLOAT>, vender_fc.cpp:154-212: out = in W^T + bias) at <= 16 batch rows, and Gemm<float> with a few rows
// against [n][k] weights (saber/funcs/gemm.h:27-66): ONE pass over the weights - VGG16's fc6 is a 411 MB stream for 1.6 GFLOP - on
// v_mfma_f32_16x16x4_f32. 16 outputs per workgroup, the 4 waves split the reduction; weight and activation chunks (16 floats per row per
// step) go straight into MFMA operand registers.
// Round 5 (round-4 verdict item 7: fc6 at 3.08 TB/s = 38 % of HBM, asked >= 60 %): the round-2 kernel loaded 16 + 16 chunks, waited for
// ALL of them, ran 64 MFMAs that all accumulate into ONE register quad (each waits out its predecessor's 8 passes) and only then asked for
// the next batch - load latency and MFMA time added up, 24 times per wave at fc6. Now:
//   * two register buffers of NS steps: the loads of buffer b ^ 1 are issued BEFORE the MFMAs of buffer b (the compiler's counted
//     s_waitcnt vmcnt leaves exactly those in flight) - 2 x NS KB of weights per wave in flight, 96 KB per CU at NS = 12;
//   * every load is unconditional (conditional loads make the wait-count pass drain the queue): beyond the reduction's end the weight
//     ADDRESS falls back to the row's start (finite values) and the activation address to the zero page;
//   * four independent accumulators (one per k-group of a chunk), summed once at the end;
//   * the weights are requested non-temporal (read once per pass; the activations, re-read by every workgroup, are not).
// Summation order differs from MKL's and from the round-2 kernel's - inside the 1e-4 FP32 tolerance like every FP32 path.
struct FcStreamArgs {
    const float* w;      // [n (padded or not)][w_pitch]
    const float* x;      // [m][c]
    float* y;            // [m][n]
    const float* bias;   // [n] or null
    const void* zero;    // >= 16 zero bytes
    int m, n, c, w_pitch;
    int w_rows;          // rows of w that may be read (rows beyond fall back to the last one; their outputs are not stored)
    int relu;
    float neg_slope, alpha, beta;      // y = act(alpha * acc + bias) (+ beta * y_old when beta != 0)
};

// PACKED: a.w is the fragment-major repack of the weights (api_conv.hip: set_weights, saber_hip_conv::d_wfc): [16-output tile][16-float
// step][lane][4 floats], lane (r = lane & 15, g = lane >> 4) holding W[tile * 16 + r][step * 16 + 4 g .. + 3], zero padded - one load
// instruction = 1 KB CONTIGUOUS, a wave's whole share of the reduction one contiguous run. With the weights as they lie ([n][k] rows, the
// Gemm entry point's raw B) an instruction gathers 16 rows x 64 bytes 100 KB apart: 16 K concurrent 64-byte streams leave HBM at ~50 %
// (3.9 TB/s on fc6 whether 96 or 192 KB are in flight per CU: profiles/r05/fc_stream.txt).
template <int NS, int NW, bool PACKED>
__global__ __launch_bounds__(NW * 64) void fc_f32_stream_kernel(const FcStreamArgs a) {
    __shared__ v4f redf[NW - 1][64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fq = lane >> 4;
    const int n0 = blockIdx.x * 16;
    const int m = frow < a.m ? frow : a.m - 1;               // rows beyond the batch re-read the last one (results dropped)
    const int wr = n0 + frow < a.w_rows ? n0 + frow : a.w_rows - 1;
    const int ksw = (a.c + NW * 16 - 1) / (NW * 16);         // 16-float steps per wave: the NW waves split the reduction
    const int k0 = wave * ksw * 16 + fq * 4;                 // this lane's first reduction index
    const float* const wrow = PACKED ? a.w + ((size_t)blockIdx.x * a.w_pitch * 64 + lane) * 4 : a.w + (size_t)wr * a.w_pitch;
    const float* const xrow = a.x + (size_t)m * a.c;
    const float* const zero = (const float*)a.zero;
    v4f acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = v4f{0.f, 0.f, 0.f, 0.f};
    v4i wf[2][NS], xf[2][NS];
    auto request = [&](int b, int s0) {                      // steps s0 .. s0 + NS - 1 of this wave into buffer b
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int k = k0 + (s0 + s) * 16;
            const bool in = s0 + s < ksw && k < a.c;
            // (address select, not a branch: the load itself is unconditional; packed weights are zero beyond the reduction's end)
            const float* wp = PACKED ? wrow + (size_t)(s0 + s < ksw ? wave * ksw + s0 + s : wave * ksw) * 256 : (in ? wrow + k : wrow);
            const float* xp = in ? xrow + k : zero;
            wf[b][s] = __builtin_nontemporal_load((const v4i*)wp);
            xf[b][s] = *(const v4i*)xp;
        }
    };
    auto multiply = [&](int b) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const v4f wv = __builtin_bit_cast(v4f, wf[b][s]);
            const v4f xv = __builtin_bit_cast(v4f, xf[b][s]);
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.x, xv.x, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.y, xv.y, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.z, xv.z, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.w, xv.w, acc[3], 0, 0, 0);
        }
    };
    // (scheduling barriers: left alone, the scheduler sinks every load next to its MFMA to save registers - one load in flight,
    // s_waitcnt vmcnt(0) in front of every MFMA group; with them the wait-count pass emits vmcnt(2 * NS) in front of a buffer's MFMAs)
    // Every workgroup walks ITS rows' reduction starting at a different chunk (rotated by its index, wrapping around): all 256 workgroups
    // start together and run at the same rate, and their rows are a multiple of 16 x c floats apart - with a common starting offset
    // they all ask the SAME few memory channels for their next lines at any moment (fc6: rows 100 352 bytes apart; both the round-2 and
    // the round-4 kernel sat at 3.1 - 4.2 TB/s). Order of summation per output: fixed by the workgroup index - deterministic.
    const int nch = (ksw + NS - 1) / NS;                     // chunks of NS steps per wave
    const int rot = (int)((blockIdx.x * 7u + wave * 3u) % (unsigned)nch);
    auto chunk_s0 = [&](int ci) { return ci < nch ? ((ci + rot) % nch) * NS : ksw; };      // (past the end: nothing in range, zeros)
    request(0, chunk_s0(0));
    __builtin_amdgcn_sched_barrier(0);
    for (int ci = 0; ci < nch; ci += 2) {
        request(1, chunk_s0(ci + 1));
        __builtin_amdgcn_sched_barrier(0);
        multiply(0);
        __builtin_amdgcn_sched_barrier(0);
        request(0, chunk_s0(ci + 2));
        __builtin_amdgcn_sched_barrier(0);
        multiply(1);
        __builtin_amdgcn_sched_barrier(0);
    }
    v4f sum = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    if (wave > 0) redf[wave - 1][lane] = sum;
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int w = 0; w < NW - 1; ++w) sum += redf[w][lane];
    const int kb = n0 + fq * 4;                              // lane: outputs kb .. kb + 3 of batch row frow
    if (frow >= a.m || kb >= a.n) return;
    float* y = a.y + (size_t)frow * a.n + kb;
    float out[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float d = a.alpha == 1.f ? sum[r] : __fmul_rn(a.alpha, sum[r]);
        if (a.bias && kb + r < a.n) d = __fadd_rn(d, a.bias[kb + r]);
        if (a.relu) d = d > 0.f ? d : (a.neg_slope == 0.f ? 0.f : __fmul_rn(d, a.neg_slope));
        if (a.beta != 0.f && kb + r < a.n) d = __fadd_rn(d, __fmul_rn(a.beta, y[r]));
        out[r] = d;
    }
    if (kb + 4 <= a.n && (a.n & 3) == 0) {
        *(float4*)y = make_float4(out[0], out[1], out[2], out[3]);
    } else {
        for (int r = 0; r < 4; ++r)
            if (kb + r < a.n) y[r] = out[r];
    }
}
// The same product for weights that must be read AS THEY LIE, row-major [n][k] - the raw B of Gemm<float>::dispatch, which may change
// between calls and cannot be repacked without a second pass: every load instruction reads 1 KB CONTIGUOUS of ONE row (lane l: floats
// 4 l .. 4 l + 3 of a 256-float super-step), sixteen instructions bring a 16-row x 256-float block into registers, the wave turns it
// into MFMA operand order through a wave-PRIVATE LDS region (row pitch 1040 bytes: the 16 lanes of a fragment column hit 16 distinct
// bank quads; no barrier - one wave's LDS operations execute in order), and the activations take the same route (MR contiguous 1 KB
// loads instead of 16 gathered ones). Two register buffers, unconditional loads, four accumulators, rotated start as above.
template <int MR>
__global__ __launch_bounds__(256) void gemm_f32_rows_lds_kernel(const FcStreamArgs a) {
    constexpr int PITCH = 260;                               // floats per LDS row (256 + 4: see above)
    __shared__ float lds[4][(16 + MR) * PITCH];
    __shared__ v4f redf[3][64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fq = lane >> 4;
    const int n0 = blockIdx.x * 16;
    float* const wl = lds[wave];
    float* const xl = lds[wave] + 16 * PITCH;
    const int nsup = (a.c + 1023) / 1024;                    // 256-float super-steps per wave (the four waves split the reduction)
    const int kw0 = wave * nsup * 256;                       // this wave's first reduction index
    const float* const zero = (const float*)a.zero;
    v4f acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = v4f{0.f, 0.f, 0.f, 0.f};
    v4i wv[2][16], xv[2][MR];
    auto request = [&](int b, int t) {                       // super-step t of this wave into buffer b
        const int k = kw0 + t * 256 + lane * 4;
        const bool in = t < nsup && k < a.c;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = n0 + r < a.w_rows ? n0 + r : a.w_rows - 1;
            const float* wrow = a.w + (size_t)row * a.w_pitch;
            wv[b][r] = __builtin_nontemporal_load((const v4i*)(in ? wrow + k : wrow));
        }
#pragma unroll
        for (int m = 0; m < MR; ++m) {
            const int mm = m < a.m ? m : a.m - 1;
            xv[b][m] = *(const v4i*)(in ? a.x + (size_t)mm * a.c + k : zero);
        }
    };
    auto multiply = [&](int b) {
#pragma unroll
        for (int r = 0; r < 16; ++r) *(v4i*)(wl + r * PITCH + lane * 4) = wv[b][r];
#pragma unroll
        for (int m = 0; m < MR; ++m) *(v4i*)(xl + m * PITCH + lane * 4) = xv[b][m];
        const float* wf = wl + frow * PITCH + fq * 4;
        const float* xf = xl + (frow < MR ? frow : frow - MR) * PITCH + fq * 4;     // (rows beyond the batch: any finite row, results dropped)
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const v4f wq = *(const v4f*)(wf + s * 16);
            const v4f xq = *(const v4f*)(xf + s * 16);
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq.x, xq.x, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq.y, xq.y, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq.z, xq.z, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq.w, xq.w, acc[3], 0, 0, 0);
        }
    };
    const int rot = (int)((blockIdx.x * 7u + wave * 3u) % (unsigned)nsup);
    auto sup = [&](int ci) { return ci < nsup ? (ci + rot) % nsup : nsup; };      // (past the end: nothing in range, zeros)
    request(0, sup(0));
    __builtin_amdgcn_sched_barrier(0);
    for (int ci = 0; ci < nsup; ci += 2) {
        request(1, sup(ci + 1));
        __builtin_amdgcn_sched_barrier(0);
        multiply(0);
        __builtin_amdgcn_sched_barrier(0);
        request(0, sup(ci + 2));
        __builtin_amdgcn_sched_barrier(0);
        multiply(1);
        __builtin_amdgcn_sched_barrier(0);
    }
    v4f sum = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    if (wave > 0) redf[wave - 1][lane] = sum;
    __syncthreads();
    if (wave > 0) return;
    sum += redf[0][lane];
    sum += redf[1][lane];
    sum += redf[2][lane];
    const int kb = n0 + fq * 4;
    if (frow >= a.m || kb >= a.n) return;
    float* y = a.y + (size_t)frow * a.n + kb;
    float out[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float d = a.alpha == 1.f ? sum[r] : __fmul_rn(a.alpha, sum[r]);
        if (a.bias && kb + r < a.n) d = __fadd_rn(d, a.bias[kb + r]);
        if (a.relu) d = d > 0.f ? d : (a.neg_slope == 0.f ? 0.f : __fmul_rn(d, a.neg_slope));
        if (a.beta != 0.f && kb + r < a.n) d = __fadd_rn(d, __fmul_rn(a.beta, y[r]));
        out[r] = d;
    }
    if (kb + 4 <= a.n && (a.n & 3) == 0) {
        *(float4*)y = make_float4(out[0], out[1], out[2], out[3]);
    } else {
        for (int r = 0; r < 4; ++r)
            if (kb + r < a.n) y[r] = out[r];
    }
}

bool fc_f32_small_ok(int m, int c, int kg_pad) { return m >= 1 && m <= 16 && c % 4 == 0 && c <= 65536 && kg_pad >= (c + 63) / 64 * 64; }
static hipError_t launch_fc_f32_stream(const FcStreamArgs& f, bool packed, hipStream_t s) {
    const dim3 grid((f.n + 15) / 16);
    // long reductions: 12 steps per buffer, two buffers (24 KB of weights per wave, 96 KB per CU in flight); short ones: 4 steps (less to
    // drain at the end). Eight waves per workgroup (192 KB in flight) measured SLOWER on the row-major weights (120 vs 104 us on fc6).
    const bool deep = (f.c + 63) / 64 >= 48;
    if (packed) {
        if (deep) hipLaunchKernelGGL((fc_f32_stream_kernel<12, 4, true>), grid, dim3(256), 0, s, f);
        else hipLaunchKernelGGL((fc_f32_stream_kernel<4, 4, true>), grid, dim3(256), 0, s, f);
    } else {
        if (deep) hipLaunchKernelGGL((fc_f32_stream_kernel<12, 4, false>), grid, dim3(256), 0, s, f);
        else hipLaunchKernelGGL((fc_f32_stream_kernel<4, 4, false>), grid, dim3(256), 0, s, f);
    }
    return hipGetLastError();
}
// a.w: the row-major repacked weights [K_pad][Kg_pad] - or, packed = true, their fragment-major form (see fc_f32_stream_kernel)
hipError_t launch_fc_f32_small(const ConvKArgs& a, bool packed, hipStream_t s) {
    if (!fc_f32_small_ok(a.M, a.C, a.Kg_pad)) return hipErrorInvalidValue;
    FcStreamArgs f;
    f.w = (const float*)a.w; f.x = (const float*)a.x; f.y = (float*)a.y; f.bias = a.bias; f.zero = a.zero;
    f.m = a.M; f.n = a.K; f.c = a.C;
    f.w_pitch = packed ? fc_f32_packed_steps(a.C) : a.Kg_pad;
    f.w_rows = (a.K + 15) / 16 * 16;        // (the repacked weights are padded to multiples of 128 rows)
    f.relu = a.relu; f.neg_slope = a.neg_slope; f.alpha = 1.f; f.beta = 0.f;
    if (!packed && a.C >= 2048) {      // long reductions on row-major weights: contiguous row reads + the wave-private LDS transpose
        const dim3 grid((f.n + 15) / 16), block(256);
        if (f.m <= 8) hipLaunchKernelGGL((gemm_f32_rows_lds_kernel<8>), grid, block, 0, s, f);
        else hipLaunchKernelGGL((gemm_f32_rows_lds_kernel<16>), grid, block, 0, s, f);
        return hipGetLastError();
    }
    return launch_fc_f32_stream(f, packed, s);
}
// 16-float steps per 16-output tile of the fragment-major weights: four waves x ceil(c / 64) steps each
int fc_f32_packed_steps(int c) { return 4 * ((c + 63) / 64); }
// Gemm<float> with m <= 16 rows of A against B stored [n][k] (trans_b): C[m][n] = alpha * A B^T + beta * C on raw pointers
bool gemm_f32_rows_ok(int m, int k) { return m >= 1 && m <= 16 && k % 4 == 0 && k <= (1 << 24); }
hipError_t launch_gemm_f32_rows(int m, int n, int k, float alpha, const float* A, const float* B, float beta, float* C, const void* zero,
                                hipStream_t s) {
    if (!gemm_f32_rows_ok(m, k)) return hipErrorInvalidValue;
    FcStreamArgs f;
    f.w = B; f.x = A; f.y = C; f.bias = nullptr; f.zero = zero;
    f.m = m; f.n = n; f.c = k; f.w_pitch = k; f.w_rows = n;
    f.relu = 0; f.neg_slope = 0.f; f.alpha = alpha; f.beta = beta;
    // long reductions: contiguous row reads turned into operand order through LDS (gemm_f32_rows_lds_kernel; SABER_HIP_GEMM_ROWS_GATHER=1:
    // the gathering stream kernel, kept for A/B); short ones: the stream kernel
    static const bool gather = [] { const char* e = std::getenv("SABER_HIP_GEMM_ROWS_GATHER"); return e && e[0] == '1'; }();
    if (k >= 2048 && !gather) {
        const dim3 grid((n + 15) / 16), block(256);
        if (m <= 8) hipLaunchKernelGGL((gemm_f32_rows_lds_kernel<8>), grid, block, 0, s, f);
        else hipLaunchKernelGGL((gemm_f32_rows_lds_kernel<16>), grid, block, 0, s, f);
        return hipGetLastError();
    }
    return launch_fc_f32_stream(f, false, s);
}

// a.M = batch rows (<= 16), a.C = reduction length (multiple of 16), a.K = outputs, a.Kg_pad = weight row pitch
bool fc_i8_small_ok(int m, int c, int kg_pad) { return m >= 1 && m <= 16 && c % 16 == 0 && c <= 4 * 16 * 64 && kg_pad >= ((c + 255) / 256) * 256; }

hipError_t launch_fc_i8_small(const ConvKArgs& a, hipStream_t s) {
    if (!fc_i8_small_ok(a.M, a.C, a.Kg_pad)) return hipErrorInvalidValue;
    const int ksw = (a.C + 255) / 256;     // k-steps per wave
    dim3 grid((a.K + 15) / 16), block(256);
#define SABER_FC_CASE(n) case n: hipLaunchKernelGGL((fc_i8_small_kernel<n>), grid, block, 0, s, a); break;
    switch (ksw) {
        SABER_FC_CASE(1) SABER_FC_CASE(2) SABER_FC_CASE(3) SABER_FC_CASE(4) SABER_FC_CASE(5) SABER_FC_CASE(6)
        SABER_FC_CASE(7) SABER_FC_CASE(8) SABER_FC_CASE(9) SABER_FC_CASE(10) SABER_FC_CASE(11) SABER_FC_CASE(12)
        SABER_FC_CASE(13) SABER_FC_CASE(14) SABER_FC_CASE(15) SABER_FC_CASE(16)
    default: return hipErrorInvalidValue;
    }
#undef SABER_FC_CASE
    return hipGetLastError();
}
// ... with the Softmax operator over the [M][K] logits in the same launch (K <= 1024: 16 logits per lane of the normalising wave)
bool fc_i8_small_softmax_ok(int m, int c, int kg_pad, int k) { return fc_i8_small_ok(m, c, kg_pad) && k >= 1 && k <= 1024; }
hipError_t launch_fc_i8_small_softmax(const ConvKArgs& a, float* prob, unsigned* ctr, hipStream_t s) {
    if (!fc_i8_small_softmax_ok(a.M, a.C, a.Kg_pad, a.K) || !prob || !ctr) return hipErrorInvalidValue;
    const int ksw = (a.C + 255) / 256;
    dim3 grid((a.K + 15) / 16), block(256);
    const FcSoftmaxTail t{prob, ctr};
    static const bool fenced = [] { const char* e = std::getenv("SABER_HIP_FC_SOFTMAX_FENCED"); return e && e[0] == '1'; }();
#define SABER_FC_CASE(n) case n: if (fenced) hipLaunchKernelGGL((fc_i8_small_softmax_kernel<n, true>), grid, block, 0, s, a, t); \
                                 else hipLaunchKernelGGL((fc_i8_small_softmax_kernel<n, false>), grid, block, 0, s, a, t); break;
    switch (ksw) {      // (ResNet's 2048- and VGG's 4096-long reductions and their neighbours; other lengths run the two launches)
        SABER_FC_CASE(2) SABER_FC_CASE(4) SABER_FC_CASE(8) SABER_FC_CASE(16)
    default: return hipErrorInvalidValue;
    }
#undef SABER_FC_CASE
    return hipGetLastError();
}

}  // namespace saber_mi355x
