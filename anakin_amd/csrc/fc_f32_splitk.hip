// anakin_amd/csrc/fc_f32_splitk.hip - the FP32 classifier tail at small batch: fc (VenderFc<X86,AK_FLOAT>, vender_fc.cpp:154-212: out = in W^T +
// bias) with the REDUCTION SPLIT OVER WORKGROUPS, and the Softmax operator that reads it (saber_softmax.cpp role) in the same launch
// (round 6; DESIGN 8 item 4 of round 5: "fc layers with few output tiles").
//
// Why: ResNet50's fc at batch <= 16 is an 8.2 MB weight stream read by 63 workgroups (one per 16 outputs: fc_small.hip) - 63 of 256 CUs,
// ~1 TB/s, 8.7 us at batch 1 and 10.8 at batch 8, followed by a 4.5 - 5.9 us softmax launch that reads 32 KB: 13 - 17 us of the FP32
// pass's tail for 4 MFLOP per image. Here the grid is (output tiles) x KS: workgroup (t, ks) reduces rows 16 t .. 16 t + 15 over slice ks of
// the reduction (its four waves a quarter each: every load of the launch is in flight at once, one memory round trip), stores its
// partial sums, and ARRIVES on the tile's counter; the last arriver of a tile adds the KS partials IN SPLIT ORDER (deterministic: the result
// does not depend on who arrives last), + bias (+ relu), writes the logits and arrives on the launch's counter; the last tile to finish
// normalises the rows when a Softmax follows. Hand-off: the form fc_small.hip's INT8 fc + softmax uses (MI355X_MICROARCH.md, valid forms) -
// write-through (agent-scope) stores, `s_waitcnt vmcnt(0)`, one returning agent-scope atomic, system-scope (L1- and L2-bypassing) loads on
// the reader; counters put back to zero by their last arriver; ONE launch of an fc object in flight at a time (saber_hip.h).
// MEASURED (profiles/r06/fc_tail.txt): NOT faster - fc + softmax 18.3 us in one launch against 10.8 + 5.9 = 16.7 for the two launches at batch 8,
// 12.6 against 8.6 + 4.5 at batch 1: the split replaces one launch boundary by two more round trips through the device's coherence point
// (partials out, partials back, logits out, logits back), and the tail is a chain of such latencies either way. Kept as an opt-in
// (SABER_HIP_FC_F32_SPLITK=1 when the fc's weights are set) with its parity test; the default stays the one-workgroup-per-tile kernel.
// Summation order differs from the one-workgroup kernels' and from MKL's: inside the 1e-4 FP32 contract like every FP32 kernel here.
#include "conv_igemm_impl.h"

#include <cstdlib>

namespace saber_mi355x {

struct FcSplitArgs {
    const float* w;      // [n rows (padded)][w_pitch]
    const float* x;      // [m][c]
    float* y;            // [m][n] logits
    float* prob;         // [m][n] softmax(y) per row, or null
    const float* bias;   // [n] or null
    float* part;         // [KS][16][n16] partial sums (n16 = 16 x tiles)
    unsigned* ctr;       // [tiles + 1]: per-tile arrivals, then the launch's; zero between launches
    const void* zero;
    int m, n, c, w_pitch, w_rows, relu;
    float neg_slope;
};

// STEPS: 16-float steps per wave (the wave's share of the slice = STEPS x 16 floats; slice = 4 waves x that)
template <int STEPS>
__global__ __launch_bounds__(256) void fc_f32_splitk_kernel(const FcSplitArgs a) {
    __shared__ v4f redf[3][64];
    __shared__ unsigned flag;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fq = lane >> 4;
    const int tiles = gridDim.x, KS = gridDim.y;
    const int t = blockIdx.x, ks = blockIdx.y;
    const int n0 = t * 16, n16 = tiles * 16;
    const int mrow = frow < a.m ? frow : a.m - 1;            // rows beyond the batch re-read the last one (results dropped)
    const int wr = n0 + frow < a.w_rows ? n0 + frow : a.w_rows - 1;
    const int k0 = (ks * 4 + wave) * (STEPS * 16) + fq * 4;  // this lane's first reduction index
    const float* const wrow = a.w + (size_t)wr * a.w_pitch;
    const float* const xrow = a.x + (size_t)mrow * a.c;
    v4i wf[STEPS], xf[STEPS];
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {                        // every load of the launch in flight before the first MFMA
        const int k = k0 + s * 16;
        const bool in = k < a.c;
        wf[s] = __builtin_nontemporal_load((const v4i*)(in ? wrow + k : wrow));
        xf[s] = *(const v4i*)(in ? xrow + k : (const float*)a.zero);
    }
    v4f acc[4] = {v4f{0.f, 0.f, 0.f, 0.f}, v4f{0.f, 0.f, 0.f, 0.f}, v4f{0.f, 0.f, 0.f, 0.f}, v4f{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
        const v4f wv = __builtin_bit_cast(v4f, wf[s]);
        const v4f xv = __builtin_bit_cast(v4f, xf[s]);
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.x, xv.x, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.y, xv.y, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.z, xv.z, acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv.w, xv.w, acc[3], 0, 0, 0);
    }
    v4f sum = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    if (wave > 0) redf[wave - 1][lane] = sum;
    __syncthreads();
    const int kb = n0 + fq * 4;                              // lane (wave 0): outputs kb .. kb + 3 of batch row frow
    if (wave == 0) {
#pragma unroll
        for (int w = 0; w < 3; ++w) sum += redf[w][lane];
        float* p = a.part + ((size_t)ks * 16 + frow) * n16 + kb;
#pragma unroll
        for (int r = 0; r < 4; ++r) __hip_atomic_store(p + r, sum[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // write-through
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the partials are at the device's coherence point
        if (lane == 0) flag = __hip_atomic_fetch_add(a.ctr + t, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)KS - 1 ? 1u : 0u;
    }
    __syncthreads();
    if (!flag) return;
    // ---- the tile's last arriver: the KS partials in split order, bias, activation -> logits
    unsigned last_tile = 0;
    if (wave == 0) {
        if (lane == 0) __hip_atomic_store(a.ctr + t, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // for the next launch
        // all KS x 4 partials requested before the first is added (a loop that loads and adds split by split pays one round trip to the
        // device's coherence point PER SPLIT: 8 x ~1.5 us), then added in split order
        float pv[16][4];
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const float* p = a.part + ((size_t)(s < KS ? s : KS - 1) * 16 + frow) * n16 + kb;
#pragma unroll
            for (int r = 0; r < 4; ++r) pv[s][r] = __hip_atomic_load(p + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        float tot[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 16; ++s)
#pragma unroll
            for (int r = 0; r < 4; ++r) tot[r] = s < KS ? __fadd_rn(tot[r], pv[s][r]) : tot[r];
        if (frow < a.m) {
            float* y = a.y + (size_t)frow * a.n + kb;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (kb + r >= a.n) continue;
                float d = tot[r];
                if (a.bias) d = __fadd_rn(d, a.bias[kb + r]);
                if (a.relu) d = d > 0.f ? d : (a.neg_slope == 0.f ? 0.f : __fmul_rn(d, a.neg_slope));
                __hip_atomic_store(y + r, d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (a.prob) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) last_tile = __hip_atomic_fetch_add(a.ctr + tiles, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)tiles - 1 ? 1u : 0u;
        }
    }
    if (!a.prob) return;
    __syncthreads();                                           // (every wave of this workgroup is here: `flag` was uniform)
    if (wave == 0 && lane == 0) flag = last_tile;
    __syncthreads();
    if (!flag) return;
    // ---- the launch's last tile: softmax over the rows (softmax_f32_kernel's arithmetic per row: max, exp(x - max), sum, divide; a row's
    // sum taken lane-major like fc_small.hip's INT8 tail), one wave per row, up to 16 logits per lane
    if (tid == 0) __hip_atomic_store(a.ctr + tiles, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    constexpr int PER = 16;
    for (int row = wave; row < a.m; row += 4) {
        const float* yr = a.y + (size_t)row * a.n;
        float v[PER];
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int c = lane + 64 * i;
            v[i] = __hip_atomic_load(yr + (c < a.n ? c : a.n - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        float mx = -3.4e38f;
#pragma unroll
        for (int i = 0; i < PER; ++i) mx = lane + 64 * i < a.n ? fmaxf(mx, v[i]) : mx;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        float sm = 0.f;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            v[i] = lane + 64 * i < a.n ? expf(v[i] - mx) : 0.f;
            sm += v[i];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sm += __shfl_xor(sm, o, 64);
        float* pr = a.prob + (size_t)row * a.n;
#pragma unroll
        for (int i = 0; i < PER; ++i)
            if (lane + 64 * i < a.n) pr[lane + 64 * i] = v[i] / sm;
    }
}

// split factor for a reduction of c floats: slices of 4 waves x STEPS x 16 floats with STEPS in {2, 4, 8}, at most 16 slices; 0 = not eligible
static int splitk_steps(int c, int* ks_out) {
    for (int steps : {4, 8, 2}) {
        const int slice = 64 * steps;
        if (c % slice == 0 && c / slice >= 2 && c / slice <= 16) { *ks_out = c / slice; return steps; }
    }
    return 0;
}
// m <= 16 rows, a reduction that divides into 2 .. 16 slices, FEW output tiles (<= 2048 outputs = 128 tiles: a layer with more fills the chip
// with one workgroup per tile already), <= 1024 outputs when a Softmax follows (16 logits per lane)
bool fc_f32_splitk_ok(int m, int c, int kg_pad, int k, bool softmax) {
    int ks = 0;
    return m >= 1 && m <= 16 && c % 4 == 0 && kg_pad >= c && splitk_steps(c, &ks) != 0 && k >= 1 && k <= 2048 && (!softmax || k <= 1024);
}
size_t fc_f32_splitk_part_floats(int c, int k) {
    int ks = 0;
    splitk_steps(c, &ks);
    return (size_t)ks * 16 * ((k + 15) / 16 * 16);
}
size_t fc_f32_splitk_counters(int k) { return (size_t)(k + 15) / 16 + 1; }
// a.w: the row-major repacked weights [K_pad][Kg_pad]; part / ctr: scratch of the sizes above (ctr zeroed once); prob may be null
hipError_t launch_fc_f32_splitk(const ConvKArgs& a, float* part, unsigned* ctr, float* prob, hipStream_t s) {
    if (!fc_f32_splitk_ok(a.M, a.C, a.Kg_pad, a.K, prob != nullptr) || !part || !ctr) return hipErrorInvalidValue;
    FcSplitArgs f;
    f.w = (const float*)a.w; f.x = (const float*)a.x; f.y = (float*)a.y; f.prob = prob; f.bias = a.bias; f.part = part; f.ctr = ctr; f.zero = a.zero;
    f.m = a.M; f.n = a.K; f.c = a.C; f.w_pitch = a.Kg_pad; f.w_rows = (a.K + 15) / 16 * 16; f.relu = a.relu; f.neg_slope = a.neg_slope;
    int ks = 0;
    const int steps = splitk_steps(a.C, &ks);
    const dim3 grid((a.K + 15) / 16, ks), block(256);
    if (steps == 2) hipLaunchKernelGGL((fc_f32_splitk_kernel<2>), grid, block, 0, s, f);
    else if (steps == 4) hipLaunchKernelGGL((fc_f32_splitk_kernel<4>), grid, block, 0, s, f);
    else hipLaunchKernelGGL((fc_f32_splitk_kernel<8>), grid, block, 0, s, f);
    return hipGetLastError();
}

}  // namespace saber_mi355x
