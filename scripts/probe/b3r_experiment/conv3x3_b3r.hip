// anakin_amd/csrc/conv3x3_b3r.hip - FP32 3x3 / stride-1 / pad-1 convolution on SMALL feature maps (ResNet's res3 / res4 / res5: 28 x 28,
// 14 x 14, 7 x 7) on the bf16 matrix cores: the FP32 counterpart of conv3x3_img.h, in the three-plane scheme of conv_igemm_impl.h MODE 3.
//
// Why: at batch 8 these thirteen layers have 392 - 6 272 pixels and 1 152 - 4 608-deep reductions. The implicit-GEMM
// kernel runs them at 21 - 24 us (17 % of the bf16-plane roof): a 64 x 64 tile grid has 100 - 200 workgroups whose 32-deep stages each
// move both operands' three planes through LDS behind a barrier (0.56 us per stage for 0.08 us of MFMAs); the LDS-halo kernel
// (conv3x3_b3h.hip) has 16-wide tiles (14- and 7-wide images waste a quarter to half of them) and 64 workgroups at 14 x 14. The work,
// 677 K MFMAs, is 661 per wave if all 1 024 SIMDs get an equal share - which is exactly one 16-channel tile x ~6 pixel groups x 1 / 4 of
// the reduction. So:
//   workgroup = 16 output channels x a SLAB of R whole image rows (all W columns: the slab's R x W pixels are taken as ceil(R W / 16)
//   groups of 16 CONSECUTIVE pixels - no column waste; res4: 8 rows = 7 groups, res3: 4 rows = 7, res5: the 7 x 7 image = 4);
//   its NW waves split the REDUCTION by 32-channel chunk (wave w owns chunks w * CPW .. + CPW - 1): per chunk a wave requests its 27
//   weight fragments (9 taps x 3 planes, straight into MFMA A-operand registers from the fragment-ordered planes the LDS-halo kernel
//   already uses, saber_hip_conv::d_w3h1), loads the chunk's (R + 2) x (W + 2) halo, splits it into the three bf16 planes ONCE and keeps
//   it in a wave-PRIVATE LDS region (no barrier: a wave's LDS operations execute in order) - the B fragment of tap (dy, dx) for a pixel
//   is the same bytes at halo offset + dy (W + 2) + dx - and runs groups x 9 taps x 6 plane products;
//   the NW partial accumulators meet in LDS once, summed in wave order (deterministic), wave g mod NW finishes group g with
//   epilogue_f32 (bias, relu / leaky, in-place sum).
// 256 workgroups for each of the three stages at batch 8 (one per CU). Accumulation order: chunk-major within a wave, then the waves -
// differs from the other FP32 kernels', inside the 1e-4 FP32 tolerance like every FP32 path.
#include "conv_igemm_impl.h"

namespace saber_mi355x {

namespace {
__device__ __forceinline__ int b3r_swz(int px, int q) { return ((0x9C >> (2 * q)) & 3) ^ ((px >> 2) & 3); }
}  // namespace

struct B3rKArgs {
    ConvKArgs c;
    int R;            // image rows per slab
    int slabs_y;      // slabs per image = ceil(H / R)
    int G;            // pixel groups per slab = ceil(R * W / 16), <= GMAX
    int ktiles;       // channel tiles of 16 * TM = ceil(K / (16 TM))
    int cpw;          // 32-channel chunks per wave: C / 32 / NW
    int halo_px;      // (R + 2) * (W + 2)
    float inv_w, inv_hw;   // 1 / W, 1 / (W + 2): exact float-reciprocal division (fast_divmod) instead of ~40-instruction integer divisions
};

// NW: waves per workgroup (each one quarter / eighth of the reduction); TM: 16-channel tiles per workgroup (a B fragment read from LDS
// serves TM MFMA rows: with one tile the kernel is LDS-bound - 3 KB of fragment reads per six MFMAs is exactly the CU's 128 B / clk);
// GMAX: accumulator groups held in registers; SIT: halo items (pixel x 8-channel group) per lane and chunk, all requested before the first is used
template <int NW, int TM, int GMAX, int SIT>
__global__ __launch_bounds__(NW * 64) void conv3x3_b3r_kernel(const B3rKArgs k) {
    const ConvKArgs& a = k.c;
    extern __shared__ v4i lds_dyn[];                  // [NW][3 planes][halo_px * 4 chunks of 16 B]; reused for the cross-wave sum
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fq = lane >> 4;
    const int b = blockIdx.x;
    const int ktile = b % k.ktiles, slab = b / k.ktiles;      // consecutive workgroups: the channel tiles of one slab
    const int n = slab / k.slabs_y, y0 = (slab - n * k.slabs_y) * k.R;
    const int W = a.W, H = a.H, HW_ = W + 2;
    const int rows = (H - y0) < k.R ? (H - y0) : k.R;
    const int npx = rows * W;                         // valid pixels of this slab
    const int plane = k.halo_px * 4;                  // 16-byte chunks per plane
    v4i* const L = lds_dyn + (size_t)wave * 3 * plane;

    // this lane's pixel in every group: halo index of its tap (0, 0) and its output pixel
    int b_hp[GMAX], opix[GMAX];
#pragma unroll
    for (int g = 0; g < GMAX; ++g) {
        const int p = g * 16 + frow;
        const bool ok = g < k.G && p < npx;
        int r, c;
        fast_divmod(ok ? p : 0, W, k.inv_w, r, c);
        b_hp[g] = r * HW_ + c;
        opix[g] = ok ? (n * H + y0 + r) * W + c : -1;
    }
    v4f acc[TM][GMAX];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int g = 0; g < GMAX; ++g) acc[i][g] = v4f{0.f, 0.f, 0.f, 0.f};

    const int nchunks = a.C >> 5;
    const v4i* const wf = (const v4i*)a.w + (size_t)(ktile * TM) * nchunks * (9 * 3 * 64) + lane;
    const float* const xg = (const float*)a.x;
    const int items = k.halo_px * 4;                  // (halo pixel, 8-channel group) items per chunk
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};      // mma_step3's order: small terms first

    // this lane's halo items: source offset (in floats, without the chunk's channel offset; -1: outside the image = zeros)
    int s_off[SIT];
#pragma unroll
    for (int it = 0; it < SIT; ++it) {
        const int idx = lane + 64 * it;
        const int hp = idx >> 2, q = idx & 3;
        int hy, hx;
        fast_divmod(hp, HW_, k.inv_hw, hy, hx);
        const int iy = y0 - 1 + hy, ix = hx - 1;
        const bool ok = idx < items && iy >= 0 && iy < H && ix >= 0 && ix < W;
        s_off[it] = ok ? ((n * H + iy) * W + ix) * a.C + q * 8 : -1;
    }

    for (int ci = 0; ci < k.cpw; ++ci) {
        const int cc = wave * k.cpw + ci;
        // the chunk's 27 TM weight fragments: requested first, they arrive while the halo is staged
        v4i wr[TM][9][3];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) wr[i][t][pl] = wf[(size_t)i * nchunks * (9 * 3 * 64) + (size_t)(cc * 9 + t) * (3 * 64) + pl * 64];
        // the halo of this chunk, in two rounds: a round's items are ALL requested (unconditionally: outside the image from the zero page)
        // before the first is split into the three bf16 planes and stored to this wave's LDS region
        constexpr int SH = (SIT + 1) / 2;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            v4i xs[SH][2];
#pragma unroll
            for (int j = 0; j < SH; ++j) {
                const int it = half * SH + j;
                if (it >= SIT) continue;
                const v4i* p = s_off[it] >= 0 ? (const v4i*)(xg + s_off[it] + cc * 32) : (const v4i*)a.zero;
                xs[j][0] = p[0];
                xs[j][1] = p[1];
            }
#pragma unroll
            for (int j = 0; j < SH; ++j) {
                const int it = half * SH + j;
                if (it >= SIT) continue;
                const int idx = lane + 64 * it;
                if (idx >= items) continue;
                const v4f f0 = __builtin_bit_cast(v4f, xs[j][0]), f1 = __builtin_bit_cast(v4f, xs[j][1]);
                unsigned h[4], m[4], l[4];
                split3_pair(f0.x, f0.y, h[0], m[0], l[0]);
                split3_pair(f0.z, f0.w, h[1], m[1], l[1]);
                split3_pair(f1.x, f1.y, h[2], m[2], l[2]);
                split3_pair(f1.z, f1.w, h[3], m[3], l[3]);
                const int hp = idx >> 2;
                const int d = hp * 4 + b3r_swz(hp, idx & 3);
                L[d] = v4i{(int)h[0], (int)h[1], (int)h[2], (int)h[3]};
                L[plane + d] = v4i{(int)m[0], (int)m[1], (int)m[2], (int)m[3]};
                L[2 * plane + d] = v4i{(int)l[0], (int)l[1], (int)l[2], (int)l[3]};
            }
        }
        // groups two at a time, 9 taps x 6 plane products x TM tiles each (2 TM accumulators alternate in the MFMA stream). The B fragments
        // of the NEXT tap are read from LDS before the current tap's MFMAs are issued, and scheduling barriers keep it at exactly that: left
        // alone the scheduler hoists the reads of many taps (hundreds of registers, spills)
        auto read_b = [&](int g0, int t, v4i (&bf)[2][3]) {
            const int toff = (t / 3) * HW_ + (t % 3);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int hp = b_hp[g0 + j] + toff;
                const int di = hp * 4 + b3r_swz(hp, fq);
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) bf[j][pl] = L[pl * plane + di];
            }
        };
        auto mma_tap = [&](int g0, int t, const v4i (&bf)[2][3]) {
#pragma unroll
            for (int tt = 0; tt < 6; ++tt)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][g0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, wr[i][t][PA[tt]]),
                                                                                 __builtin_bit_cast(v8bf, bf[j][PB[tt]]), acc[i][g0 + j], 0, 0, 0);
        };
        v4i bf0[2][3], bf1[2][3];
        read_b(0, 0, bf0);
#pragma unroll
        for (int g0 = 0; g0 < GMAX; g0 += 2) {
            if (g0 >= k.G) break;
            constexpr int GN_LAST = GMAX - 2;
#pragma unroll
            for (int t = 0; t < 9; t += 2) {
                // tap t sits in bf0; request tap t + 1 (or, after tap 8, the next pair of groups' tap 0) into bf1
                if (t + 1 < 9) read_b(g0, t + 1, bf1);
                else read_b(g0 < GN_LAST ? g0 + 2 : g0, 0, bf1);
                __builtin_amdgcn_sched_barrier(0);
                mma_tap(g0, t, bf0);
                __builtin_amdgcn_sched_barrier(0);
                if (t + 1 < 9) {
                    if (t + 2 < 9) read_b(g0, t + 2, bf0);
                    __builtin_amdgcn_sched_barrier(0);
                    mma_tap(g0, t + 1, bf1);
                    __builtin_amdgcn_sched_barrier(0);
                } else {
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int pl = 0; pl < 3; ++pl) bf0[j][pl] = bf1[j][pl];      // (the next pair's tap 0 belongs in bf0)
                }
            }
        }
    }

    // ---- the NW partial sums meet in LDS: [wave][tile][group][lane] v4f; wave (g mod NW) sums group g in wave order and finishes it ----
    __syncthreads();                                  // every wave is done with its halo region
    v4f* const red = (v4f*)lds_dyn;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int g = 0; g < GMAX; ++g)
            if (g < k.G) red[(((size_t)wave * TM + i) * GMAX + g) * 64 + lane] = acc[i][g];
    __syncthreads();
    const int ohw = H * W;
    for (int g = wave; g < k.G; g += NW) {
        // (opix[] is indexed by a loop variable here: pick the entry without dynamic register indexing)
        int p = -1;
#pragma unroll
        for (int gg = 0; gg < GMAX; ++gg) p = gg == g ? opix[gg] : p;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int kb = (ktile * TM + i) * 16 + fq * 4;
            v4f s = red[(((size_t)0 * TM + i) * GMAX + g) * 64 + lane];
#pragma unroll
            for (int w = 1; w < NW; ++w) s += red[(((size_t)w * TM + i) * GMAX + g) * 64 + lane];
            if (p < 0 || kb >= a.K) continue;
            ChanParams<4> cp;
            load_chan_params<4>(a, kb, cp);
            const float v[4] = {s[0], s[1], s[2], s[3]};
            epilogue_f32<4>(a, v, cp, p, kb, n, p - n * ohw);
        }
    }
}

// Geometry of the launch for an op: rows per slab R, groups G, waves NW, tiles per workgroup TM, chunks per wave; false when the kernel does
// not take the shape (3x3 / stride 1 / pad 1 / dilation 1, NHWC f32 in and out, C in {128, 256, 512}, W <= 30, K % 16 == 0, no pair / pooling
// epilogue).
bool conv3x3_b3r_plan(int n, int h, int w, int c, int k, int* R, int* G, int* NW, int* TM, int* cpw, size_t* lds_bytes) {
    if (w > 30 || w < 4 || h < 1 || k % 16 || (c != 128 && c != 256 && c != 512)) return false;
    const int nw = c == 512 ? 8 : 4;                  // (C = 512: 16 chunks; eight waves = two per SIMD, 256 registers each: one tile)
    const int tm = (nw == 4 && k % 32 == 0) ? 2 : 1;
    const int gmax = nw == 8 ? 4 : 8, sit = nw == 8 ? 7 : 13;
    const int chunks = c / 32;
    if (chunks % nw) return false;
    // rows per slab: the NW private halo regions fit 150 KB of LDS, the groups fit the accumulator budget, the halo items the staging registers;
    // among those the slab height that executes the fewest 16-pixel groups per image (a slab's last group may be partly empty, a short last
    // slab runs all G groups), then the one that fills whole rounds of 256 workgroups best, then the taller one
    int best = 0;
    long best_groups = 0;
    double best_fill = 0.0;
    for (int r = h < 16 ? h : 16; r >= 1; --r) {
        const int g = (r * w + 15) / 16;
        const size_t bytes = (size_t)nw * 3 * (r + 2) * (w + 2) * 64;
        if (g > gmax || bytes > 150 * 1024 || (r + 2) * (w + 2) * 4 > sit * 64) continue;
        const int sy = (h + r - 1) / r;
        const long groups = (long)sy * g, wgs = (long)n * sy * (k / (16 * tm));
        const double fill = (double)wgs / (double)(((wgs + 255) / 256) * 256);
        if (!best || groups < best_groups || (groups == best_groups && fill > best_fill + 1e-9)) {
            best = r; best_groups = groups; best_fill = fill;
        }
    }
    if (!best) return false;
    *R = best; *G = (best * w + 15) / 16; *NW = nw; *TM = tm; *cpw = chunks / nw;
    const size_t halo = (size_t)nw * 3 * (best + 2) * (w + 2) * 64, red = (size_t)nw * tm * gmax * 64 * 16;
    *lds_bytes = halo > red ? halo : red;
    return true;
}

hipError_t launch_conv3x3_b3r(const ConvKArgs& a, hipStream_t s) {
    if (a.kh != 3 || a.kw != 3 || a.stride_h != 1 || a.stride_w != 1 || a.pad_h != 1 || a.pad_w != 1 || a.dil_h != 1 || a.dil_w != 1 ||
        a.out_nchw || a.K2 || a.pool_ow || (a.res_mode != RES_NONE && a.res_mode != RES_SUM_INPLACE))
        return hipErrorInvalidValue;
    B3rKArgs k;
    k.c = a;
    int nw = 0, tm = 0;
    size_t lds = 0;
    if (!conv3x3_b3r_plan(a.N, a.H, a.W, a.C, a.K, &k.R, &k.G, &nw, &tm, &k.cpw, &lds)) return hipErrorInvalidValue;
    k.slabs_y = (a.H + k.R - 1) / k.R;
    k.ktiles = (a.K + 16 * tm - 1) / (16 * tm);
    k.halo_px = (k.R + 2) * (a.W + 2);
    k.inv_w = 1.0f / (float)a.W;
    k.inv_hw = 1.0f / (float)(a.W + 2);
    const dim3 grid((unsigned)(a.N * k.slabs_y * k.ktiles));
    static bool attr_set = false;
    if (!attr_set) {      // (> 64 KB of dynamic LDS needs the attribute; idempotent, per process)
        (void)hipFuncSetAttribute((const void*)conv3x3_b3r_kernel<4, 2, 8, 13>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
        (void)hipFuncSetAttribute((const void*)conv3x3_b3r_kernel<4, 1, 8, 13>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
        (void)hipFuncSetAttribute((const void*)conv3x3_b3r_kernel<8, 1, 4, 7>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
        attr_set = true;
    }
    if (nw == 4 && tm == 2) hipLaunchKernelGGL((conv3x3_b3r_kernel<4, 2, 8, 13>), grid, dim3(256), lds, s, k);
    else if (nw == 4) hipLaunchKernelGGL((conv3x3_b3r_kernel<4, 1, 8, 13>), grid, dim3(256), lds, s, k);
    else hipLaunchKernelGGL((conv3x3_b3r_kernel<8, 1, 4, 7>), grid, dim3(512), lds, s, k);
    return hipGetLastError();
}

}  // namespace saber_mi355x
