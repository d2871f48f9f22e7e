// anakin_amd/csrc/api_gemm.hip - saber_hip_gemm_f32: Gemm<MI355X, SABER_IMPL, float, float>::dispatch
// (saber/funcs/gemm.h:27-66; role of saber/funcs/impl/cuda/saber_gemm.cpp:6-29 -> the SASS sgemm, sass_funcs.h:632-697; numerics of
// the x86 MKL cblas_sgemm path within 1e-4): row-major C[m,n] = alpha * op(A)[m,k] * op(B)[k,n] + beta * C on raw device pointers.
//
// Round 4: the GEMM runs on the bf16 matrix cores through the FP32 implicit-GEMM machinery of conv_igemm_impl.h MODE 3 (an f32 value
// is exactly h + m + l with three bf16 terms; six products per f32 product accumulated in f32 reproduce it to ~2^-24) -
// gfx950's f32 MFMA peaks at 157 TFLOP/s, its dense bf16 MFMA at 2.5 PFLOP/s; round 3's kernel (gemm_f32_kernel, elementwise.hip: one
// 64 x 64 tile on v_mfma_f32_16x16x4_f32, 4-byte loads, two barriers per 16-deep step) reached 45 TFLOP/s = 28 % of the f32 peak and
// stays as the path for shapes the plane kernels do not take (k % 8 != 0, tiny problems). A few rows (m <= 16) against [n][k] weights - the
// shape of an fc - take neither: gemm_f32_rows_kernel below streams B exactly once.
//   C^T view: out-channels = n (MFMA rows), pixels = m (columns), reduction = k. The "convolution" is a 1 x 1 conv on an NHWC tensor
//   [1, m, 1, k] -> [1, m, 1, n]: x = op(A) as it lies when A is [m, k] (trans_a: one transposing pre-pass into the plan's scratch),
//   the weights W[n][k] = alpha * op(B)^T split on the DEVICE into the three bf16 planes the kernel streams (gemm_pack_planes_kernel: B is
//   a raw device pointer whose contents may change from call to call, so the split is redone per call: 10 bytes moved per element of B
//   against 2 m flops - 7 % of a 2048^3 GEMM, negligible for the tall problems of an fc);
//   beta: 0 -> plain store, 1 -> the kernel's in-place sum epilogue (out = acc + C), otherwise C is scaled by beta first (exactly:
//   beta/2 * C + beta/2 * C) and then summed in place.
// The entry point is stateless, the device work is not: conv object (tile choice, zero bias), plane and transpose scratch live in a
// small per-thread plan cache keyed by (device, stream, trans_a, trans_b, m, n, k, beta class) - the equivalent of Gemm<>::init's state
// (gemm.h:30-33). Plans are built outside stream capture on first use.
#include "api_internal.h"

namespace saber_mi355x {

// W[r][c] (r < n rows = out-channels, c < k) = alpha * (tb ? B[r * k + c] : B[c * n + r]) -> planes[p][r * kg_pad + c], p = 0..2
// 64 x 64 tiles through LDS so that both the read (along B's contiguous dimension) and the write (along k) are coalesced.
__global__ __launch_bounds__(256) void gemm_pack_planes_kernel(const float* __restrict__ B, int tb, int n, int k, float alpha,
                                                               unsigned short* __restrict__ planes, int kg_pad, size_t plane_elems) {
    __shared__ float t[64][65];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;      // 64 x 4
    if (tb) {
#pragma unroll 4
        for (int i = ty; i < 64; i += 4) {
            const int r = r0 + i, c = c0 + tx;
            t[i][tx] = (r < n && c < k) ? B[(size_t)r * k + c] : 0.f;
        }
    } else {
#pragma unroll 4
        for (int i = ty; i < 64; i += 4) {                       // i: column offset (k), tx: row offset (n) - B's contiguous dim
            const int c = c0 + i, r = r0 + tx;
            t[tx][i] = (r < n && c < k) ? B[(size_t)c * n + r] : 0.f;
        }
    }
    __syncthreads();
    auto rne = [](float x) -> unsigned short {
        unsigned u = __float_as_uint(x);
        u += 0x7fffu + ((u >> 16) & 1u);
        return (unsigned short)(u >> 16);
    };
    auto bf = [](unsigned short h) -> float { return __uint_as_float((unsigned)h << 16); };
#pragma unroll 4
    for (int i = ty; i < 64; i += 4) {
        const int r = r0 + i, c = c0 + tx;
        if (r < n && c < k) {
            const float w = alpha * t[i][tx];
            const unsigned short h = rne(w);
            const float r1 = w - bf(h);
            const unsigned short m = rne(r1);
            const float r2 = r1 - bf(m);
            const size_t o = (size_t)r * kg_pad + c;
            planes[o] = h;
            planes[plane_elems + o] = m;
            planes[2 * plane_elems + o] = rne(r2);
        }
    }
}

// Few rows (m <= 16) against weights stored [n][k] (trans_b, how an fc keeps them): the problem is ONE pass over B - VGG16's fc6 as a GEMM is
// a 411 MB stream for 1.6 GFLOP. A workgroup owns RN = 8 rows of B; its four waves split k in 1 KB steps (64 lanes x 16 bytes: coalesced,
// RN + MR 16-byte loads in flight per lane), every lane keeps RN x MR partial sums, which meet through DPP reductions and LDS. A ([m][k],
// <= 1.6 MB) is re-read by every workgroup from the L2. Summation order: per lane over its k positions, then lanes, then waves - fixed,
// so the result is deterministic; vs MKL's order inside the 1e-4 tolerance like every FP32 path.
template <int MR>
__global__ __launch_bounds__(256) void gemm_f32_rows_kernel(int m, int n, int k, float alpha, const float* __restrict__ A,
                                                            const float* __restrict__ B, float beta, float* __restrict__ C) {
    constexpr int RN = 8;
    __shared__ float red[4][RN * MR];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = blockIdx.x * RN;
    float acc[RN][MR];
#pragma unroll
    for (int r = 0; r < RN; ++r)
#pragma unroll
        for (int i = 0; i < MR; ++i) acc[r][i] = 0.f;
    const int k4 = k >> 2;                                   // k % 4 == 0 (checked by the caller)
    typedef float f4 __attribute__((ext_vector_type(4)));
    for (int c = wave * 64 + lane; c < k4; c += 256) {
        f4 bv[RN], av[MR];
#pragma unroll
        for (int r = 0; r < RN; ++r) {
            const int row = n0 + r < n ? n0 + r : n - 1;     // (rows beyond n repeat the last one, never stored)
            bv[r] = __builtin_nontemporal_load((const f4*)(B + (size_t)row * k) + c);      // streamed once: do not keep it in the caches
        }
#pragma unroll
        for (int i = 0; i < MR; ++i) av[i] = *((const f4*)(A + (size_t)(i < m ? i : m - 1) * k) + c);
#pragma unroll
        for (int r = 0; r < RN; ++r)
#pragma unroll
            for (int i = 0; i < MR; ++i) {
                acc[r][i] = __builtin_fmaf(bv[r].x, av[i].x, acc[r][i]);
                acc[r][i] = __builtin_fmaf(bv[r].y, av[i].y, acc[r][i]);
                acc[r][i] = __builtin_fmaf(bv[r].z, av[i].z, acc[r][i]);
                acc[r][i] = __builtin_fmaf(bv[r].w, av[i].w, acc[r][i]);
            }
    }
#pragma unroll
    for (int r = 0; r < RN; ++r)
#pragma unroll
        for (int i = 0; i < MR; ++i) {
            float v = acc[r][i];
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
            if (lane == 0) red[wave][r * MR + i] = v;
        }
    __syncthreads();
    if (tid < RN * MR) {
        const int r = tid / MR, i = tid % MR;
        const float v = ((red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid];
        if (n0 + r < n && i < m) {
            const size_t o = (size_t)i * n + n0 + r;
            const float av = alpha * v;
            C[o] = beta == 0.f ? av : av + beta * C[o];
        }
    }
}

}  // namespace saber_mi355x

namespace {

struct GemmPlan {
    int dev = -1;
    hipStream_t stream = nullptr;
    int ta = 0, tb = 0, m = 0, n = 0, k = 0, sum = 0;
    saber_hip_conv* op = nullptr;
    DevBuf<float> a_t;           // trans_a: A transposed to [m][k]
    unsigned long long stamp = 0;
    bool pinned = false;         // a hipGraph captured on `stream` holds this plan's buffers: never evicted, never matched by eager calls
    ~GemmPlan() {
        if (op) saber_hip_conv2d_destroy(op);
    }
};
// leaked at thread exit on purpose: the HIP runtime may already be gone when thread_local destructors run
thread_local std::vector<GemmPlan*>* g_plans = nullptr;
thread_local unsigned long long g_stamp = 0;
constexpr size_t kMaxPlans = 16;

// the plane kernels need 8-element k granularity and pay off once the problem fills the chip
bool plane_path(int m, int n, int k) {
    if (const char* e = std::getenv("SABER_HIP_GEMM_F32_PLANES")) return e[0] == '1' && k % 8 == 0;
    // (also for a few rows: VGG16's fc6 as a GEMM, m = 8, is a 411 MB weight stream - 618 us on the planes, pack included, against
    // 2 768 us on the 64 x 64 f32-MFMA tiles, whose 64 workgroups each walk 6 MB of B with 4-byte loads)
    return k % 8 == 0 && k >= 64 && (double)m * n * k >= 64.0 * 64 * 64 * 64;
}

int pick_tile(const saber_hip_conv* op, int m, int n) {
    // largest block tile (out-channels x pixels = n x m) that still gives every CU a workgroup; the 8-wave forms from 64 x 64 up
    struct T { int id, bn, bm; } cand[] = {{TILE_W8_256x128, 256, 128}, {TILE_W8_128x128, 128, 128}, {TILE_W8_128x64, 128, 64},
                                           {TILE_W8_64x64, 64, 64}, {TILE_64x32, 64, 32}, {TILE_32x32, 32, 32}};
    for (const T& t : cand) {
        if (!b3_tile_ok(op, t.id, 1)) continue;
        const long tiles = (long)((n + t.bn - 1) / t.bn) * ((m + t.bm - 1) / t.bm);
        if (tiles >= 256 || t.id == TILE_32x32) return t.id;
    }
    return TILE_32x32;
}

int build_plan(GemmPlan* p) {
    saber_hip_conv_desc d;
    std::memset(&d, 0, sizeof d);
    d.n = 1; d.h = p->m; d.w = 1; d.c = p->k; d.k = p->n; d.kh = d.kw = 1;
    d.stride_h = d.stride_w = d.dil_h = d.dil_w = 1;
    d.group = 1;
    d.in_dtype = d.out_dtype = SABER_HIP_F32;
    d.in_layout = d.out_layout = SABER_HIP_NHWC;
    d.act = SABER_HIP_ACT_NONE;
    d.res_mode = p->sum ? SABER_HIP_RES_SUM_INPLACE : SABER_HIP_RES_NONE;
    d.sum_scale = 1.f;
    int rc = saber_hip_conv2d_create(&d, &p->op);
    if (rc) return rc;
    saber_hip_conv* op = p->op;
    if (op->algo != ALGO_IGEMM_F32) return fail(SABER_HIP_UNIMPL, "gemm: shape not on the implicit-GEMM path");
    // the device-side halves of set_weights for an FP32 op whose weights arrive as device planes: zero bias, zeroed (padded) planes
    const int K_pad = round_up(p->n, 128);
    HIP_TRY(op->d_w3.alloc_zero((size_t)3 * K_pad * op->Kg_pad * 2));
    HIP_TRY(op->d_bias.alloc_zero(K_pad));
    op->has_bias = false;
    op->weights_set = true;
    op->b3 = 1; op->ks = 1; op->dma = 0; op->ksplit = 0;
    op->tile = pick_tile(op, p->m, p->n);
    name_algo(op);
    if (p->ta) HIP_TRY(p->a_t.alloc_zero((size_t)p->m * p->k));
    return SABER_HIP_OK;
}

// A plan's device buffers (weight planes, transpose scratch) are what a launch reads, so their lifetime is the caller-visible part of
// this "stateless" entry point (round-4 advisor finding):
//   * a call made while `s` is CAPTURING takes the plan an earlier eager call of the same key built and PINS it: the graph keeps the
//     pointers, so the plan is never evicted and no eager call uses it again (an eager call of that key builds a fresh one - a replay
//     and an eager launch would otherwise race on the plane buffer). With no such plan the call does not fail: *stateless = true and the
//     caller records the f32-MFMA kernel, which owns nothing (nothing may be allocated under capture);
//   * eviction (least recently used, unpinned only) drains the plan's stream first;
//   * saber_hip_gemm_f32_release_plans() frees the calling thread's plans (pinned ones too: the caller says its graphs are gone).
GemmPlan* find_plan(int dev, hipStream_t s, int ta, int tb, int m, int n, int k, int sum, int* rc, bool* stateless) {
    if (!g_plans) g_plans = new std::vector<GemmPlan*>();
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    const bool capturing = hipStreamIsCapturing(s, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone;
    GemmPlan* pinned_match = nullptr;
    GemmPlan* eager_match = nullptr;
    for (GemmPlan* p : *g_plans)
        if (p->dev == dev && p->stream == s && p->ta == ta && p->tb == tb && p->m == m && p->n == n && p->k == k && p->sum == sum) {
            if (p->pinned) pinned_match = pinned_match ? pinned_match : p;
            else eager_match = eager_match ? eager_match : p;
        }
    if (capturing) {
        // (round-5 advisor) a capture of a key that already HAS a graph-owned plan shares it - pinning the next unpinned match on every
        // capture grew the pinned set by one plan per capture / eager cycle, and pinned plans are never evicted
        if (pinned_match) return pinned_match;
        if (eager_match) {
            eager_match->pinned = true;
            eager_match->stamp = ++g_stamp;
            return eager_match;
        }
        *stateless = true;
        return nullptr;
    }
    if (eager_match) {
        eager_match->stamp = ++g_stamp;
        return eager_match;
    }
    size_t unpinned = 0;
    for (GemmPlan* p : *g_plans) unpinned += p->pinned ? 0 : 1;
    if (unpinned >= kMaxPlans) {      // evict the least recently used (its device buffers are idle: the stream is drained first)
        size_t lru = g_plans->size();
        for (size_t i = 0; i < g_plans->size(); ++i)
            if (!(*g_plans)[i]->pinned && (lru == g_plans->size() || (*g_plans)[i]->stamp < (*g_plans)[lru]->stamp)) lru = i;
        (void)hipStreamSynchronize((*g_plans)[lru]->stream);
        delete (*g_plans)[lru];
        g_plans->erase(g_plans->begin() + lru);
    }
    GemmPlan* p = new GemmPlan();
    p->dev = dev; p->stream = s; p->ta = ta; p->tb = tb; p->m = m; p->n = n; p->k = k; p->sum = sum;
    p->stamp = ++g_stamp;
    *rc = build_plan(p);
    if (*rc) {
        delete p;
        return nullptr;
    }
    g_plans->push_back(p);
    return p;
}

}  // namespace

int saber_hip_gemm_f32(int ta, int tb, int m, int n, int k, float alpha, const float* a, const float* b, float beta,
                       float* c, saber_hip_stream_t stream) {
    if (m <= 0 || n <= 0 || k <= 0) return fail(SABER_HIP_INVALID_VALUE, "bad gemm shape");
    if (!a || !b || !c) return fail(SABER_HIP_INVALID_VALUE, "null argument");
    if (g_capture) return capture_unsupported("saber_hip_gemm_f32");
    hipStream_t s = (hipStream_t)stream;
    ta = ta ? 1 : 0;
    tb = tb ? 1 : 0;
    if (m <= 16 && tb && !ta && k % 4 == 0 && (size_t)n * k >= ((size_t)1 << 20) && !std::getenv("SABER_HIP_GEMM_F32_PLANES")) {
        // a few rows against [n][k] weights: one pass over B on the f32 MFMA (fc_small.hip: fc_f32_stream_kernel, the kernel VenderFc's
        // FP32 role runs inside a net; SABER_HIP_GEMM_ROWS_VALU=1: round 4's VALU kernel, kept for A/B)
        const char* valu = std::getenv("SABER_HIP_GEMM_ROWS_VALU");
        if (!(valu && valu[0] == '1') && gemm_f32_rows_ok(m, k)) {
            HIP_TRY(launch_gemm_f32_rows(m, n, k, alpha, a, b, beta, c, zero_page(), s));
            return SABER_HIP_OK;
        }
        const dim3 grid((n + 7) / 8), block(256);
        if (m <= 4) hipLaunchKernelGGL(gemm_f32_rows_kernel<4>, grid, block, 0, s, m, n, k, alpha, a, b, beta, c);
        else if (m <= 8) hipLaunchKernelGGL(gemm_f32_rows_kernel<8>, grid, block, 0, s, m, n, k, alpha, a, b, beta, c);
        else hipLaunchKernelGGL(gemm_f32_rows_kernel<16>, grid, block, 0, s, m, n, k, alpha, a, b, beta, c);
        HIP_TRY(hipGetLastError());
        return SABER_HIP_OK;
    }
    if (!plane_path(m, n, k)) {
        HIP_TRY(launch_gemm_f32(ta, tb, m, n, k, alpha, a, b, beta, c, s));
        return SABER_HIP_OK;
    }
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    int rc = SABER_HIP_OK;
    bool stateless = false;
    GemmPlan* p = find_plan(dev, s, ta, tb, m, n, k, beta != 0.f ? 1 : 0, &rc, &stateless);
    if (stateless) {      // first sight of this shape while the stream is capturing: the kernel that owns no device state
        HIP_TRY(launch_gemm_f32(ta, tb, m, n, k, alpha, a, b, beta, c, s));
        return SABER_HIP_OK;
    }
    if (!p) return rc;
    saber_hip_conv* op = p->op;
    // W = alpha * op(B)^T as three bf16 planes [3][K_pad][Kg_pad] (the padding stays zero from the plan's allocation)
    {
        const int K_pad = round_up(n, 128);
        const dim3 grid((k + 63) / 64, (n + 63) / 64), block(256);
        hipLaunchKernelGGL(gemm_pack_planes_kernel, grid, block, 0, s, b, tb, n, k, alpha, (unsigned short*)op->d_w3.p, op->Kg_pad,
                           (size_t)K_pad * op->Kg_pad);
        HIP_TRY(hipGetLastError());
    }
    const float* x = a;
    if (ta) {      // A is stored [k][m]: NCHW [1, k, m, 1] -> NHWC [1, m, 1, k]
        HIP_TRY(launch_transpose_nchw_to_nhwc_f32(1, k, m, 1, k, a, p->a_t.p, s));
        x = p->a_t.p;
    }
    if (beta != 0.f && beta != 1.f) HIP_TRY(launch_eltwise_sum_f32((size_t)m * n, c, c, 0.5f * beta, 0.5f * beta, 0, c, s));
    return saber_hip_conv2d_run(op, x, c, nullptr, nullptr, s);
}

int saber_hip_gemm_f32_release_plans(void) {
    if (!g_plans) return 0;
    int n = 0;
    for (GemmPlan* p : *g_plans) {
        (void)hipStreamSynchronize(p->stream);
        delete p;
        ++n;
    }
    g_plans->clear();
    return n;
}
