// anakin_amd/csrc/api_internal.h - what the translation units behind include/saber_hip.h share: error reporting, device
// buffers, the operator structs behind the opaque handles, the autotuner's kernel-selection record and timing loop, and the
// op-list executor's structs. Nothing here is part of the ABI (the entry points get their C linkage from saber_hip.h).
//   api_conv.hip          convolution: create / set_tile / set_weights (quantise + repack) / run, sibling pairs
//   api_autotune.hip      per-op autotuner (cold-L2 timing loop, kernel-reuse preference)
//   api_ops.hip           fc, INT8 / FP32 GEMM, the streaming operators' wrappers
//   api_chain.hip         conv1x1 chains (two / three convs in one launch): stream repacking, create / run
//   api_net.hip           op-list executor: arena, lanes, hipGraph capture / replay, in-pass timing
//   api_net_optimize.hip  executor-level fusions (saber_hip_net_optimize)
//   api_net_autotune.hip  whole-net autotuner, selection save / restore
//   api_gemm.hip          FP32 GEMM on the bf16-plane kernels (device-side plane split, per-thread plan cache)
//   api_capture.hip       op-list capture: the *_run calls of a caller's own op loop recorded into a saber_hip_net
#pragma once
#include "../../include/saber_hip.h"
#include "kernels.h"

#include <cmath>
#include <cstdio>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <utility>
#include <type_traits>
#include <vector>

using namespace saber_mi355x;

namespace saber_api {

extern thread_local std::string g_err;      // defined in api_conv.hip; saber_hip_last_error reads it
inline int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
inline int hip_fail(hipError_t e, const char* where) {
    g_err = std::string(where) + ": " + hipGetErrorString(e);
    return SABER_HIP_RUNTIME_ERROR;
}
#define HIP_TRY(expr)                                    \
    do {                                                 \
        hipError_t _e = (expr);                          \
        if (_e != hipSuccess) return hip_fail(_e, #expr); \
    } while (0)

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }
inline int conv_out(int in, int pad, int k, int dil, int stride) {
    return (in + 2 * pad - (dil * (k - 1) + 1)) / stride + 1;  // funcs_utils.h:29-53
}

template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    ~DevBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
    hipError_t upload(const std::vector<T>& h) {
        release();
        if (h.empty()) return hipSuccess;
        hipError_t e = hipMalloc((void**)&p, h.size() * sizeof(T));
        if (e != hipSuccess) return e;
        n = h.size();
        return hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
    }
    hipError_t alloc_zero(size_t count) {
        release();
        hipError_t e = hipMalloc((void**)&p, count * sizeof(T));
        if (e != hipSuccess) return e;
        n = count;
        e = hipMemset(p, 0, count * sizeof(T));
        return e != hipSuccess ? e : hipStreamSynchronize(nullptr);
    }
};

// 256 zero bytes in device memory per device, shared by every op created on it (padded taps of the LDS-DMA kernels).
// One page per device (an op on a second GPU must not DMA from the first one's memory), created under a lock, and
// the memset is complete before any kernel on a non-blocking stream can read it.
void* zero_page();      // api_conv.hip

enum Algo { ALGO_IGEMM_I8 = 0, ALGO_IGEMM_I8_C4 = 1, ALGO_IGEMM_F32 = 2, ALGO_DIRECT_I8 = 3, ALGO_DIRECT_F32 = 4 };

}  // namespace saber_api
using namespace saber_api;

struct saber_hip_conv {
    saber_hip_conv_desc d;
    int oh = 0, ow = 0;
    int algo = ALGO_DIRECT_I8;
    int tile = TILE_64x64;
    int ks = 1;              // 64-byte k-steps per pipeline stage (1, 2, 4)
    int dma = 0;             // 0: register-staged kernel; 1/2/4: LDS-DMA ring kernel with that many wave groups
    int stem = 0;            // 1: LDS-patch stem kernel (conv_stem.h) instead of the NHWC4 implicit GEMM
    int pool_fused = 0, pool_oh = 0, pool_ow = 0;   // SaberConv2DPooling: fused stem conv + 3x3/2 max pooling
    int pool2 = 0;           // SaberConv2DPooling, FP32: relu'd implicit-GEMM conv + 2x2/2 max pooling in the epilogue
    int stem32 = 0;          // SaberConv2DPooling, FP32 stem (with pool_fused): NCHW image -> 7x7/2 conv + relu + 3x3/2 max pooling in ONE launch
                             // (conv_stem_f32.hip); d_wstem32 = its weight planes in MFMA fragment order, packed by set_pooling from w_stem_host
    DevBuf<uint8_t> d_wstem32;
    std::vector<float> w_stem_host;   // the OIHW f32 weights of a 3 -> 64 7x7 FP32 conv as handed to set_weights (9.4 K floats)
    int halo = 0;            // 4 / 8: LDS-halo 3x3 kernel with that many tile rows (conv3x3_halo.h); 0: not used
    int fc_small = 0;        // 1: small-batch fc kernel (fc_small.hip) instead of the implicit-GEMM conv kernel
    int b3h = 0;             // FP32 3x3: 1..5 = LDS-halo bf16-plane kernel variant (conv3x3_b3h.hip), 0: not used
    DevBuf<uint8_t> d_w3h1, d_w3h2;   // its weight planes in MFMA fragment order for 1 / 2 row tiles per wave
    int pw = 0;              // FP32 1x1 / stride 1: 1 = persistent register-weights kernel (C = 64 / 128, conv1x1_pw.hip), 2 .. 5 = the
                             // reduction-split kernel's variants 1 .. 4 (C = 128 .. 2048, conv1x1_pwk.hip), 0: not used
    DevBuf<uint8_t> d_wpw;   // its weight planes in that kernel's fragment order
    DevBuf<float> d_fcpart;  // FP32 fc at <= 16 rows and <= 2048 outputs: the split-K kernel's partial sums + arrival counters (fc_f32_splitk.hip;
    DevBuf<unsigned> d_fcctr; // allocated by set_weights when the shape is eligible AND SABER_HIP_FC_F32_SPLITK=1: opt-in, measured no faster - profiles/r06/fc_tail.txt)
    DevBuf<float> d_wfc;     // FP32 fc at <= 16 rows: the weights fragment-major for the streaming kernel (fc_small.hip: fc_f32_stream_kernel PACKED)
    int img1 = 0;            // INT8: 1 = image-resident kernel (stage_xcd.hip: img_conv_kernel): workgroup = one image x 16 NT channels
    int gpool = 0;           // ... with the global average pooling of its output fused (saber_hip_net_optimize flag 128): img1 only
    struct saber_hip_stage* img_stage = nullptr;   // the single-phase descriptor + repacked weights of that kernel (img_conv_prepare)
    int ksplit = 0;          // b3 only: log2 of the split-K factor (conv_igemm_impl.h: splits of one tile share an XCD), 0: none
    bool no_placement = false;   // the op belongs to a net that does NOT own the device (saber_hip_net_optimize flag 2048): no split-K through one XCD's L2
    int b3 = 0;              // FP32: 1 = the implicit GEMM runs on the bf16 matrix cores (three bf16 operand planes, conv_igemm_impl.h
                             // MODE 3): needs c_eff % 8 == 0 and the pre-split weight planes d_w3
    int img_ib = 0, img_rb = 0, img_nw = 4;   // img_rb > 0: small-image 3x3 kernel (conv3x3_img.h): images / output rows
                                              // per workgroup slab, waves per workgroup (4 or 8)
    int epi = EPI_I8_CONV;
    bool is_i8 = false;
    int x_dtype = DT_S8;     // dtype of the tensor the conv kernel itself reads
    int c_eff = 0;           // channel count the conv kernel sees (after padding)
    bool pre_quant = false;  // f32 NCHW input quantised into the workspace first
    bool pre_pad = false;    // 8-bit C<4 input padded to NHWC4 into the workspace
    bool pre_transpose = false;  // f32 NCHW input transposed to NHWC(c_eff) into the workspace
    size_t ws_bytes = 0;
    int Kg = 0, Kg_pad = 0, kw_pad = 0;
    float in_scale = 1.f, out_scale = 1.f;
    bool weights_set = false;
    std::vector<int8_t> wq_oihw;
    std::vector<float> w_scale;
    std::vector<float> bias_host;   // the op's f32 bias as handed to set_weights (saber_hip_net_optimize re-creates ops from it)
    std::vector<float> bias_p_host, scale_host;   // INT8: the device-side bias' / scale / comp arrays (conv1x1 chain repacks them)
    std::vector<int> comp_host;
    DevBuf<uint8_t> d_w;
    DevBuf<float> d_part;    // split-K: partial accumulators [tile][split] and the tiles' arrival counters (split_prepare)
    DevBuf<unsigned> d_part_ctr;
    unsigned* h_part_err = nullptr;   // pinned, device-mapped word the split-K kernels count placement violations in (split_prepare)
    DevBuf<uint8_t> d_w3;    // FP32 convs: the repacked weights split into three bf16 planes [3][K_pad][Kg_pad] (b3 variant)
    DevBuf<float> d_bias, d_scale;
    DevBuf<int> d_comp;
    DevBuf<unsigned> d_sm_ctr;   // INT8 fc + softmax in one launch (fc_small.hip): the arrival counter, zero between launches
    bool has_bias = false, has_comp = false;
    std::string algo_name;
    // sibling pair (saber_hip_conv2d_create_pair): d.k = k1 + k2, rows >= k1 belong to the second conv
    int pair_k1 = 0, pair_k2 = 0, pair_relu2 = 0, pair_dtype2 = 0;
    const saber_hip_conv* pair_src_a = nullptr;   // the two ops the pair was made of (not owned; saber_hip_net_optimize flag 512 packs
    const saber_hip_conv* pair_src_b = nullptr;   // their weights again for the stem kernel's tail)
};

// the fused stem conv + max pooling with the sibling pair of 1x1 convs reading the pooled tensor in the same launch
// (conv_stem.h: conv_stem_pool_pair_kernel); refers to the three ops, owns the pair's weights in fragment order and its constants
struct saber_hip_stem_pair {
    saber_hip_conv* stem = nullptr;
    const saber_hip_conv* a = nullptr;
    const saber_hip_conv* b = nullptr;
    DevBuf<uint8_t> d_w, d_prm;
};

// two 1x1 INT8 convs in one launch (conv1x1_chain.hip); refers to the two ops, owns the repacked stream
struct saber_hip_chain {
    saber_hip_conv* c3 = nullptr;   // the block's 3x3 conv in front of `a` (saber_hip_conv2d_chain_create3), or null
    saber_hip_conv* a = nullptr;
    saber_hip_conv* b = nullptr;
    saber_hip_conv* b2 = nullptr;   // strided head + sibling pair (saber_hip_conv2d_chain_create3_pair): b and b2 both read a's output
    int c1 = 0, k1 = 0, k2 = 0, tn = 0;
    DevBuf<uint8_t> d_stream, d_prm0, d_prm1, d_prm2;
    DevBuf<uint8_t> d_stream_split;   // 1x1 chains with C >= 256: [half][wave] streams for the split second conv (tile | 8)
    DevBuf<uint8_t> d_stream_split8;  // C == 256: the same for 8 waves per workgroup (tile 11)
    DevBuf<uint8_t> d_stream_w8;      // C == 128: the whole stream for 8 waves per workgroup (tile | 4)
    // C == 256, 3x3-led, tile 7: two cooperating workgroups per pixel tile (conv_chain_coop.hip): [half][wave] streams, the pairs'
    // arrival counters, the exchange buffer of the 3x3 conv's tile, the halves' XCC ids, and the pinned error word
    // tile 15: FOUR cooperating workgroups per tile of 2 rows x 16 columns (conv_stage_coop.hip with one block): [quarter][wave]
    // streams here, everything else in `stage1`
    DevBuf<uint8_t> d_stream_coop, d_stream_coop4, d_coop_xch;
    DevBuf<uint8_t> d_stream_stage1;  // C == 128, 3x3-led with a second 1x1 conv: per-wave streams of the one-workgroup-per-tile stage kernel
    DevBuf<unsigned long long> d_coop_ctr;
    DevBuf<unsigned> d_coop_xcc;
    unsigned* h_coop_err = nullptr;
    int coop_tiles = 0;
    struct saber_hip_chain_stage* stage1 = nullptr;
    ~saber_hip_chain();      // api_chain.hip
};

// A run of 3x3-led C = 256 chains as ONE persistent launch (conv_stage_coop.hip): the chains' own weight streams and constants, the
// hand-off counters / exchange tiles / XCC words of the launch, a pinned error word. The chains are NOT owned.
struct saber_hip_chain_stage {
    std::vector<saber_hip_chain*> chains;
    DevBuf<saber_mi355x::StageBlk> d_blk;
    DevBuf<unsigned long long> d_grp_ctr, d_img_ctr;
    DevBuf<uint8_t> d_xch;
    DevBuf<unsigned> d_xcc;
    unsigned* h_err = nullptr;
    int c1 = 0, n = 0, h = 0, w = 0, tiles_x = 0, tiles_per_img = 0;
    bool per_image = false;
    ~saber_hip_chain_stage() {
        if (h_err) (void)hipHostFree(h_err);
    }
};

struct saber_hip_fc {
    saber_hip_fc_desc d;
    saber_hip_conv* conv = nullptr;
    bool pre_quant = false;
    float in_scale = 1.f;
};

namespace saber_api {
// one selection of kernel variant for an op (what the autotuner saves / restores)
struct ConvChoice {
    int tile, ks, dma, stem, halo, img_ib, img_rb, img_nw, fc_small, b3, ksplit, img1, b3h, pw;
};
inline ConvChoice get_choice(const saber_hip_conv* op) {
    return {op->tile, op->ks, op->dma, op->stem, op->halo, op->img_ib, op->img_rb, op->img_nw, op->fc_small, op->b3, op->ksplit, op->img1, op->b3h, op->pw};
}
inline void set_choice(saber_hip_conv* op, const ConvChoice& c) {
    op->tile = c.tile; op->ks = c.ks; op->dma = c.dma; op->stem = c.stem; op->halo = c.halo;
    op->img_ib = c.img_ib; op->img_rb = c.img_rb; op->img_nw = c.img_nw; op->fc_small = c.fc_small; op->b3 = c.b3; op->ksplit = c.ksplit; op->img1 = c.img1 || op->gpool; op->b3h = c.b3h; op->pw = c.pw;
}
inline bool pw_ok(const saber_hip_conv* op) {      // the persistent pointwise kernel exists for this op (planes packed by pw_prepare)
    return op->algo == ALGO_IGEMM_F32 && op->d_wpw.p != nullptr && !op->pair_k2 && !op->pool2 && conv1x1_pw_ok(op->c_eff, op->d.k);
}
inline bool pwk_ok(const saber_hip_conv* op, int variant) {      // ... the reduction-split pointwise kernel, variant 1 .. 4
    int tm, p, dd, mb;
    return conv1x1_pwk_variant(variant, &tm, &p, &dd, &mb) && op->algo == ALGO_IGEMM_F32 && op->d_wpw.p != nullptr && !op->pair_k2 &&
           !op->pool2 && conv1x1_pwk_ok(op->d.n * op->d.h * op->d.w, op->c_eff, op->d.k) && (op->c_eff >> 7) >= dd;      // (slabs in flight <= slabs per wave)
}
inline bool b3_ok(const saber_hip_conv* op) {   // the bf16-plane variant exists for this op (planes uploaded by set_weights)
    return op->algo == ALGO_IGEMM_F32 && op->d_w3.p != nullptr;
}
// (tile, stage depth) combinations of the bf16-plane kernels: depth 2 below 128 x 128; the 256-row tile reads 256 weight rows per
// workgroup without a row predicate, the planes are padded to multiples of 128 rows
inline bool b3_tile_ok(const saber_hip_conv* op, int tile, int ks) {
    if (!b3_ok(op) || tile < 0 || tile >= TILE_COUNT_B3 || (ks != 1 && ks != 2)) return false;
    if (ks == 2 && (tile == TILE_128x128 || tile >= TILE_W8_128x128)) return false;
    if (tile == TILE_W8_256x128 && ((op->d.k + 127) / 128) % 2 != 0) return false;
    return true;
}
inline bool b3h_ok(const saber_hip_conv* op, int variant) {      // the halo variant exists for this op (planes packed by set_weights)
    int bmk, th, tm, thr;
    if (!conv3x3_b3h_variant(variant, &bmk, &th, &tm, &thr)) return false;
    if ((variant >= 6) != (op->d.kh == 1)) return false;       // 1..5: the 3x3 forms, 6..8: pointwise
    if (variant >= 6 && op->pool2) return false;
    return op->algo == ALGO_IGEMM_F32 && (tm == 1 ? op->d_w3h1.p : op->d_w3h2.p) != nullptr && !op->pair_k2;
}
inline bool fc_small_ok(const saber_hip_conv* op) {
    if (op->algo == ALGO_IGEMM_F32)   // FP32 fc: a 1x1 "conv" on a [m, 1, 1, k] NHWC tensor, plain f32 epilogue, no residual
        return op->epi == EPI_F32 && op->d.h == 1 && op->d.w == 1 && op->d.kh == 1 && op->d.kw == 1 && !op->pre_transpose &&
               op->d.out_layout == SABER_HIP_NHWC && op->d.res_mode == SABER_HIP_RES_NONE && !op->pair_k2 && !op->pool2 &&
               fc_f32_small_ok(op->d.n, op->c_eff, op->Kg_pad);
    return op->algo == ALGO_IGEMM_I8 && (op->epi == EPI_I8_FC_S8 || op->epi == EPI_I8_FC_U8) && op->d.h == 1 && op->d.w == 1 &&
           fc_i8_small_ok(op->d.n, op->c_eff, op->Kg_pad);
}
struct EventPair {   // RAII: destroyed on every exit path
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t init() {
        hipError_t e = hipEventCreate(&e0);
        return e != hipSuccess ? e : hipEventCreate(&e1);
    }
    ~EventPair() {
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
    }
};
// Times launches the way they run inside an op list: between repetitions a 64 MB stream through every XCD pushes the
// operands out of the L2s (the Infinity Cache keeps them), so a kernel that re-reads many weights per workgroup is not
// flattered by finding them in L2 the way a back-to-back loop of the same launch does (measured: conv3x3 + chain at
// C = 256 reads 12.9 us back to back, 16 us in the forward pass; the two launches it replaces 13.9 -> 15.3 us).
struct ColdBench {
    // 64 MB pushes the operands out of the 8 x 4 MB of L2 (the 256 MB Infinity Cache keeps them: right for a net whose whole forward
    // pass fits it, ResNet50 INT8 at batch 8 moves ~190 MB). A net whose pass moves more (VGG16 FP32 at batch 8: > 400 MB of
    // activations) finds its weights in neither cache: saber_hip_net_autotune then asks for a flush of that size (up to 512 MB) -
    // measured: with the 64 MB flush the tuner preferred the LDS-halo FP32 kernel for VGG16's 512-channel layers (90 / 169 us "cold")
    // which then ran at 120 / 227 us in the pass, where every workgroup streams its 1.7 MB of weight planes from HBM.
    size_t kBytes = (size_t)64 << 20;
    void* buf = nullptr;
    std::vector<hipEvent_t> ev;
    int reps = 0;
    hipError_t init(int r, size_t bytes = 0) {
        if (bytes) kBytes = std::min<size_t>(std::max<size_t>(bytes, (size_t)64 << 20), (size_t)512 << 20) & ~(size_t)0xfffff;
        reps = r < 3 ? 3 : (r > 32 ? 32 : r);
        hipError_t e = hipMalloc(&buf, kBytes + 256);
        if (e != hipSuccess) return e;
        e = hipMemset(buf, 1, kBytes + 256);
        if (e != hipSuccess) return e;
        ev.assign(2 * (size_t)reps, nullptr);
        for (hipEvent_t& x : ev)
            if ((e = hipEventCreate(&x)) != hipSuccess) return e;
        return hipStreamSynchronize(nullptr);
    }
    ~ColdBench() {
        for (hipEvent_t x : ev)
            if (x) (void)hipEventDestroy(x);
        if (buf) (void)hipFree(buf);
    }
    // median microseconds of fn() (which enqueues on s and returns a status); < 0 on failure
    float run(hipStream_t s, const std::function<int()>& fn) {
        if (fn() != 0) return -1.f;                  // warm-up: code, kernel arguments
        for (int r = 0; r < reps; ++r) {
            if (launch_l2_flush(buf, kBytes, (unsigned*)((char*)buf + kBytes), s) != hipSuccess) return -1.f;
            if (hipEventRecord(ev[2 * r], s) != hipSuccess) return -1.f;
            if (fn() != 0) return -1.f;
            if (hipEventRecord(ev[2 * r + 1], s) != hipSuccess) return -1.f;
        }
        if (hipEventSynchronize(ev[2 * reps - 1]) != hipSuccess) return -1.f;
        std::vector<float> t(reps);
        for (int r = 0; r < reps; ++r)
            if (hipEventElapsedTime(&t[r], ev[2 * r], ev[2 * r + 1]) != hipSuccess) return -1.f;
        std::sort(t.begin(), t.end());
        return t[reps / 2] * 1000.f;
    }
};

// One ColdBench per top-level autotune call: nested calls (saber_hip_net_autotune -> saber_hip_conv2d_autotune) share it.
// SABER_HIP_AUTOTUNE_WARM=1 in the environment restores the back-to-back timing loop (kept for A/B measurements).
extern thread_local ColdBench* g_cold;      // api_autotune.hip
// Kernel reuse across the ops of one net: the FIRST launch of a given kernel function in a forward pass pays for its cold
// code (0.3-0.6 us on most boxes of the pool, 3-9 us on some: profiles/r02/slow_box/ - repeats of the same function
// later in the pass run at full speed), so among candidates within g_reuse_tol of the fastest the tuner prefers a
// function another op of the net already uses. Set by saber_hip_net_autotune for the duration of its run.
extern thread_local std::vector<unsigned long long>* g_used_kernels;      // api_autotune.hip
constexpr float g_reuse_tol = 0.03f;
inline unsigned long long kernel_key(const saber_hip_conv* op, const ConvChoice& c) {
    const saber_hip_conv_desc& d = op->d;
    int ek = 3;   // conv_igemm.hip: epilogue_kind
    if (op->pair_k2) ek = 4;
    else if (op->algo != ALGO_IGEMM_F32 && op->epi == EPI_I8_CONV && d.res_mode != SABER_HIP_RES_SUM_INPLACE && d.k % 16 == 0)
        ek = d.res_mode == SABER_HIP_RES_ELTWISE ? 2 : (d.out_dtype == SABER_HIP_U8 ? 1 : (d.out_dtype == SABER_HIP_S8 ? 0 : 3));
    unsigned long long k = (unsigned long long)op->algo | ((unsigned long long)ek << 4);
    if (c.pw > 1) return k | (10ull << 8) | ((unsigned long long)c.pw << 16) | ((unsigned long long)(d.res_mode == SABER_HIP_RES_SUM_INPLACE) << 32);
    if (c.pw) return k | (9ull << 8) | ((unsigned long long)op->c_eff << 16) | ((unsigned long long)(d.res_mode == SABER_HIP_RES_SUM_INPLACE) << 32);
    if (c.b3h) return k | (8ull << 8) | ((unsigned long long)c.b3h << 16);
    if (c.img1) return k | (7ull << 8) | ((unsigned long long)(op->d.kh == 3) << 16);      // one function for all image-resident shapes
    if (c.fc_small) return k | (1ull << 8) | ((unsigned long long)((op->c_eff + 255) / 256) << 16);
    if (c.stem) return k | (2ull << 8);
    if (c.img_rb)   // <EK, NW, CW, NCH, GPW>: channel count and pixel groups per wave
        return k | (3ull << 8) | ((unsigned long long)op->c_eff << 16) |
               ((unsigned long long)((c.img_ib * c.img_rb * op->ow + 15) / 16) << 32);
    if (c.halo) return k | (4ull << 8) | ((unsigned long long)c.halo << 16) | ((unsigned long long)(op->c_eff % 128 == 0) << 24);
    return k | ((c.b3 ? 6ull : 5ull) << 8) | ((unsigned long long)c.tile << 16) | ((unsigned long long)c.ks << 24) | ((unsigned long long)c.dma << 32);
}
struct ColdScope {
    ColdBench local;
    bool owner = false;
    hipError_t enter(int reps, size_t flush_bytes = 0) {
        const char* w = std::getenv("SABER_HIP_AUTOTUNE_WARM");
        if (w && w[0] == '1') return hipSuccess;      // g_cold stays null: callers fall back to the warm loop
        if (g_cold) return hipSuccess;
        hipError_t e = local.init(reps, flush_bytes);
        if (e != hipSuccess) return e;
        g_cold = &local;
        owner = true;
        return hipSuccess;
    }
    ~ColdScope() {
        if (owner) g_cold = nullptr;
    }
};

}  // namespace saber_api

// op-list executor
namespace saber_api {
enum OpKind { OP_CONV, OP_CONV_PAIR, OP_FC, OP_QUANT, OP_DEQUANT, OP_TRANSPOSE_IN, OP_ELT_I8, OP_ELT_F32, OP_POOL_I8, OP_POOL_F32, OP_POOL_F32_I8, OP_FC_Q, OP_SOFTMAX, OP_RELU_F32, OP_ACT_F32 };
struct NetOp {
    OpKind kind;
    std::string name;
    saber_hip_conv* conv = nullptr;
    saber_hip_fc* fc = nullptr;
    int in = -1, in2 = -1, out = -1, out2 = -1;
    // conv1x1 chain (saber_hip_net_optimize flag 16): this conv and the NEXT op (a 1x1 conv reading its output) run as one
    // launch while use_chain is set; the next op carries `skip` and launches nothing
    saber_hip_chain* chain = nullptr;
    int chain_out = -1;
    bool use_chain = false, skip = false;
    // ... with the block's 3x3 conv in front (flag 32): THIS op is that 3x3 conv, the next two are the chain; while use_chain3
    // is set it launches all three (its own output edge is then not written) and both followers carry `skip`
    saber_hip_chain* chain3 = nullptr;
    int chain3_res = -1, chain3_y1 = -1, chain3_y2 = -1;
    int chain3_y3 = -1;      // strided head + sibling pair (flag 1024): the pair's second output; the pair op (ops[i + 2]) carries `skip`
    bool use_chain3 = false;
    // ... and a RUN of such 3x3-led C = 256 chains (flag 256): THIS op is the first block's 3x3 conv; while use_stage is set it launches
    // all stage_n chains (3 * stage_n ops, the others carry `skip`) as one persistent launch (saber_hip_conv2d_stage_run)
    saber_hip_chain_stage* stage = nullptr;
    int stage_n = 0;
    bool use_stage = false;
    // the fused stem conv + pooling with the sibling pair that reads the pooled tensor (flag 512): THIS op is the stem conv, the next
    // op (the pair, `skip`) launches nothing; stem_y1 / stem_y2 are the pair's outputs and this op's own output edge is not written
    saber_hip_stem_pair* stem_pair = nullptr;
    int stem_y1 = -1, stem_y2 = -1;
    int lane = 0;            // 0: caller's stream, 1: the net's side stream (graph::Lane, operator_func.h:103-114)
    bool record = false;     // an op on the other lane consumes this op's output: record an event after it
    int p[16] = {0};
    float f[6] = {0};
    size_t count = 0;
};
}  // namespace saber_api

struct saber_hip_net {
    std::vector<size_t> tensor_bytes;
    std::vector<size_t> tensor_off;
    std::vector<void*> tensor_ext;     // per tensor: caller-owned storage (saber_hip_net_bind_tensor) or null = a slot of the arena
    std::vector<std::pair<const void*, int>> captured_ptr;   // captured nets: caller pointer -> id of the LAST tensor seen there
    void* ptr(int id) const {
        if (id < 0) return nullptr;
        if ((size_t)id < tensor_ext.size() && tensor_ext[id]) return tensor_ext[id];
        return (void*)(arena + tensor_off[id]);
    }
    std::vector<NetOp> ops;
    char* arena = nullptr;
    size_t arena_bytes = 0, ws_off = 0, ws_bytes = 0;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    bool finalized = false;
    bool compacted = false;      // saber_hip_net_compact_arena ran: edges of disjoint lifetimes share memory (intermediate edges are not readable after a pass)
    // saber_hip_net_optimize flag 2048: other streams / processes run kernels on this device while the net does. Kernel variants whose
    // SPEED or completion depends on where the hardware places workgroups relative to each other - the persistent stage launch (needs
    // every workgroup of an image resident on its XCD at once), the cooperating-workgroup chains (tile codes 7 / 15), FP32 split-K
    // through one XCD's L2 - are then never selected: not statically, not by the autotuner, not from a restored selection.
    bool shared_device = false;
    // saber_hip_net_optimize flag 8192: FP32 ops keep their STATIC kernel selection - saber_hip_net_autotune and restored selections leave
    // them alone - so that two nets of one model answer bit-identically (the FP32 kernels differ in accumulation order; which one a
    // timing-based tuner picks depends on the box and the moment)
    bool reproducible_fp32 = false;
    bool inplace_external = false;   // captured nets: an in-place sum accumulates into a caller-owned input (api_capture.hip: readwrite)
    int coop_fallbacks = 0;   // cooperative launches that reported a failed pass (saber_hip_net_status), since the net was created
    // two-lane execution: independent branches (ResNet branch1 vs branch2a/2b) run on a side stream
    hipStream_t side = nullptr;
    hipEvent_t ev_start = nullptr, ev_join = nullptr;
    std::vector<hipEvent_t> ev_op;     // one per op that needs to publish its output to the other lane
    std::vector<int> writer;           // tensor id -> index of the op that last wrote it (-1: external)
    bool lanes_ready = false, has_side = false;
    std::vector<saber_hip_conv*> owned;   // ops created by saber_hip_net_optimize (destroyed with the net)
    std::vector<saber_hip_chain*> owned_chains;
    std::vector<saber_hip_chain_stage*> owned_stages;
    std::vector<saber_hip_stem_pair*> owned_stem_pairs;
};

// Op-list capture (api_capture.hip; saber_hip_capture_begin / _end): while g_capture is set on the calling thread every
// capturable *_run entry point records itself into the capture's net instead of launching. Each recorder returns a status.
namespace saber_api {
struct Capture;
extern thread_local Capture* g_capture;
int capture_conv(saber_hip_conv* op, const void* x, void* y, const void* res);
int capture_fc(saber_hip_fc* fc, const void* x, float* y, bool quantised_input);
int capture_unsupported(const char* what);
// streaming ops: kind + the argument block of the matching saber_hip_net_add_* call; in2 / out2 may be null
int capture_stream_op(OpKind kind, const char* name, const int* p, int np, const float* f, int nf, size_t count, const void* in,
                      size_t in_bytes, const void* in2, size_t in2_bytes, void* out, size_t out_bytes, void* out2, size_t out2_bytes);
}  // namespace saber_api

// helpers one translation unit defines and another uses
bool halo_ok(const saber_hip_conv* op);      // api_conv.hip
bool img_ok(const saber_hip_conv* op, int nw, int ib, int rb);      // api_conv.hip
bool stem_ok(const saber_hip_conv* op);      // api_conv.hip
void name_algo(saber_hip_conv* op);      // api_conv.hip
std::string stem_pair_name(const saber_api::NetOp& o);      // api_net_optimize.hip
void conv_fill_args(const saber_hip_conv* op, saber_mi355x::ConvKArgs& a, const void* x, void* y, const void* res);   // api_conv.hip
int stem_pool_args(const saber_hip_conv* op, const void* x, void* y, void* workspace, hipStream_t s, saber_mi355x::ConvKArgs* a);   // api_conv.hip
// FP32 split-K (b3 kernels): 2^sh workgroups per tile; needs >= 2 stages per split, a bounded partial buffer, and the
// workgroup -> XCD placement the hand-off relies on (checked once per device). split_prepare allocates the buffers.
bool split_ok(const saber_hip_conv* op, int tile, int ks, int sh);      // api_conv.hip
bool xcd_round_robin();      // api_conv.hip: workgroups 8 apart in a 1-D grid share an XCD on the current device (probed once)
int split_prepare(saber_hip_conv* op);      // api_conv.hip
// image-resident kernel variant of an INT8 conv on <= 64-pixel images (api_stage.hip)
bool img_conv_ok(const saber_hip_conv* op);
int img_conv_prepare(saber_hip_conv* op);
int pw_prepare(saber_hip_conv* op);      // api_conv.hip: the FP32 pointwise kernels' fragment-ordered weight planes, on demand
bool pw_eligible(const saber_hip_conv* op);
int img_conv_run(saber_hip_conv* op, const void* x, void* y, const void* res, void* y_pool, hipStream_t stream);
void img_conv_release(saber_hip_conv* op);
int net_launch(saber_hip_net* net, const NetOp& o, hipStream_t s);      // api_net.hip
bool fc_softmax_ok(const saber_hip_fc* fc, bool quantised_input = false);      // api_ops.hip: the INT8 small-batch fc kernel can normalise its own logits (fc_small.hip)
int fc_run_softmax(saber_hip_fc* fc, const void* x, float* y, float* prob, void* workspace, hipStream_t s, bool quantised_input);
int fc_softmax_prepare(saber_hip_fc* fc);        // ... allocates the arrival counter (not under stream capture)
void net_set_chain_mode(saber_hip_net* net, int ia, int mode);      // api_net_optimize.hip
int net_chain_mode(const saber_hip_net* net, int ia);      // api_net_optimize.hip
// the stage headed by ops[i0] (NetOp::stage) on / off: on forces every block's 3x3-led chain form and makes ops[i0] launch them all
void net_set_stage(saber_hip_net* net, int i0, bool on);   // api_net_optimize.hip
int stage_run(saber_hip_chain_stage* st, const void* x, const void* res, void* const* y1, void* const* y2, hipStream_t s);   // api_chain.hip
