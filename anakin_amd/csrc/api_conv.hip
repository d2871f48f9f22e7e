// anakin_amd/csrc/api_conv.hip - convolution half of the C ABI declared in include/saber_hip.h.
//
// Host side of the MI355X Saber target: what SaberConv2D<X86,*>::init/create/dispatch
// (saber/funcs/impl/x86/saber_conv.cpp:21-324) and GemmX8S8S32XConv::create
// (gemm_x8s8s32x_conv.cpp:40-185) do on the host - quantise + repack weights, pre-scale the bias,
// derive the per-channel requantisation scales, pick an algorithm - is done here once per operator;
// `*_run` only fills a POD argument block and enqueues kernels on the caller's stream.
#include "api_internal.h"

namespace saber_api {
thread_local std::string g_err;

void* zero_page() {
    static std::mutex mu;
    static void* pages[64] = {nullptr};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    if (!pages[dev]) {
        void* p = nullptr;
        if (hipMalloc(&p, 256) != hipSuccess) return nullptr;
        if (hipMemset(p, 0, 256) != hipSuccess || hipStreamSynchronize(nullptr) != hipSuccess) {
            (void)hipFree(p);
            return nullptr;
        }
        pages[dev] = p;
    }
    return pages[dev];
}

}  // namespace saber_api


const char* saber_hip_last_error(void) { return g_err.c_str(); }

int saber_hip_device_ok(void) {   // the CURRENT device of the calling thread must be a gfx950
    int n = 0, dev = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, dev) != hipSuccess) return 0;
    return std::strncmp(p.gcnArchName, "gfx950", 6) == 0 ? 1 : 0;
}

// ================================================================================================
// convolution
// ================================================================================================
static void choose_tile(saber_hip_conv* op) {
    // Largest tile that still yields >= ~1.5 workgroups per CU (256 CUs); otherwise the smallest.
    const long M = (long)op->d.n * op->oh * op->ow;
    const int order[] = {TILE_128x128, TILE_128x64, TILE_64x128, TILE_64x64, TILE_64x32, TILE_32x32};
    op->tile = TILE_32x32;
    for (int t : order) {
        int bmk, bnp;
        tile_dims(t, &bmk, &bnp);
        if (bmk > round_up(op->d.k, 32) && t != TILE_32x32) continue;
        const long blocks = ((M + bnp - 1) / bnp) * ((op->d.k + bmk - 1) / bmk);
        if (blocks >= 384) {
            op->tile = t;
            break;
        }
    }
}

bool halo_ok(const saber_hip_conv* op) {
    const saber_hip_conv_desc& d = op->d;
    return op->algo == ALGO_IGEMM_I8 && op->epi == EPI_I8_CONV && d.kh == 3 && d.kw == 3 && d.stride_h == 1 &&
           d.stride_w == 1 && d.dil_h == 1 && d.dil_w == 1 && d.group == 1 && op->c_eff % 64 == 0 && d.pad_h <= 1 &&
           d.pad_w <= 1;
}


bool img_ok(const saber_hip_conv* op, int nw, int ib, int rb) {
    return halo_ok(op) && !op->pair_k2 && conv3x3_img_feasible(op->c_eff, op->ow, op->oh, op->d.n, nw, ib, rb);
}

bool stem_ok(const saber_hip_conv* op) {
    const saber_hip_conv_desc& d = op->d;
    return op->algo == ALGO_IGEMM_I8_C4 && op->epi == EPI_I8_CONV && d.kh == 7 && d.kw == 7 && d.stride_h == 2 &&
           d.stride_w == 2 && d.dil_h == 1 && d.dil_w == 1 && d.group == 1;
}

// STATIC default of the bf16-plane FP32 kernel: on for MFMA-bound layers (a 3x3 or larger filter over >= 64 channels and
// >= 3136 output pixels), where the six bf16 MFMAs per slab beat the eight f32 ones (profiles/r03_*_fp32/; the latency-bound
// 1x1 and small-image layers do not gain); the RUNTIME strategy (saber_hip_conv2d_autotune) times both anyway.
static bool f32_static_b3(const saber_hip_conv* op) {
    return op->d.kh * op->d.kw > 1 && op->c_eff >= 64 && (long)op->d.n * op->oh * op->ow >= 3136;
}
// 1 when workgroup b of a 1-D grid runs on XCD (b + const) % 8 on the current device - workgroups 8 apart share an XCD
// (checked once per device with 2048-workgroup launches; 0 also on any failure): the split-K kernels hand partial sums over inside one XCD's L2 and are offered only then.
bool xcd_round_robin() {
    static std::mutex mu;
    static int state[64] = {0};          // 0 unknown, 1 yes, 2 no
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
    std::lock_guard<std::mutex> lock(mu);
    if (state[dev]) return state[dev] == 1;
    state[dev] = 2;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess || prop.multiProcessorCount != 256) {
        if (std::getenv("SABER_HIP_AUTOTUNE_LOG")) std::fprintf(stderr, "xcd_round_robin: %d CUs\n", prop.multiProcessorCount);
        return false;
    }
    const int n = 2048;
    DevBuf<unsigned> d;
    if (d.alloc_zero(n) != hipSuccess) return false;
    std::vector<unsigned> h(n, 99u);
    for (int rep = 0; rep < 2; ++rep) {
        if (launch_xcd_map_probe(d.p, n, nullptr) != hipSuccess || hipStreamSynchronize(nullptr) != hipSuccess) return false;
        if (hipMemcpy(h.data(), d.p, n * sizeof(unsigned), hipMemcpyDeviceToHost) != hipSuccess) return false;
        for (int b = 0; b < n; ++b)
            if (h[b] != ((h[0] + (unsigned)b) & 7u)) {      // (the first workgroup's XCD depends on what the queue dispatched before)
                if (std::getenv("SABER_HIP_AUTOTUNE_LOG")) std::fprintf(stderr, "xcd_round_robin: workgroup %d ran on XCD %u\n", b, h[b]);
                return false;
            }
    }
    state[dev] = 1;
    return true;
}
bool split_ok(const saber_hip_conv* op, int tile, int ks, int sh) {
    if (sh == 0) return true;
    if (sh < 0 || sh > 3 || !b3_tile_ok(op, tile, ks)) return false;
    if (op->no_placement) return false;      // a net that shares its device: the splits' common XCD is a dispatch property of an idle GPU
    const int steps = (op->Kg + 32 * ks - 1) / (32 * ks);
    if ((steps >> sh) < 2) return false;
    const size_t m = (size_t)op->d.n * op->oh * op->ow;
    if ((m + 127) * ((size_t)op->d.k + 127) * 4 * 8 > ((size_t)96 << 20)) return false;   // partial buffer: <= 96 MB
    return xcd_round_robin();
}
int split_prepare(saber_hip_conv* op) {
    if (op->d_part.p) return SABER_HIP_OK;
    const size_t m = (size_t)op->d.n * op->oh * op->ow;
    HIP_TRY(op->d_part.alloc_zero((m + 127) * ((size_t)op->d.k + 127) * 8));        // 8 splits of the padded f32 output
    HIP_TRY(op->d_part_ctr.alloc_zero(((m + 31) / 32) * (((size_t)op->d.k + 31) / 32)));
    if (!op->h_part_err) {
        HIP_TRY(hipHostMalloc((void**)&op->h_part_err, sizeof(unsigned), hipHostMallocMapped));
        *op->h_part_err = 0u;
    }
    return SABER_HIP_OK;
}
// A split-K launch of this operator found its splits on different XCDs (the kernel poisoned that output with NaN and counted
// itself in the pinned word): report it as an error status NOW and run without split-K from here on.
static int split_check(saber_hip_conv* op) {
    if (!op->ksplit || !op->h_part_err || !*(volatile unsigned*)op->h_part_err) return SABER_HIP_OK;
    *(volatile unsigned*)op->h_part_err = 0u;
    op->ksplit = 0;
    name_algo(op);
    return fail(SABER_HIP_RUNTIME_ERROR, "FP32 split-K: in an earlier launch of this operator the splits of a tile ran on different XCDs "
                "(that output was poisoned with NaN); split-K is now off for it");
}
void name_algo(saber_hip_conv* op) {
    static const char* an[] = {"igemm_i8", "igemm_i8_c4", "igemm_f32", "direct_i8", "direct_f32"};
    int bmk = 0, bnp = 0;
    tile_dims(op->tile, &bmk, &bnp);
    char buf[96];
    if (op->stem32) snprintf(buf, sizeof buf, "stem7x7s2_maxpool3x3s2_f32_bf16x3_nchw_in");
    else if (op->pool_fused) snprintf(buf, sizeof buf, "stem7x7s2_maxpool3x3s2_i8_4x8%s", op->pre_quant ? "_fusedquant" : "");
    else if (op->stem) snprintf(buf, sizeof buf, "stem7x7s2_i8_8x16%s", op->pre_quant ? "_fusedquant" : "");
    else if (op->fc_small) snprintf(buf, sizeof buf, op->algo == ALGO_IGEMM_F32 ? ((op->d_fcpart.p && !op->d_wfc.p) ? "fc_f32_splitk_16xk4" : "fc_f32_small_16xk4") : "fc_i8_small_16xk4");
    else if (op->pw > 1) {
        int ptm = 0, pp = 0, pd = 0, pmb = 0;
        (void)conv1x1_pwk_variant(op->pw - 1, &ptm, &pp, &pd, &pmb);
        snprintf(buf, sizeof buf, "pw1x1_f32_bf16x3_ksplit4_%dch_%dpx_d%d%s%s", ptm * 16, pp * 16, pd, pmb > 1 ? "_2wg" : "",
                 op->d.res_mode == SABER_HIP_RES_SUM_INPLACE ? "+sum" : "");
    } else if (op->pw) snprintf(buf, sizeof buf, "pw1x1_f32_bf16x3_regs_c%d_%dch_per_wave%s", op->c_eff, op->c_eff == 64 ? 64 : 32,
                              op->d.res_mode == SABER_HIP_RES_SUM_INPLACE ? "+sum" : "");
    else if (op->b3h) {
        int hb, ht, htm, hthr;
        (void)conv3x3_b3h_variant(op->b3h, &hb, &ht, &htm, &hthr);
        if (op->b3h >= 6) snprintf(buf, sizeof buf, "pw1x1_f32_bf16x3_%dch_%dpx_w%d", hb, ht * 16, hthr / 64);
        else snprintf(buf, sizeof buf, "halo3x3_f32_bf16x3_%dch_%dx16_w%d%s", hb, ht, hthr / 64, op->pool2 ? "+maxpool2x2" : "");
    }
    else if (op->img1) snprintf(buf, sizeof buf, "imgres%dx%d_i8_%dch%s", op->d.kh, op->d.kw, 16 * ((op->d.k / 16 + 31) / 32), op->gpool ? "+gpool" : "");
    else if (op->img_rb) snprintf(buf, sizeof buf, "img3x3_i8_%dimg_x_%drows_k16_w%d", op->img_ib, op->img_rb, op->img_nw);
    else if (op->halo) snprintf(buf, sizeof buf, "halo3x3_i8_%dx16", op->halo);
    else if (op->algo <= ALGO_IGEMM_F32)
        snprintf(buf, sizeof buf, "%s_%dx%d_k%d%s%s%s%s", op->b3 ? "igemm_f32_bf16x3" : an[op->algo], bmk, bnp, op->ks,
                 op->b3 && op->tile >= TILE_W8_64x64 ? "_w8" : "",
                 op->b3 && op->ksplit ? (op->ksplit == 1 ? "_split2" : (op->ksplit == 2 ? "_split4" : "_split8")) : "",
                 op->dma == 0 ? "" : (op->dma == 1 ? "_dma" : (op->dma == 2 ? "_dma_wg2" : "_dma_wg4")),
                 op->pool2 ? "+maxpool2x2" : "");
    else snprintf(buf, sizeof buf, "%s", an[op->algo]);
    op->algo_name = std::string(op->pair_k2 ? "pair_" : "") + buf;
}

int saber_hip_conv2d_create(const saber_hip_conv_desc* desc, saber_hip_conv_t** out) {
    if (!desc || !out) return fail(SABER_HIP_INVALID_VALUE, "null argument");
    const saber_hip_conv_desc& d = *desc;
    if (d.n <= 0 || d.h <= 0 || d.w <= 0 || d.c <= 0 || d.k <= 0 || d.kh <= 0 || d.kw <= 0 || d.group <= 0 ||
        d.stride_h <= 0 || d.stride_w <= 0 || d.dil_h <= 0 || d.dil_w <= 0 || d.pad_h < 0 || d.pad_w < 0)
        return fail(SABER_HIP_INVALID_VALUE, "bad conv geometry");
    if (d.c % d.group || d.k % d.group) return fail(SABER_HIP_INVALID_VALUE, "invalid input_channel or output_channel");
    if ((d.act != SABER_HIP_ACT_NONE && d.act != SABER_HIP_ACT_RELU) ||
        (d.res_act != SABER_HIP_ACT_NONE && d.res_act != SABER_HIP_ACT_RELU))
        return fail(SABER_HIP_UNIMPL, "only ReLU is fused into the convolution (as in the x86 INT8 path); other activations are separate ops");
    if (d.act_negative_slope != 0.f && (d.int8_weights || d.act != SABER_HIP_ACT_RELU))
        return fail(SABER_HIP_UNIMPL, "negative_slope is honoured by the FP32 convolution with Active_relu only (the x86 INT8 conv clamps to 0)");
    const int oh = conv_out(d.h, d.pad_h, d.kh, d.dil_h, d.stride_h);
    const int ow = conv_out(d.w, d.pad_w, d.kw, d.dil_w, d.stride_w);
    if (oh <= 0 || ow <= 0) return fail(SABER_HIP_INVALID_VALUE, "empty output");
    if (saber_hip_device_ok()) (void)zero_page();   // allocate outside any stream capture
    auto* op = new saber_hip_conv();
    op->d = d;
    op->oh = oh;
    op->ow = ow;
    op->is_i8 = d.int8_weights != 0;
    op->c_eff = d.c;
    const size_t in_pixels = (size_t)d.n * d.h * d.w;
    if (op->is_i8) {
        if (d.out_dtype != SABER_HIP_F32 && d.out_layout != SABER_HIP_NHWC) {
            delete op;
            return fail(SABER_HIP_UNIMPL, "8-bit outputs are NHWC (calibrator_parse.cpp:194-244)");
        }
        if (d.out_dtype == SABER_HIP_F32 && d.out_layout != SABER_HIP_NHWC) {
            delete op;
            return fail(SABER_HIP_UNIMPL, "INT8 conv with f32 output: NHWC only");
        }
        op->x_dtype = d.in_dtype;
        if (d.in_dtype == SABER_HIP_F32) {
            // quantise on entry: SaberConv2D<X86,AK_INT8>::dispatch -> reorder_nhwc_nchw (saber_conv.cpp:308)
            if (d.in_layout != SABER_HIP_NCHW) {
                delete op;
                return fail(SABER_HIP_UNIMPL, "f32 input of an INT8 conv must be NCHW");
            }
            op->pre_quant = true;
            op->x_dtype = DT_S8;
            op->c_eff = (d.c < 4 && d.group == 1) ? 4 : d.c;
            op->ws_bytes = in_pixels * op->c_eff;
        } else {
            if (d.in_layout != SABER_HIP_NHWC) {
                delete op;
                return fail(SABER_HIP_UNIMPL, "8-bit inputs are NHWC");
            }
            if (d.c < 4 && d.group == 1) {
                op->pre_pad = true;
                op->c_eff = 4;
                op->ws_bytes = in_pixels * 4;
            }
        }
        if (d.group == 1 && op->c_eff % 16 == 0) op->algo = ALGO_IGEMM_I8;
        else if (d.group == 1 && op->c_eff == 4) op->algo = ALGO_IGEMM_I8_C4;
        else op->algo = ALGO_DIRECT_I8;
        if (op->algo == ALGO_DIRECT_I8 && d.res_mode == SABER_HIP_RES_NONE) { /* fine */ }
        op->epi = EPI_I8_CONV;
        if (d.res_mode == SABER_HIP_RES_SUM_INPLACE && d.res_has_dtype &&
            ((d.out_dtype == SABER_HIP_F32) != (d.res_dtype == SABER_HIP_F32) ||
             (d.res_dtype != SABER_HIP_F32 && d.res_dtype != SABER_HIP_S8 && d.res_dtype != SABER_HIP_U8))) {
            delete op;
            return fail(SABER_HIP_INVALID_VALUE, "RES_SUM_INPLACE: the bytes in y must have the output's element size (s8 / u8 into an 8-bit output)");
        }
        if (d.res_mode == SABER_HIP_RES_ELTWISE && d.out_dtype != SABER_HIP_S8) {
            delete op;
            return fail(SABER_HIP_INVALID_VALUE, "RES_ELTWISE produces s8 (SaberEltwise<X86,AK_INT8>)");
        }
        if (d.res_stride > 1) {
            if (d.res_mode != SABER_HIP_RES_ELTWISE || op->algo == ALGO_DIRECT_I8 ||
                (d.res_h - 1) / d.res_stride + 1 != oh || (d.res_w - 1) / d.res_stride + 1 != ow) {
                delete op;
                return fail(SABER_HIP_INVALID_VALUE, "res_stride: RES_ELTWISE on the implicit-GEMM path with (res_h - 1) / s + 1 == oh, (res_w - 1) / s + 1 == ow");
            }
        }
    } else {
        if (d.in_dtype != SABER_HIP_F32 || d.out_dtype != SABER_HIP_F32) {
            delete op;
            return fail(SABER_HIP_INVALID_VALUE, "FP32 conv needs f32 tensors");
        }
        if (d.res_mode == SABER_HIP_RES_ELTWISE) {
            delete op;
            return fail(SABER_HIP_UNIMPL, "RES_ELTWISE is INT8 only");
        }
        op->x_dtype = DT_F32;
        op->epi = EPI_F32;
        if (d.in_layout == SABER_HIP_NCHW) {
            op->pre_transpose = true;
            op->c_eff = d.group == 1 ? round_up(d.c, 4) : d.c;
            op->ws_bytes = in_pixels * op->c_eff * sizeof(float);
        }
        if (d.group == 1 && op->c_eff % 4 == 0) op->algo = ALGO_IGEMM_F32;
        else op->algo = ALGO_DIRECT_F32;
    }
    if (op->algo == ALGO_IGEMM_I8_C4) {
        op->kw_pad = round_up(d.kw, 4);
        op->Kg = d.kh * op->kw_pad * 4;
        op->Kg_pad = round_up(op->Kg, 256);    // register-staged only: widest stage is 256 B
    } else if (op->algo == ALGO_IGEMM_I8) {
        op->Kg = d.kh * d.kw * op->c_eff;
        op->Kg_pad = round_up(op->Kg, 1024);   // widest stage: 4 k-steps x 4 wave groups x 64 B; the zero tail
                                               // keeps out-of-range k-steps inert (also for XOR-shifted u8 pads)
    } else if (op->algo == ALGO_IGEMM_F32) {
        op->Kg = d.kh * d.kw * op->c_eff;
        op->Kg_pad = round_up(op->Kg, 256);    // f32 elements: 1024 B
    }
    op->stem = stem_ok(op) ? 1 : 0;
    choose_tile(op);
    {   // stage depth: as many 64-byte k-steps per barrier as the reduction has (max 4)
        const int kbytes = op->Kg * (op->algo == ALGO_IGEMM_F32 ? 4 : 1);
        op->ks = kbytes >= 256 ? 4 : (kbytes >= 128 ? 2 : 1);
    }
    name_algo(op);
    *out = op;
    return SABER_HIP_OK;
}

void saber_hip_conv2d_out_shape(const saber_hip_conv_t* op, int* oh, int* ow) {
    if (oh) *oh = (op->pool_fused || op->pool2) ? op->pool_oh : op->oh;
    if (ow) *ow = (op->pool_fused || op->pool2) ? op->pool_ow : op->ow;
}

int saber_hip_conv2d_set_pooling(saber_hip_conv_t* op, int pool_type, int kh, int kw, int stride_h, int stride_w,
                                 int pad_h, int pad_w, int floor_mode) {
    if (!op) return fail(SABER_HIP_INVALID_VALUE, "null argument");
    const saber_hip_conv_desc& d = op->d;
    // FP32: any implicit-GEMM conv with relu + 2x2 / stride-2 / unpadded max pooling over even output dims (VGG16's five
    // conv+relu+pool stages): pool-ordered GEMM columns, maximum taken in the epilogue (conv_igemm_impl.h)
    if (!op->is_i8 && op->algo == ALGO_IGEMM_F32 && pool_type == SABER_HIP_POOL_MAX && kh == 2 && kw == 2 && stride_h == 2 &&
        stride_w == 2 && pad_h == 0 && pad_w == 0 && d.res_mode == SABER_HIP_RES_NONE && d.act == SABER_HIP_ACT_RELU &&
        d.act_negative_slope == 0.f && d.out_layout == SABER_HIP_NHWC && !op->pair_k2 && op->oh % 2 == 0 && op->ow % 2 == 0) {
        op->pool_oh = op->oh / 2;
        op->pool_ow = op->ow / 2;
        op->pool2 = 1;
        name_algo(op);
        return SABER_HIP_OK;
    }
    // FP32 stem: NCHW f32 image -> 7x7 / 2 / pad 3 conv (3 -> 64) + relu -> 3x3 / 2 max pooling -> NHWC f32 as ONE launch (conv_stem_f32.hip)
    if (!op->is_i8 && op->algo == ALGO_IGEMM_F32 && op->pre_transpose && d.in_dtype == SABER_HIP_F32 && d.out_dtype == SABER_HIP_F32 &&
        d.out_layout == SABER_HIP_NHWC && d.c == 3 && d.k == 64 && d.kh == 7 && d.kw == 7 && d.stride_h == 2 && d.stride_w == 2 && d.pad_h == 3 &&
        d.pad_w == 3 && d.dil_h == 1 && d.dil_w == 1 && d.group == 1 && d.res_mode == SABER_HIP_RES_NONE && d.act == SABER_HIP_ACT_RELU &&
        d.act_negative_slope == 0.f && !op->pair_k2 && pool_type == SABER_HIP_POOL_MAX && kh == 3 && kw == 3 && stride_h == 2 && stride_w == 2 &&
        pad_h == 0 && pad_w == 0 && op->weights_set && op->w_stem_host.size() == (size_t)64 * 3 * 49 && !getenv("SABER_HIP_NO_STEM_F32")) {
        const int poh = saber_hip_pool_out_dim(op->oh, pad_h, kh, stride_h, floor_mode), pow_ = saber_hip_pool_out_dim(op->ow, pad_w, kw, stride_w, floor_mode);
        if ((poh - 1) * 2 < op->oh && (pow_ - 1) * 2 < op->ow) {      // every window starts inside the conv image (ceil and floor shapes)
            std::vector<uint8_t> fr;
            stem_f32_pack(op->w_stem_host.data(), fr);
            HIP_TRY(op->d_wstem32.upload(fr));
            op->pool_oh = poh;
            op->pool_ow = pow_;
            op->pool_fused = 1;
            op->stem32 = 1;
            op->ws_bytes = 0;      // (no NHWC4 copy of the image any more)
            name_algo(op);
            return SABER_HIP_OK;
        }
    }
    const bool fusable = stem_ok(op) && pool_type == SABER_HIP_POOL_MAX && kh == 3 && kw == 3 && stride_h == 2 &&
                         stride_w == 2 && pad_h == 0 && pad_w == 0 && d.res_mode == SABER_HIP_RES_NONE &&
                         (d.out_dtype == SABER_HIP_S8 || d.out_dtype == SABER_HIP_U8);
    if (!fusable)
        return fail(SABER_HIP_UNIMPL, "conv+pooling: no fused kernel for this combination (run the two ops)");
    op->pool_oh = saber_hip_pool_out_dim(op->oh, pad_h, kh, stride_h, floor_mode);
    op->pool_ow = saber_hip_pool_out_dim(op->ow, pad_w, kw, stride_w, floor_mode);
    // every 3x3 window of the fused kernel must start inside the conv image (true for the ceil and floor shapes)
    if ((op->pool_oh - 1) * 2 >= op->oh || (op->pool_ow - 1) * 2 >= op->ow)
        return fail(SABER_HIP_UNIMPL, "conv+pooling: pooled shape outside the conv image");
    op->pool_fused = 1;
    name_algo(op);
    return SABER_HIP_OK;
}
size_t saber_hip_conv2d_workspace_bytes(const saber_hip_conv_t* op) { return op->ws_bytes; }
const char* saber_hip_conv2d_algo(const saber_hip_conv_t* op) { return op->algo_name.c_str(); }

static inline bool tile_arg_ks(int ks) { return ks == 1 || ks == 2 || ks == 4; }
// every specialised-kernel selector off (run / get_tile test img1 and b3h FIRST): a set_tile that selects one kernel family
// starts from here, so a selection made by an earlier autotune / set_tile cannot keep running under the new one's name
static void clear_selectors(saber_hip_conv* op) {
    if (!op->gpool) op->img1 = 0;      // (conv + fused global pooling exists only as the image-resident kernel)
    op->b3h = 0; op->b3 = 0; op->ksplit = 0; op->halo = 0; op->stem = 0; op->img_ib = op->img_rb = 0; op->fc_small = 0; op->pw = 0;
}

int saber_hip_conv2d_set_tile(saber_hip_conv_t* op, int tile) {
    // tile id in the low byte, optional stage depth (k-steps per stage: 1, 2, 4) in bits 8..15,
    // optional staging variant in bits 16..23 (1 = register-staged, 2 = LDS-DMA ring, 3 / 4 = LDS-DMA ring
    // with 2 / 4 wave groups: needs stage depth 4 and a 32x32, 64x32 or 64x64 tile)
    if (op->stem32) {      // FP32 stem launch: variant 15, low byte 0 = tile by launch size, 1 = 8 x 8 pooled pixels per workgroup, 2 = 4 x 8, 3 = 4 x 4
        if (((tile >> 16) & 0xff) != 15 || (tile & 0xff) > 3) return fail(SABER_HIP_INVALID_VALUE, "FP32 stem launch: (15 << 16) | 0..3");
        op->stem32 = 1 + (tile & 0xff);
        return SABER_HIP_OK;
    }
    if (op->pool_fused) return fail(SABER_HIP_INVALID_VALUE, "fused conv+pooling has a single kernel");
    const int ks = (tile >> 8) & 0xff;
    const int var = (tile >> 16) & 0xff;
    tile &= 0xff;
    if (var == 7 || var == 8) {   // stem kernel on / off (first-layer path)
        if (var == 7 && !stem_ok(op)) return fail(SABER_HIP_INVALID_VALUE, "stem kernel needs an INT8 7x7 stride-2 conv with <= 4 channels");
        op->stem = var == 7;
        tile &= 0xff;
        if (var == 8 && tile < TILE_COUNT) op->tile = tile;
        if (var == 8 && ((tile_arg_ks(ks)))) op->ks = ks;
        name_algo(op);
        return SABER_HIP_OK;
    }
    if (var == 11) {   // FP32 implicit GEMM on three bf16 planes (register-staged, one 32-deep slab per stage)
        if (!b3_ok(op)) return fail(SABER_HIP_INVALID_VALUE, "bf16x3 variant: FP32 implicit-GEMM conv with C % 8 == 0 (not a sibling pair, not an fc)");
        const int sh = ks >> 4, ksd = ks & 15;      // bits 12..15 of the code: log2 of the split-K factor
        if (!b3_tile_ok(op, tile, ksd ? ksd : 1))     // tiles 6..9: the 8-wave forms
            return fail(SABER_HIP_INVALID_VALUE, "bf16x3: tile 0..9, stage depth 1 (or 2 below 128x128), 256x128 only for k padded to a multiple of 256");
        if (!split_ok(op, tile, ksd ? ksd : 1, sh)) return fail(SABER_HIP_INVALID_VALUE, "bf16x3 split-K: 2 / 4 / 8 splits with >= 2 stages each, bounded output, 8 x 32 CU device");
        if (sh) {
            const int rc = split_prepare(op);
            if (rc) return rc;
        }
        clear_selectors(op);
        op->b3 = 1; op->dma = 0; op->ks = ksd ? ksd : 1; op->tile = tile; op->ksplit = sh;
        name_algo(op);
        return SABER_HIP_OK;
    }
    if (var == 14) {   // FP32 pointwise conv, C = 64 / 128: persistent waves with their weight planes in registers (conv1x1_pw.hip)
        // low byte 0: the register-weights kernel (C = 64 / 128); 1 .. 4: the reduction-split kernel's variants (C = 128 .. 2048)
        if (pw_prepare(op) != SABER_HIP_OK || (tile == 0 ? !pw_ok(op) : !pwk_ok(op, tile)))
            return fail(SABER_HIP_INVALID_VALUE, "pointwise kernels: FP32 NHWC 1x1 / stride-1 conv, K % 64 == 0, C in {64, 128} (variant 0) or C % 128 == 0 (1..4)");
        clear_selectors(op);
        op->pw = 1 + tile; op->dma = 0;
        name_algo(op);
        return SABER_HIP_OK;
    }
    if (var == 13) {   // FP32 3x3 LDS-halo kernel on the bf16 planes, variant 1..5 in the low byte
        if (!b3h_ok(op, tile)) return fail(SABER_HIP_INVALID_VALUE, "bf16x3 halo kernel: FP32 NHWC stride-1 conv, 3x3 pad 1 with C % 32 == 0 (variant 1..5) or 1x1 with C % 64 == 0 (6..8)");
        clear_selectors(op);
        op->b3h = tile; op->dma = 0;
        name_algo(op);
        return SABER_HIP_OK;
    }
    if (var == 12) {   // image-resident kernel (<= 64 pixels per image: one workgroup = one image x a channel group)
        const int rc = img_conv_prepare(op);
        if (rc) return rc;
        clear_selectors(op);
        op->img1 = 1;
        name_algo(op);
        return SABER_HIP_OK;
    }
    if (op->gpool) return fail(SABER_HIP_INVALID_VALUE, "conv + fused global pooling has a single kernel");
    if (var == 10) {   // small-batch fc kernel
        if (!fc_small_ok(op)) return fail(SABER_HIP_INVALID_VALUE, "small-batch fc kernel: INT8 fc with <= 16 rows and k <= 4096");
        clear_selectors(op);
        op->fc_small = 1;
        name_algo(op);
        return SABER_HIP_OK;
    }
    if (var == 9) {   // small-image 3x3 kernel: output rows per slab in the low byte, images per slab in bits 8..15
        const int rb = tile, ib = ks & 0x7f, nw = (ks & 0x80) ? 8 : 4;   // bit 15: 8 waves per workgroup
        if (!img_ok(op, nw, ib, rb))
            return fail(SABER_HIP_INVALID_VALUE, "small-image 3x3 kernel: needs an INT8 3x3 stride-1 conv with C in {64,128,256,512} "
                                                 "and a slab (images x rows) that fits its LDS / accumulator budget");
        clear_selectors(op);
        op->img_ib = ib; op->img_rb = rb; op->img_nw = nw;
        name_algo(op);
        return SABER_HIP_OK;
    }
    if (var == 5 || var == 6) {   // LDS-halo 3x3 kernel, 4 / 8 tile rows
        if (!halo_ok(op) || op->pair_k2) return fail(SABER_HIP_INVALID_VALUE, "halo kernel needs an INT8 3x3 stride-1 conv with C % 64 == 0");
        clear_selectors(op);
        op->halo = var == 5 ? 4 : 8;
        name_algo(op);
        return SABER_HIP_OK;
    }
    if (var) clear_selectors(op);   // an explicit implicit-GEMM variant switches the specialised kernels off
    if (var > 4 || (var >= 2 && op->algo == ALGO_IGEMM_I8_C4)) return fail(SABER_HIP_INVALID_VALUE, "bad staging variant");
    if ((var >= 3 && ((ks ? ks : op->ks) != 4 || tile > TILE_64x64)) || (var == 4 && tile != TILE_32x32))
        return fail(SABER_HIP_INVALID_VALUE, "wave groups need stage depth 4 and a tile <= 64x64 (32x32 for 4 groups)");
    if (var) op->dma = var == 1 ? 0 : (var == 2 ? 1 : (var == 3 ? 2 : 4));
    if (tile < 0 || tile >= TILE_COUNT || !(ks == 0 || ks == 1 || ks == 2 || ks == 4))
        return fail(SABER_HIP_INVALID_VALUE, "bad tile id");
    if (ks) op->ks = ks;
    op->tile = tile;
    name_algo(op);
    return SABER_HIP_OK;
}
int saber_hip_conv2d_get_tile(const saber_hip_conv_t* op) {
    if (op->pw) return (14 << 16) | (op->pw - 1);
    if (op->b3h) return op->b3h | (13 << 16);
    if (op->img1) return 12 << 16;
    if (op->fc_small) return 10 << 16;
    if (op->b3) return op->tile | ((op->ks | (op->ksplit << 4)) << 8) | (11 << 16);
    if (op->stem) return 7 << 16;
    if (op->img_rb) return op->img_rb | ((op->img_ib | (op->img_nw == 8 ? 0x80 : 0)) << 8) | (9 << 16);
    if (op->halo) return op->tile | (op->ks << 8) | ((op->halo == 4 ? 5 : 6) << 16);
    const int var = op->dma == 0 ? 1 : (op->dma == 1 ? 2 : (op->dma == 2 ? 3 : 4));
    return op->tile | (op->ks << 8) | (var << 16);
}

int saber_hip_conv2d_set_weights(saber_hip_conv_t* op, const void* w, int w_dtype, const float* w_scale,
                                 const float* bias, float in_scale, float out_scale) {
    if (!op || !w) return fail(SABER_HIP_INVALID_VALUE, "null argument");
    const saber_hip_conv_desc& d = op->d;
    const int K = d.k, Cg = d.c / d.group, kh = d.kh, kw = d.kw;
    const size_t inner = (size_t)Cg * kh * kw;
    op->in_scale = in_scale;
    op->out_scale = out_scale;
    op->has_bias = bias != nullptr;
    if (bias) op->bias_host.assign(bias, bias + K);
    else op->bias_host.clear();
    const int K_pad = round_up(K, 128);
    if (op->is_i8) {
        // ---- weights: quantise (if f32) exactly as scale_conv_weights_to_nchw_host -------------
        op->wq_oihw.assign((size_t)K * inner, 0);
        op->w_scale.assign(K, 0.f);
        if (w_dtype == SABER_HIP_F32) {
            const float* wf = (const float*)w;
            for (int k = 0; k < K; ++k) {
                float max_val = -1e20f;  // get_tensor_scale, x86_utils.h:141-166
                for (size_t i = 0; i < inner; ++i) {
                    const float a = fabsf(wf[k * inner + i]);
                    max_val = a > max_val ? a : max_val;
                }
                const float sc = max_val / 127.f;
                op->w_scale[k] = sc;
                for (size_t i = 0; i < inner; ++i)  // static_cast<char>(w / scale): truncation, x86_utils.h:316
                    op->wq_oihw[k * inner + i] = (int8_t)(wf[k * inner + i] / sc);
            }
        } else if (w_dtype == SABER_HIP_S8) {
            if (!w_scale) return fail(SABER_HIP_INVALID_VALUE, "s8 weights need w_scale[k]");
            std::memcpy(op->wq_oihw.data(), w, (size_t)K * inner);
            std::memcpy(op->w_scale.data(), w_scale, sizeof(float) * K);
        } else {
            return fail(SABER_HIP_INVALID_VALUE, "weights must be f32 or s8");
        }
        // ---- per-channel bias' and scale: GemmX8S8S32XConv::create :90-105, :145-182 -----------
        const int in_dt = d.in_dtype == SABER_HIP_F32 ? DT_S8 : d.in_dtype;  // f32 input is quantised to s8
        std::vector<float> bias_p(K, 0.f), scale(K, 0.f);
        for (int k = 0; k < K; ++k) {
            float s_in;
            if (in_dt == DT_U8) s_in = op->w_scale[k] * in_scale * (127.f / 255.f);
            else s_in = op->w_scale[k] * in_scale;
            if (bias) bias_p[k] = bias[k] * (1.f / s_in);
            if (d.out_dtype == SABER_HIP_F32) scale[k] = s_in;
            else if (d.out_dtype == SABER_HIP_U8) scale[k] = s_in / (out_scale * (127.f / 255.f));
            else scale[k] = s_in / out_scale;
        }
        // ---- repack -----------------------------------------------------------------------------
        std::vector<uint8_t> wr;
        std::vector<int> comp;
        const int8_t* q = op->wq_oihw.data();
        if (op->algo == ALGO_IGEMM_I8 || op->algo == ALGO_IGEMM_I8_C4) {
            wr.assign((size_t)K_pad * op->Kg_pad, 0);
            const int Ce = op->c_eff;
            for (int k = 0; k < K; ++k)
                for (int c = 0; c < Cg; ++c)
                    for (int i = 0; i < kh; ++i)
                        for (int j = 0; j < kw; ++j) {
                            const int8_t v = q[(((size_t)k * Cg + c) * kh + i) * kw + j];
                            size_t kk = op->algo == ALGO_IGEMM_I8 ? ((size_t)(i * kw + j) * Ce + c)
                                                                   : ((size_t)(i * op->kw_pad + j) * 4 + c);
                            wr[(size_t)k * op->Kg_pad + kk] = (uint8_t)v;
                        }
            if (in_dt == DT_U8) {  // +128 * sum(w): compensation of the u8 -> s8 shift
                comp.assign(K_pad, 0);
                for (int k = 0; k < K; ++k) {
                    int s = 0;
                    for (size_t i = 0; i < inner; ++i) s += (int)q[k * inner + i];
                    comp[k] = 128 * s;
                }
            }
        } else {  // direct: [K][kh][kw][Cg]
            wr.assign((size_t)K * inner, 0);
            for (int k = 0; k < K; ++k)
                for (int c = 0; c < Cg; ++c)
                    for (int i = 0; i < kh; ++i)
                        for (int j = 0; j < kw; ++j)
                            wr[(((size_t)k * kh + i) * kw + j) * Cg + c] =
                                (uint8_t)q[(((size_t)k * Cg + c) * kh + i) * kw + j];
        }
        bias_p.resize(K_pad, 0.f);
        scale.resize(K_pad, 0.f);
        op->bias_p_host = bias_p;
        op->scale_host = scale;
        op->comp_host = comp;
        HIP_TRY(op->d_w.upload(wr));
        HIP_TRY(op->d_bias.upload(bias_p));
        HIP_TRY(op->d_scale.upload(scale));
        op->has_comp = !comp.empty();
        if (op->has_comp) HIP_TRY(op->d_comp.upload(comp));
    } else {
        if (w_dtype != SABER_HIP_F32) return fail(SABER_HIP_INVALID_VALUE, "FP32 conv needs f32 weights");
        const float* wf = (const float*)w;
        std::vector<float> wr;
        if (d.group == 1 && Cg == 3 && kh == 7 && kw == 7 && K == 64) op->w_stem_host.assign(wf, wf + (size_t)K * inner);      // (the FP32 stem launch packs its own planes: set_pooling)
        if (op->algo == ALGO_IGEMM_F32) {
            wr.assign((size_t)K_pad * op->Kg_pad, 0.f);
            const int Ce = op->c_eff;
            for (int k = 0; k < K; ++k)
                for (int c = 0; c < Cg; ++c)
                    for (int i = 0; i < kh; ++i)
                        for (int j = 0; j < kw; ++j)
                            wr[(size_t)k * op->Kg_pad + (size_t)(i * kw + j) * Ce + c] =
                                wf[(((size_t)k * Cg + c) * kh + i) * kw + j];
        } else {
            wr.assign((size_t)K * inner, 0.f);
            for (int k = 0; k < K; ++k)
                for (int c = 0; c < Cg; ++c)
                    for (int i = 0; i < kh; ++i)
                        for (int j = 0; j < kw; ++j)
                            wr[(((size_t)k * kh + i) * kw + j) * Cg + c] = wf[(((size_t)k * Cg + c) * kh + i) * kw + j];
        }
        std::vector<uint8_t> raw((const uint8_t*)wr.data(), (const uint8_t*)wr.data() + wr.size() * sizeof(float));
        HIP_TRY(op->d_w.upload(raw));
        // an fc with few output tiles: scratch of the split-K kernel (fc_f32_splitk.hip) - OPT-IN (SABER_HIP_FC_F32_SPLITK=1): measured no
        // faster than the one-workgroup-per-tile fc + the softmax launch (18.3 vs 16.7 us at batch 8, 12.6 vs 13.2 at batch 1,
        // profiles/r06/fc_tail.txt): the split adds two round trips through the device's coherence point to a tail that is a latency chain
        {
            const char* e = getenv("SABER_HIP_FC_F32_SPLITK");
            if (op->algo == ALGO_IGEMM_F32 && d.h == 1 && d.w == 1 && kh == 1 && kw == 1 && d.group == 1 && op->epi == EPI_F32 && !op->pre_transpose &&
                d.out_layout == SABER_HIP_NHWC && d.res_mode == SABER_HIP_RES_NONE && e && e[0] == '1' &&
                fc_f32_splitk_ok(d.n, op->c_eff, op->Kg_pad, K, false)) {
                HIP_TRY(op->d_fcpart.alloc_zero(fc_f32_splitk_part_floats(op->c_eff, K)));
                HIP_TRY(op->d_fcctr.alloc_zero(fc_f32_splitk_counters(K)));
                if (op->fc_small) name_algo(op);
            }
        }
        // an fc at <= 16 batch rows is ONE pass over its weights. SABER_HIP_FC_F32_PACKED=1 (A/B): the same matrix once more
        // fragment-major, so that every load instruction of the streaming kernel reads 1 KB contiguous ([16-output tile][16-float step]
        // [lane][4]; fc_small.hip) - superseded by the kernel that reads the row-major weights in contiguous runs and transposes in LDS.
        if (op->algo == ALGO_IGEMM_F32 && d.h == 1 && d.w == 1 && kh == 1 && kw == 1 && fc_f32_small_ok(d.n, op->c_eff, op->Kg_pad) &&
            getenv("SABER_HIP_FC_F32_PACKED")) {      // (A/B only since gemm_f32_rows_lds_kernel: the row-major weights stream faster)
            const int Ce = op->c_eff, tiles = (K + 15) / 16, steps = fc_f32_packed_steps(Ce);
            std::vector<float> pk((size_t)tiles * steps * 256, 0.f);
            for (int t = 0; t < tiles; ++t)
                for (int r = 0; r < 16; ++r) {
                    const int nrow = t * 16 + r;
                    if (nrow >= K) continue;
                    const float* src = wr.data() + (size_t)nrow * op->Kg_pad;
                    for (int k = 0; k < Ce; ++k) {
                        const int st = k >> 4, g = (k >> 2) & 3, e = k & 3;
                        pk[(((size_t)t * steps + st) * 64 + g * 16 + r) * 4 + e] = src[k];
                    }
                }
            HIP_TRY(op->d_wfc.upload(pk));
        }
        // the same matrix as three bf16 planes (w = h + m + l exactly: 3 x 8 mantissa bits) for the bf16-MFMA variant; spatial
        // convolutions only (an fc streams its weights once: 6 bytes per weight instead of 4 would only slow it down)
        if (op->algo == ALGO_IGEMM_F32 && op->c_eff % 8 == 0 && (long)d.h * d.w > 1 && !getenv("SABER_HIP_NO_BF16X3")) {
            auto rne = [](float x) {
                uint32_t u;
                std::memcpy(&u, &x, 4);
                u += 0x7fffu + ((u >> 16) & 1u);
                return (uint16_t)(u >> 16);
            };
            auto bf = [](uint16_t h) {
                const uint32_t u = (uint32_t)h << 16;
                float f;
                std::memcpy(&f, &u, 4);
                return f;
            };
            const size_t n = wr.size();
            std::vector<uint8_t> planes(n * 6);
            uint16_t* pl = (uint16_t*)planes.data();
            for (size_t i = 0; i < n; ++i) {
                const uint16_t h = rne(wr[i]);
                const float r1 = wr[i] - bf(h);
                const uint16_t m = rne(r1);
                const float r2 = r1 - bf(m);
                pl[i] = h; pl[n + i] = m; pl[2 * n + i] = rne(r2);
            }
            HIP_TRY(op->d_w3.upload(planes));
            // 3x3 / stride 1 / pad 1 on NHWC f32: the same planes once more in MFMA A-fragment order for the LDS-halo kernel
            // (conv3x3_b3h.hip): [16-row tile][32-channel chunk][tap][plane][lane] x 8 bf16. Row rho of tile i of a wave's tm
            // tiles is channel  base + (rho >> 2) * 4 tm + 4 i + (rho & 3)  (a lane then owns 4 tm consecutive channels).
            const bool h3 = kh == 3 && kw == 3 && d.pad_h == 1 && d.pad_w == 1 && op->c_eff % 32 == 0;
            const bool h1 = kh == 1 && kw == 1 && d.pad_h == 0 && d.pad_w == 0 && op->c_eff % 64 == 0;      // the pointwise form of the same kernel
            if ((h3 || h1) && d.stride_h == 1 && d.stride_w == 1 && d.dil_h == 1 && d.dil_w == 1 &&
                d.group == 1 && !op->pre_transpose && d.out_layout == SABER_HIP_NHWC) {
                const int Ce = op->c_eff, nch = Ce / 32, ktiles = K_pad / 16, taps = kh * kw;
                for (int tm = 1; tm <= 2; ++tm) {
                    std::vector<uint8_t> fr((size_t)ktiles * nch * taps * 3 * 64 * 16, 0);
                    uint16_t* fp = (uint16_t*)fr.data();
                    for (int kt = 0; kt < ktiles; ++kt)
                        for (int cc = 0; cc < nch; ++cc)
                            for (int t = 0; t < taps; ++t)
                                for (int lane = 0; lane < 64; ++lane) {
                                    const int rho = lane & 15, fq = lane >> 4;
                                    const int ch = (kt / tm) * tm * 16 + (rho >> 2) * 4 * tm + (kt % tm) * 4 + (rho & 3);
                                    for (int e = 0; e < 8; ++e) {
                                        const size_t src = (size_t)ch * op->Kg_pad + (size_t)t * Ce + cc * 32 + fq * 8 + e;
                                        for (int pl3 = 0; pl3 < 3; ++pl3)
                                            fp[((((size_t)(kt * nch + cc) * taps + t) * 3 + pl3) * 64 + lane) * 8 + e] = pl[pl3 * n + src];
                                    }
                                }
                    HIP_TRY((tm == 1 ? op->d_w3h1 : op->d_w3h2).upload(fr));
                }
            }
            // (the pointwise kernels' fragment-ordered planes, d_wpw, are packed ON DEMAND from d_w3 - pw_prepare below: the autotuner's
            // candidate list, set_tile(14 << 16) and a restored selection ask for them; a net that never selects those kernels - the
            // static selection never does - no longer carries K * C * 6 dead bytes per 1x1 conv (round-5 advisor))
            // STATIC choice (BaseFunc STATIC strategy): SABER_HIP_F32_BF16X3=1 makes the bf16-plane kernel the default of every
            // eligible FP32 convolution (0 keeps the f32-MFMA kernels); unset: see f32_static_b3()
            const char* e = getenv("SABER_HIP_F32_BF16X3");
            const bool want = e ? (e[0] == '1') : f32_static_b3(op);
            if (want && !op->pair_k2) { op->b3 = 1; op->ks = 1; op->dma = 0; name_algo(op); }
        }
        std::vector<float> b(K_pad, 0.f);
        if (bias) std::memcpy(b.data(), bias, sizeof(float) * K);
        HIP_TRY(op->d_bias.upload(b));
    }
    op->weights_set = true;
    return SABER_HIP_OK;
}

// The FP32 pointwise kernels' weights (conv1x1_pw.hip / conv1x1_pwk.hip; kernel variant 14): the bf16 planes of d_w3 once more in MFMA
// A-fragment order - [64-channel block][16-row tile][32-deep slab][plane][lane] x 8 bf16; row = lane & 15 is output channel
// block * 64 + 16 tile + row, element j of k-group kg = lane >> 4 is input channel 32 slab + (j < 4 ? 4 kg + j : 16 + 4 kg + j - 4).
// Packed when first asked for (from the device's d_w3: one download + one upload, init-time work) and again after
// saber_hip_conv2d_autotune released them for an op that selected another family (api_autotune.hip).
bool pw_eligible(const saber_hip_conv* op) {
    const saber_hip_conv_desc& d = op->d;
    return op->algo == ALGO_IGEMM_F32 && op->d_w3.p != nullptr && d.kh == 1 && d.kw == 1 && d.pad_h == 0 && d.pad_w == 0 && d.stride_h == 1 &&
           d.stride_w == 1 && d.dil_h == 1 && d.dil_w == 1 && d.group == 1 && !op->pre_transpose && d.out_layout == SABER_HIP_NHWC &&
           !op->pair_k2 && !op->pool2 &&
           (conv1x1_pw_ok(op->c_eff, d.k) || conv1x1_pwk_ok(d.n * d.h * d.w, op->c_eff, d.k)) &&
           (d.res_mode == SABER_HIP_RES_NONE || d.res_mode == SABER_HIP_RES_SUM_INPLACE);
}
int pw_prepare(saber_hip_conv* op) {
    if (op->d_wpw.p) return SABER_HIP_OK;
    if (!pw_eligible(op)) return SABER_HIP_INVALID_VALUE;
    const int K = op->d.k, Ce = op->c_eff, NS = Ce / 32;
    const size_t n = op->d_w3.n / 6;                        // elements per plane: K_pad x Kg_pad
    std::vector<uint8_t> planes(op->d_w3.n);
    HIP_TRY(hipMemcpy(planes.data(), op->d_w3.p, planes.size(), hipMemcpyDeviceToHost));
    const uint16_t* pl = (const uint16_t*)planes.data();
    std::vector<uint8_t> fr((size_t)(K / 64) * 4 * NS * 3 * 64 * 16, 0);
    uint16_t* fp = (uint16_t*)fr.data();
    for (int kb = 0; kb < K / 64; ++kb)
        for (int i = 0; i < 4; ++i)
            for (int sl = 0; sl < NS; ++sl)
                for (int lane = 0; lane < 64; ++lane) {
                    const int r = lane & 15, kg = lane >> 4;
                    const int ch = kb * 64 + i * 16 + r;
                    for (int j = 0; j < 8; ++j) {
                        const int kk = sl * 32 + (j < 4 ? kg * 4 + j : 16 + kg * 4 + (j - 4));
                        const size_t src = (size_t)ch * op->Kg_pad + kk;
                        for (int pl3 = 0; pl3 < 3; ++pl3)
                            fp[(((((size_t)kb * 4 + i) * NS + sl) * 3 + pl3) * 64 + lane) * 8 + j] = pl[pl3 * n + src];
                    }
                }
    HIP_TRY(op->d_wpw.upload(fr));
    return SABER_HIP_OK;
}

int saber_hip_conv2d_get_quantized_weights(const saber_hip_conv_t* op, int8_t* wq, float* ws) {
    if (!op->is_i8 || !op->weights_set) return fail(SABER_HIP_INVALID_VALUE, "no quantised weights");
    if (wq) std::memcpy(wq, op->wq_oihw.data(), op->wq_oihw.size());
    if (ws) std::memcpy(ws, op->w_scale.data(), sizeof(float) * op->w_scale.size());
    return SABER_HIP_OK;
}

static void fill_args(const saber_hip_conv* op, ConvKArgs& a, const void* x, void* y, const void* res,
                      void* y2 = nullptr) {
    const saber_hip_conv_desc& d = op->d;
    std::memset(&a, 0, sizeof a);
    a.x = x;
    a.w = op->d_w.p;
    a.y = y;
    a.res = res;
    a.bias = (op->has_bias || !op->is_i8) ? op->d_bias.p : nullptr;
    if (!op->is_i8 && !op->has_bias) a.bias = nullptr;
    a.scale = op->d_scale.p;
    a.comp = op->has_comp ? op->d_comp.p : nullptr;
    a.zero = zero_page();
    a.N = d.n; a.H = d.h; a.W = d.w; a.C = op->c_eff; a.K = d.k; a.OH = op->oh; a.OW = op->ow;
    a.kh = d.kh; a.kw = d.kw; a.pad_h = d.pad_h; a.pad_w = d.pad_w;
    a.stride_h = d.stride_h; a.stride_w = d.stride_w; a.dil_h = d.dil_h; a.dil_w = d.dil_w;
    a.M = d.n * op->oh * op->ow;
    a.Kg = op->Kg; a.Kg_pad = op->Kg_pad; a.kw_pad = op->kw_pad;
    const int estage = op->b3 ? 32 * op->ks : (op->algo == ALGO_IGEMM_F32 ? 16 : 64) * op->ks * (op->dma > 1 ? op->dma : 1);   // elements per stage
    if (op->b3) {
        a.w = op->d_w3.p;
        a.w_plane_chunks = (int)((size_t)round_up(d.k, 128) * op->Kg_pad / 8);
        a.ksplit_sh = op->ksplit;
        a.part = op->d_part.p;
        a.part_ctr = op->d_part_ctr.p;
        a.part_err = op->h_part_err;      // (unified addressing: the pinned word's host pointer is its device pointer)
    }
    a.steps = (op->Kg + estage - 1) / estage;
    a.inv_ohw = 1.0f / (float)(op->oh * op->ow);
    a.inv_ow = 1.0f / (float)op->ow;
    a.in_u8 = op->x_dtype == DT_U8;
    a.out_dtype = d.out_dtype;
    a.out_nchw = (!op->is_i8 && d.out_layout == SABER_HIP_NCHW) ? 1 : 0;
    a.relu = d.act == SABER_HIP_ACT_RELU;
    a.neg_slope = d.act_negative_slope;
    a.epi = op->epi;
    a.res_mode = d.res_mode;
    a.res_relu = d.res_act == SABER_HIP_ACT_RELU;
    a.res_dtype = d.res_has_dtype ? d.res_dtype : d.out_dtype;
    a.sum_scale = d.sum_scale;
    a.coeff_conv = d.coeff_conv; a.coeff_res = d.coeff_res;
    a.scale_conv = op->out_scale; a.scale_res = d.scale_res;
    if (op->pool2) { a.pool_oh = op->pool_oh; a.pool_ow = op->pool_ow; }
    if (d.res_stride > 1) { a.res_sub = d.res_stride; a.res_H = d.res_h; a.res_W = d.res_w; }
    a.y2 = y2;
    a.K1 = op->pair_k1; a.K2 = op->pair_k2; a.relu2 = op->pair_relu2; a.out_dtype2 = op->pair_dtype2;
    if (!op->is_i8 && d.res_mode == SABER_HIP_RES_SUM_INPLACE) {
        // out = act(conv + bias + y): the activation belongs to the eltwise when fused
        a.relu = (d.res_act == SABER_HIP_ACT_RELU) || (d.act == SABER_HIP_ACT_RELU);
    }
}

void conv_fill_args(const saber_hip_conv* op, saber_mi355x::ConvKArgs& a, const void* x, void* y, const void* res) { fill_args(op, a, x, y, res); }

// the fused stem conv + pooling's argument block (and its channel-padding pre-pass); shared with saber_hip_conv2d_stem_pair_run
int stem_pool_args(const saber_hip_conv* op, const void* x, void* y, void* workspace, hipStream_t s, saber_mi355x::ConvKArgs* a) {
    const saber_hip_conv_desc& d = op->d;
    const void* xin = x;
    if (op->pre_pad) {
        HIP_TRY(launch_pad_channels_i8((size_t)d.n * d.h * d.w, d.c, 4, x, workspace, s));
        xin = workspace;
    }
    fill_args(op, *a, xin, y, nullptr);
    a->pool_oh = op->pool_oh; a->pool_ow = op->pool_ow;
    a->Cin = d.c;
    a->qinv = 1.f / op->in_scale;
    return SABER_HIP_OK;
}

int saber_hip_conv2d_run(saber_hip_conv_t* op, const void* x, void* y, const void* res, void* workspace,
                         saber_hip_stream_t stream) {
    if (g_capture) return capture_conv(op, x, y, res);
    if (!op || !x || !y) return fail(SABER_HIP_INVALID_VALUE, "null argument");
    if (!op->weights_set) return fail(SABER_HIP_INVALID_VALUE, "set_weights not called");
    if (op->ws_bytes && !workspace) return fail(SABER_HIP_INVALID_VALUE, "workspace required");
    if (op->d.res_mode == SABER_HIP_RES_ELTWISE && !res) return fail(SABER_HIP_INVALID_VALUE, "residual tensor required");
    if (op->pair_k2) return fail(SABER_HIP_INVALID_VALUE, "sibling pair: use saber_hip_conv2d_run_pair");
    hipStream_t s = (hipStream_t)stream;
    const saber_hip_conv_desc& d = op->d;
    const void* xin = x;
    if (op->gpool) return fail(SABER_HIP_INVALID_VALUE, "conv + fused global pooling: use saber_hip_conv2d_run_gpool");
    if (op->img1) return img_conv_run(op, x, y, res, nullptr, s);
    if (op->stem32) {
        HIP_TRY(launch_conv_stem_f32_pool_raw((const float*)x, op->d_wstem32.p, op->has_bias ? op->d_bias.p : nullptr, (float*)y, d.n, d.h, d.w, op->oh,
                                              op->ow, op->pool_oh, op->pool_ow, op->stem32 - 1, s));
        return SABER_HIP_OK;
    }
    if (op->pool_fused) {
        ConvKArgs a;
        const int rc = stem_pool_args(op, x, y, workspace, s, &a);
        if (rc) return rc;
        HIP_TRY(launch_conv_stem_pool(op->pre_quant ? 1 : 0, a, s));
        return SABER_HIP_OK;
    }
    if (op->stem && op->pre_quant) {
        // fused: the stem kernel reads the f32 NCHW image and quantises while staging its LDS patch
        ConvKArgs a;
        fill_args(op, a, x, y, res);
        a.Cin = d.c;
        a.qinv = 1.f / op->in_scale;
        HIP_TRY(launch_conv_stem(1, a, s));
        return SABER_HIP_OK;
    }
    if (op->pre_quant) {
        HIP_TRY(launch_quantize_nchw_to_nhwc(d.n, d.c, d.h, d.w, op->c_eff, DT_S8, op->in_scale, (const float*)x,
                                             workspace, s));
        xin = workspace;
    } else if (op->pre_pad) {
        HIP_TRY(launch_pad_channels_i8((size_t)d.n * d.h * d.w, d.c, 4, x, workspace, s));
        xin = workspace;
    } else if (op->pre_transpose) {
        HIP_TRY(launch_transpose_nchw_to_nhwc_f32(d.n, d.c, d.h, d.w, op->c_eff, (const float*)x, (float*)workspace, s));
        xin = workspace;
    }
    ConvKArgs a;
    fill_args(op, a, xin, y, res);
    switch (op->algo) {
    case ALGO_IGEMM_I8:
        if (op->fc_small) {
            HIP_TRY(launch_fc_i8_small(a, s));
            break;
        }
        if (op->img_rb) {
            HIP_TRY(launch_conv3x3_img(a, op->img_nw, op->img_ib, op->img_rb, s));
            break;
        }
        if (op->halo) {
            HIP_TRY(launch_conv3x3_halo(op->halo, a, s));
            break;
        }
        HIP_TRY(op->dma ? launch_conv_igemm_dma(0, op->tile, op->ks, op->dma, a, s) : launch_conv_igemm(0, op->tile, op->ks, a, s));
        break;
    case ALGO_IGEMM_I8_C4:
        if (op->stem) HIP_TRY(launch_conv_stem(0, a, s));
        else HIP_TRY(launch_conv_igemm(1, op->tile, op->ks, a, s));
        break;
    case ALGO_IGEMM_F32:
        if (op->fc_small) {
            if (op->d_fcpart.p && !op->d_wfc.p) {      // few output tiles: the reduction split over workgroups (fc_f32_splitk.hip)
                HIP_TRY(launch_fc_f32_splitk(a, op->d_fcpart.p, op->d_fcctr.p, nullptr, s));
                break;
            }
            if (op->d_wfc.p) a.w = op->d_wfc.p;
            HIP_TRY(launch_fc_f32_small(a, op->d_wfc.p != nullptr, s));
            break;
        }
        if (op->pw) {
            a.w = op->d_wpw.p;
            if (op->pw > 1) HIP_TRY(launch_conv1x1_pwk(op->pw - 1, a, s));
            else HIP_TRY(launch_conv1x1_pw(a, s));
            break;
        }
        if (op->b3h) {
            int hb, ht, htm, hthr;
            (void)conv3x3_b3h_variant(op->b3h, &hb, &ht, &htm, &hthr);
            a.w = htm == 1 ? op->d_w3h1.p : op->d_w3h2.p;
            HIP_TRY(launch_conv3x3_b3h(op->b3h, a, s));
            break;
        }
        if (op->b3 && op->ksplit && !op->d_part.p) return fail(SABER_HIP_INVALID_VALUE, "split-K selected without its buffers (saber_hip_conv2d_set_tile / autotune allocate them)");
        if (op->b3 && op->ksplit) {
            const int rs = split_check(op);
            if (rs) return rs;
        }
        if (op->b3) HIP_TRY(launch_conv_igemm(3, op->tile, op->ks, a, s));
        else HIP_TRY(op->dma ? launch_conv_igemm_dma(2, op->tile, op->ks, op->dma, a, s) : launch_conv_igemm(2, op->tile, op->ks, a, s));
        break;
    case ALGO_DIRECT_I8:
        a.comp = nullptr;
        HIP_TRY(launch_conv_direct(0, a, d.group, s));
        break;
    case ALGO_DIRECT_F32: HIP_TRY(launch_conv_direct(1, a, d.group, s)); break;
    default: return fail(SABER_HIP_UNIMPL, "no algorithm");
    }
    return SABER_HIP_OK;
}


// ------------------------------------------------------------------------------------------------
// sibling pair: two INT8 convs over one input in one launch (see saber_hip.h)
// ------------------------------------------------------------------------------------------------
int saber_hip_conv2d_create_pair(const saber_hip_conv_t* a, const saber_hip_conv_t* b, saber_hip_conv_t** out) {
    if (!a || !b || !out) return fail(SABER_HIP_INVALID_VALUE, "null argument");
    const saber_hip_conv_desc &da = a->d, &db = b->d;
    auto plain = [](const saber_hip_conv* o) {
        if (!o->weights_set || o->d.res_mode != SABER_HIP_RES_NONE || o->pair_k2 || o->pool_fused || o->pool2) return false;
        if (o->is_i8)
            return o->algo == ALGO_IGEMM_I8 && o->epi == EPI_I8_CONV && !o->pre_quant && !o->pre_pad &&
                   (o->d.out_dtype == SABER_HIP_S8 || o->d.out_dtype == SABER_HIP_U8);
        return o->algo == ALGO_IGEMM_F32 && !o->pre_transpose && o->d.out_layout == SABER_HIP_NHWC;   // FP32: NHWC in / out
    };
    if (!plain(a) || !plain(b) || a->is_i8 != b->is_i8)
        return fail(SABER_HIP_INVALID_VALUE, "pair: both ops must be plain implicit-GEMM convs of one precision (INT8 with 8-bit NHWC "
                                              "outputs, or FP32 NHWC) with weights set");
    if (da.n != db.n || da.h != db.h || da.w != db.w || da.c != db.c || da.kh != db.kh || da.kw != db.kw ||
        da.pad_h != db.pad_h || da.pad_w != db.pad_w || da.stride_h != db.stride_h || da.stride_w != db.stride_w ||
        da.dil_h != db.dil_h || da.dil_w != db.dil_w || da.in_dtype != db.in_dtype || a->Kg_pad != b->Kg_pad)
        return fail(SABER_HIP_INVALID_VALUE, "pair: the two convs must share the input tensor and geometry");
    if (da.k % 128 || db.k % 16) return fail(SABER_HIP_INVALID_VALUE, "pair: needs a.k % 128 == 0 and b.k % 16 == 0");
    auto* op = new saber_hip_conv();
    op->d = da;
    op->d.k = da.k + db.k;
    op->oh = a->oh; op->ow = a->ow;
    op->algo = a->algo;
    op->epi = a->epi;
    op->is_i8 = a->is_i8;
    op->x_dtype = a->x_dtype;
    op->c_eff = a->c_eff;
    op->Kg = a->Kg; op->Kg_pad = a->Kg_pad;
    op->in_scale = a->in_scale; op->out_scale = a->out_scale;
    op->pair_k1 = da.k; op->pair_k2 = db.k;
    op->pair_src_a = a; op->pair_src_b = b;
    op->pair_relu2 = db.act == SABER_HIP_ACT_RELU;
    op->pair_dtype2 = db.out_dtype;
    const size_t k2_pad = round_up(db.k, 128), rows = (size_t)da.k + k2_pad;
    auto cat = [&](auto& dst, const auto& sa, const auto& sb, size_t per_row) -> hipError_t {
        hipError_t e = dst.alloc_zero(rows * per_row);
        if (e != hipSuccess) return e;
        typedef typename std::remove_reference<decltype(*dst.p)>::type T;
        if (sa.p) e = hipMemcpy(dst.p, sa.p, (size_t)da.k * per_row * sizeof(T), hipMemcpyDeviceToDevice);
        if (e != hipSuccess) return e;
        if (sb.p) e = hipMemcpy(dst.p + (size_t)da.k * per_row, sb.p, k2_pad * per_row * sizeof(T), hipMemcpyDeviceToDevice);
        return e;
    };
    hipError_t e = cat(op->d_w, a->d_w, b->d_w, (size_t)a->Kg_pad * (a->is_i8 ? 1 : sizeof(float)));
    if (e == hipSuccess && !a->is_i8 && a->d_w3.p && b->d_w3.p) {   // bf16-plane variant: [3][rows][Kg_pad] from the ops' planes
        const size_t row = (size_t)a->Kg_pad * 2, pa = (size_t)round_up(da.k, 128) * row, pb = k2_pad * row;
        e = op->d_w3.alloc_zero(3 * rows * row);
        for (int pl = 0; pl < 3 && e == hipSuccess; ++pl) {
            e = hipMemcpy(op->d_w3.p + pl * rows * row, a->d_w3.p + pl * pa, (size_t)da.k * row, hipMemcpyDeviceToDevice);
            if (e == hipSuccess) e = hipMemcpy(op->d_w3.p + pl * rows * row + (size_t)da.k * row, b->d_w3.p + pl * pb, pb, hipMemcpyDeviceToDevice);
        }
    }
    if (e == hipSuccess) e = cat(op->d_bias, a->d_bias, b->d_bias, 1);
    if (e == hipSuccess && a->is_i8) e = cat(op->d_scale, a->d_scale, b->d_scale, 1);
    op->has_bias = a->has_bias || b->has_bias;
    op->has_comp = a->has_comp;   // same input dtype -> both or neither
    if (e == hipSuccess && op->has_comp) e = cat(op->d_comp, a->d_comp, b->d_comp, 1);
    if (e != hipSuccess) {
        delete op;
        return hip_fail(e, "pair: device copies");
    }
    op->weights_set = true;
    choose_tile(op);
    {
        const int kbytes = op->Kg * (op->is_i8 ? 1 : 4);
        op->ks = kbytes >= 256 ? 4 : (kbytes >= 128 ? 2 : 1);
    }
    name_algo(op);
    *out = op;
    return SABER_HIP_OK;
}

int saber_hip_conv2d_run_pair(saber_hip_conv_t* op, const void* x, void* y_a, void* y_b, saber_hip_stream_t stream) {
    if (g_capture) return capture_unsupported("saber_hip_conv2d_run_pair (an executor-level object: saber_hip_net_optimize forms pairs itself)");
    if (!op || !x || !y_a || !y_b) return fail(SABER_HIP_INVALID_VALUE, "null argument");
    if (!op->pair_k2) return fail(SABER_HIP_INVALID_VALUE, "not a sibling pair");
    ConvKArgs a;
    fill_args(op, a, x, y_a, nullptr, y_b);
    hipStream_t s = (hipStream_t)stream;
    const int mode = op->is_i8 ? 0 : (op->b3 ? 3 : 2);
    if (op->b3 && op->ksplit && !op->d_part.p) return fail(SABER_HIP_INVALID_VALUE, "split-K selected without its buffers (autotune / set_tile allocate them)");
    if (op->b3 && op->ksplit) {
        const int rs = split_check(op);
        if (rs) return rs;
    }
    HIP_TRY(op->dma && !op->b3 ? launch_conv_igemm_dma(mode, op->tile, op->ks, op->dma, a, s) : launch_conv_igemm(mode, op->tile, op->ks, a, s));
    return SABER_HIP_OK;
}

int saber_hip_conv2d_autotune_pair(saber_hip_conv_t* op, const void* x, void* y_a, void* y_b, saber_hip_stream_t stream,
                                   int iters) {
    if (!op || !op->pair_k2) return fail(SABER_HIP_INVALID_VALUE, "not a sibling pair");
    hipStream_t s = (hipStream_t)stream;
    EventPair ev;
    HIP_TRY(ev.init());
    ColdScope scope;
    HIP_TRY(scope.enter(7));
    std::vector<std::pair<float, ConvChoice>> pcands;
    float best = 1e30f;
    ConvChoice best_c = get_choice(op);   // the entry selection stays if nothing runs
    auto time_current = [&]() {
        if (g_cold) {
            const float us = g_cold->run(s, [&] { return saber_hip_conv2d_run_pair(op, x, y_a, y_b, s); });
            if (us >= 0.f) pcands.emplace_back(us, get_choice(op));
            if (us >= 0.f && us < best) { best = us; best_c = get_choice(op); }
            return;
        }
        int rc = saber_hip_conv2d_run_pair(op, x, y_a, y_b, s);
        if (rc) return;   // a variant that does not launch is skipped
        float ms = 0;
        if (hipEventRecord(ev.e0, s) != hipSuccess) return;
        for (int i = 0; i < iters; ++i) rc |= saber_hip_conv2d_run_pair(op, x, y_a, y_b, s);
        if (rc || hipEventRecord(ev.e1, s) != hipSuccess || hipEventSynchronize(ev.e1) != hipSuccess ||
            hipEventElapsedTime(&ms, ev.e0, ev.e1) != hipSuccess)
            return;
        if (ms < best) { best = ms; best_c = get_choice(op); }
    };
    const int ks_list[3] = {1, 2, 4};
    const int dma_list[4] = {0, 1, 2, 4};
    for (int vi = 0; vi < 4; ++vi)
        for (int t = 0; t < TILE_COUNT; ++t)
            for (int ki = 0; ki < 3; ++ki) {
                if (dma_list[vi] > 1 && (ks_list[ki] != 4 || t > TILE_64x64)) continue;
                if (dma_list[vi] == 4 && t != TILE_32x32) continue;
                ConvChoice c = {t, ks_list[ki], dma_list[vi], 0, 0, 0, 0, 4, 0, 0, 0};
                set_choice(op, c);
                time_current();
            }
    if (b3_ok(op))      // FP32 pair on the bf16 matrix cores, with split-K where the reduction is deep and the pixels few
        for (int kd = 1; kd <= 2; ++kd)
            for (int t = 0; t < TILE_COUNT_B3; ++t) {
                if (!b3_tile_ok(op, t, kd)) continue;
                for (int sh = 0; sh <= 3; ++sh) {
                    if (sh && (!split_ok(op, t, kd, sh) || split_prepare(op) != SABER_HIP_OK)) continue;
                    int bmk, bnp;
                    tile_dims(t, &bmk, &bnp);
                    const long tiles = (long)((op->d.n * op->oh * op->ow + bnp - 1) / bnp) * ((op->d.k + bmk - 1) / bmk);
                    if (sh && (tiles << sh) > 2048) continue;
                    ConvChoice c = {t, kd, 0, 0, 0, 0, 0, 4, 0, 1, sh};
                    set_choice(op, c);
                    time_current();
                }
            }
    set_choice(op, best_c);
    if (g_used_kernels && best < 1e30f) {   // kernel reuse across the net's sibling pairs (see kernel_key)
        float reuse_best = best * (1.f + g_reuse_tol);
        for (const auto& cd : pcands) {
            const unsigned long long key = kernel_key(op, cd.second);
            if (cd.first <= reuse_best && std::find(g_used_kernels->begin(), g_used_kernels->end(), key) != g_used_kernels->end()) {
                reuse_best = cd.first;
                set_choice(op, cd.second);
            }
        }
        g_used_kernels->push_back(kernel_key(op, get_choice(op)));
    }
    name_algo(op);
    if (!op->ksplit) {
        op->d_part.release();
        op->d_part_ctr.release();
    }
    return saber_hip_conv2d_run_pair(op, x, y_a, y_b, s);   // both outputs hold the selected kernel's result
}

int saber_hip_conv2d_set_global_pooling(saber_hip_conv_t* op) {
    if (!op || !op->weights_set) return fail(SABER_HIP_INVALID_VALUE, "set_weights first");
    if (op->img_stage) img_conv_release(op);
    op->gpool = 1;
    const int rc = img_conv_prepare(op);
    if (rc) {
        op->gpool = 0;
        return fail(SABER_HIP_UNIMPL, "conv + global average pooling: no fused kernel for this op (run the two ops)");
    }
    op->img1 = 1; op->halo = 0; op->img_ib = op->img_rb = 0; op->stem = 0; op->fc_small = 0;
    name_algo(op);
    return SABER_HIP_OK;
}
int saber_hip_conv2d_run_gpool(saber_hip_conv_t* op, const void* x, void* y, const void* res, void* y_pool, saber_hip_stream_t stream) {
    if (g_capture) return capture_unsupported("saber_hip_conv2d_run_gpool");
    if (!op || !op->gpool) return fail(SABER_HIP_INVALID_VALUE, "not a conv with fused global pooling");
    return img_conv_run(op, x, y, res, y_pool, (hipStream_t)stream);
}

void saber_hip_conv2d_destroy(saber_hip_conv_t* op) {
    if (op) img_conv_release(op);
    if (op && op->h_part_err) (void)hipHostFree(op->h_part_err);
    delete op;
}

