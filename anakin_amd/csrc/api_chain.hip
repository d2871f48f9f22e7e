// anakin_amd/csrc/api_chain.hip - conv1x1 chains: two (or conv3x3 + two) INT8 convolutions in one launch (conv1x1_chain.hip).
#include "api_internal.h"

// ------------------------------------------------------------------------------------------------
// conv1x1 chain: `a` (1x1, fused SaberEltwise epilogue, s8 out) feeding `b` (1x1, s8 / u8 out) in one launch
// ------------------------------------------------------------------------------------------------
static bool chain_1x1(const saber_hip_conv* o, bool sub_res_ok = false) {
    const saber_hip_conv_desc& d = o->d;
    return o->is_i8 && o->weights_set && !o->gpool && o->algo == ALGO_IGEMM_I8 && o->epi == EPI_I8_CONV && d.kh == 1 && d.kw == 1 &&
           d.stride_h == 1 && d.stride_w == 1 && d.pad_h == 0 && d.pad_w == 0 && d.group == 1 && !o->pair_k2 &&
           !o->pool_fused && !o->pool2 && !o->pre_quant && !o->pre_pad && o->c_eff == d.c && d.act_negative_slope == 0.f &&
           d.in_layout == SABER_HIP_NHWC && d.out_layout == SABER_HIP_NHWC && (d.res_stride <= 1 || sub_res_ok);
}
static void pack_chain_params(const saber_hip_conv* o, size_t chunks_pad, std::vector<uint8_t>& out) {
    const int K = o->d.k;
    out.assign(chunks_pad * 16, 0);
    for (int k4 = 0; k4 < K / 4; ++k4) {
        float* f = (float*)(out.data() + (size_t)k4 * 48);
        int* ip = (int*)(out.data() + (size_t)k4 * 48 + 32);
        for (int r = 0; r < 4; ++r) {
            const int k = k4 * 4 + r;
            f[r] = o->scale_host.empty() ? 1.f : o->scale_host[k];
            f[4 + r] = (o->has_bias && !o->bias_p_host.empty()) ? o->bias_p_host[k] : 0.f;
            ip[r] = o->comp_host.empty() ? 0 : o->comp_host[k];
        }
    }
}
// one conv's weights [K][C] -> per wave, groups of 16*mfg channels, steps ordered [group][k-step][accumulator], each step
// = 64 lanes x 16 bytes in MFMA A-operand order (row = lane & 15, k-group = lane >> 4); row rho of accumulator mf is
// channel  base + (rho >> 2) * 4*mfg + mf*4 + (rho & 3)   (conv1x1_chain.hip)
static void pack_chain_weights(const int8_t* w, int K, int C, int mfg, int wave, std::vector<uint8_t>& out, int kbase = 0,
                               int nw = 4, int ks0 = 0) {
    // K: channels of this workgroup's share (rows kbase .. kbase + K - 1 of w), nw waves; ks0: the k-step the stream starts with
    // (cooperative form: the k-steps over this workgroup's OWN half of the input channels come first)
    const int kw = K / nw, groups = kw / (16 * mfg), ksn = C / 64;
    for (int g = 0; g < groups; ++g)
        for (int ksi = 0; ksi < ksn; ++ksi)
            for (int mf = 0; mf < mfg; ++mf) {
                const int ks = (ks0 + ksi) % ksn;
                for (int lane = 0; lane < 64; ++lane) {
                    const int rho = lane & 15, kq = lane >> 4;
                    const int ch = kbase + wave * kw + g * 16 * mfg + (rho >> 2) * 4 * mfg + mf * 4 + (rho & 3);
                    const int8_t* src = w + (size_t)ch * C + ks * 64 + kq * 16;
                    out.insert(out.end(), (const uint8_t*)src, (const uint8_t*)src + 16);
                }
            }
}
// the 3x3 conv's weights [K][C][3][3] -> per wave, steps ordered [tap][k-step][accumulator] (conv1x1_chain.hip phase 0)
static void pack_chain_weights3(const int8_t* w, int C, int wave, std::vector<uint8_t>& out, int nw = 4, int kbase = 0, int kcount = 0) {
    // kcount > 0: this workgroup's share of the output channels (rows kbase .. kbase + kcount - 1)
    const int kw = (kcount ? kcount : C) / nw, mf0 = kw / 16, ksn = C / 64;
    for (int tap = 0; tap < 9; ++tap)
        for (int ks = 0; ks < ksn; ++ks)
            for (int mf = 0; mf < mf0; ++mf)
                for (int lane = 0; lane < 64; ++lane) {
                    const int rho = lane & 15, kq = lane >> 4;
                    const int ch = kbase + wave * kw + (rho >> 2) * 4 * mf0 + mf * 4 + (rho & 3);
                    for (int t = 0; t < 16; ++t) {
                        const int c = ks * 64 + kq * 16 + t;
                        out.push_back((uint8_t)w[((size_t)ch * C + c) * 9 + tap]);
                    }
                }
}
// one 1 KB MFMA A-operand fragment (64 lanes x 16 bytes: row = lane & 15, k-group = lane >> 4) of a 1x1 conv's weights [K][C]: rows =
// channels ch_base + (rho >> 2) * 4*mfg + mf*4 + (rho & 3), input channels ks*64 .. +63; and of a 3x3 conv's [K][C][3][3] at one tap
static void pack_frag1(const int8_t* w, int C, int ch_base, int mfg, int mf, int ks, std::vector<uint8_t>& out) {
    for (int lane = 0; lane < 64; ++lane) {
        const int rho = lane & 15, kq = lane >> 4;
        const int8_t* src = w + (size_t)(ch_base + (rho >> 2) * 4 * mfg + mf * 4 + (rho & 3)) * C + ks * 64 + kq * 16;
        out.insert(out.end(), (const uint8_t*)src, (const uint8_t*)src + 16);
    }
}
static void pack_frag3(const int8_t* w, int C, int ch_base, int tap, int ks, std::vector<uint8_t>& out) {
    for (int lane = 0; lane < 64; ++lane) {
        const int rho = lane & 15, kq = lane >> 4;
        for (int t = 0; t < 16; ++t) out.push_back((uint8_t)w[((size_t)(ch_base + rho) * C + ks * 64 + kq * 16 + t) * 9 + tap]);
    }
}
// the four-workgroup cooperative form's streams (conv_chain_coop.hip: conv_chain_coop4_c256_kernel), C = 256: per (quarter q, wave w =
// K-half kh * 4 + n-tile nt), in consumption order:
//   3x3, channels q*64 + nt*16 .. +15:        [tap][k-step kh*2 + {0, 1}]                                        18 fragments
//   1x1 + eltwise, channels q*256 + w*32 ..:  [k-step (q + i) % 4, i = 0..3][accumulator 0, 1]                    8 (its own mid k-step first)
//   1x1, channels q*64 + nt*16 .. +15:        k-steps (q*4 + j) % 16, j = kh*2 + {0, 1} then 4 + kh*6 + {0..5}    8 (its own y1 channels first)
static void pack_coop4_stream(const saber_hip_conv* c3, const saber_hip_conv* a, const saber_hip_conv* b, std::vector<uint8_t>& out) {
    const int C = 256, K1 = 1024;
    for (int q = 0; q < 4; ++q)
        for (int w = 0; w < 8; ++w) {
            const int nt = w & 3, kh = w >> 2;
            for (int tap = 0; tap < 9; ++tap)
                for (int kl = 0; kl < 2; ++kl) pack_frag3(c3->wq_oihw.data(), C, q * 64 + nt * 16, tap, kh * 2 + kl, out);
            for (int i = 0; i < 4; ++i)
                for (int mf = 0; mf < 2; ++mf) pack_frag1(a->wq_oihw.data(), C, q * 256 + w * 32, 2, mf, (q + i) & 3, out);
            for (int jj = 0; jj < 8; ++jj) {
                const int j = jj < 2 ? kh * 2 + jj : 4 + kh * 6 + (jj - 2);
                pack_frag1(b->wq_oihw.data(), K1, q * 64 + nt * 16, 1, 0, (q * 4 + j) & 15, out);
            }
        }
}
// ... and the one-workgroup-per-tile stage kernel's (conv_stage1_c128_kernel), C = 128: per wave w, in consumption order:
//   3x3, channels w*16 .. +15: [tap][k-step 0, 1] (18) | 1x1 + eltwise, channels w*64 .. +63: [k-step 0, 1][accumulator 0..3] (8) |
//   1x1, channels w*16 .. +15: [k-step 0..7] (8)
static void pack_stage1_c128_stream(const saber_hip_conv* c3, const saber_hip_conv* a, const saber_hip_conv* b, std::vector<uint8_t>& out) {
    const int C = 128, K1 = 512;
    for (int w = 0; w < 8; ++w) {
        for (int tap = 0; tap < 9; ++tap)
            for (int ks = 0; ks < 2; ++ks) pack_frag3(c3->wq_oihw.data(), C, w * 16, tap, ks, out);
        for (int ks = 0; ks < 2; ++ks)
            for (int mf = 0; mf < 4; ++mf) pack_frag1(a->wq_oihw.data(), C, w * 64, 4, mf, ks, out);
        for (int ks = 0; ks < 8; ++ks) pack_frag1(b->wq_oihw.data(), K1, w * 16, 1, 0, ks, out);
    }
}
// ------------------------------------------------------------------------------------------------
// stem conv + max pooling + the sibling pair of 1x1 convs on the pooled tensor in ONE launch (conv_stem.h)
// ------------------------------------------------------------------------------------------------
int saber_hip_conv2d_stem_pair_create(saber_hip_conv_t* stem, const saber_hip_conv_t* a, const saber_hip_conv_t* b, saber_hip_stem_pair_t** out) {
    if (!stem || !a || !b || !out) return fail(SABER_HIP_INVALID_VALUE, "null argument");
    const saber_hip_conv_desc& ds = stem->d;
    if (!stem->pool_fused || !stem->weights_set || ds.k != 64 || (ds.out_dtype != SABER_HIP_U8 && ds.out_dtype != SABER_HIP_S8))
        return fail(SABER_HIP_INVALID_VALUE, "stem pair: the head must be the fused stem conv + max pooling with 64 output channels (8-bit)");
    auto plain1x1 = [&](const saber_hip_conv* o) {
        const saber_hip_conv_desc& d = o->d;
        return o->is_i8 && o->weights_set && o->algo == ALGO_IGEMM_I8 && o->epi == EPI_I8_CONV && d.res_mode == SABER_HIP_RES_NONE &&
               !o->pair_k2 && !o->pool_fused && !o->pool2 && !o->pre_quant && !o->pre_pad && d.kh == 1 && d.kw == 1 && d.stride_h == 1 &&
               d.stride_w == 1 && d.pad_h == 0 && d.pad_w == 0 && d.group == 1 && d.c == 64 && o->c_eff == 64 && d.k % 32 == 0 &&
               d.in_layout == SABER_HIP_NHWC && d.out_layout == SABER_HIP_NHWC && d.act_negative_slope == 0.f &&
               (d.out_dtype == SABER_HIP_S8 || d.out_dtype == SABER_HIP_U8) && d.n == ds.n && d.h == stem->pool_oh && d.w == stem->pool_ow &&
               (o->x_dtype == DT_U8) == (ds.out_dtype == SABER_HIP_U8) && (int)o->wq_oihw.size() == d.k * 64;
    };
    if (!plain1x1(a) || !plain1x1(b) || a->d.k + b->d.k > 320)
        return fail(SABER_HIP_INVALID_VALUE, "stem pair: two plain 1x1 / stride-1 INT8 convs (64 -> k, k % 32 == 0, k_a + k_b <= 320, 8-bit NHWC outputs) "
                                              "on the pooled tensor");
    auto* sp = new saber_hip_stem_pair();
    sp->stem = stem; sp->a = a; sp->b = b;
    const int K1 = a->d.k, Kt = K1 + b->d.k;
    std::vector<int8_t> wcat((size_t)Kt * 64);
    std::memcpy(wcat.data(), a->wq_oihw.data(), (size_t)K1 * 64);
    std::memcpy(wcat.data() + (size_t)K1 * 64, b->wq_oihw.data(), (size_t)b->d.k * 64);
    std::vector<uint8_t> w, pa, pb, prm(256 * 16, 0);
    w.reserve((size_t)Kt * 64);
    for (int g = 0; g < Kt / 32; ++g)
        for (int mf = 0; mf < 2; ++mf) pack_frag1(wcat.data(), 64, g * 32, 2, mf, 0, w);
    pack_chain_params(a, (size_t)K1 / 4 * 3, pa);
    pack_chain_params(b, (size_t)b->d.k / 4 * 3, pb);
    std::memcpy(prm.data(), pa.data(), pa.size());
    std::memcpy(prm.data() + pa.size(), pb.data(), pb.size());
    hipError_t e = sp->d_w.upload(w);
    if (e == hipSuccess) e = sp->d_prm.upload(prm);
    if (e != hipSuccess) {
        delete sp;
        return hip_fail(e, "stem pair: device copies");
    }
    *out = sp;
    return SABER_HIP_OK;
}
void saber_hip_conv2d_stem_pair_destroy(saber_hip_stem_pair_t* sp) { delete sp; }
int saber_hip_conv2d_stem_pair_run(saber_hip_stem_pair_t* sp, const void* x, void* y_pool, void* y_a, void* y_b, void* workspace,
                                   saber_hip_stream_t stream) {
    if (g_capture) return capture_unsupported("saber_hip_conv2d_stem_pair_run (an executor-level object: saber_hip_net_optimize forms it itself)");
    if (!sp || !x || !y_a || !y_b) return fail(SABER_HIP_INVALID_VALUE, "null argument");
    saber_hip_conv* op = sp->stem;
    if (op->ws_bytes && !workspace) return fail(SABER_HIP_INVALID_VALUE, "workspace required");
    saber_mi355x::StemPairKArgs ka;
    const int rc = stem_pool_args(op, x, y_pool, workspace, (hipStream_t)stream, &ka.c);
    if (rc) return rc;
    ka.t.w = sp->d_w.p; ka.t.prm = sp->d_prm.p; ka.t.y1 = y_a; ka.t.y2 = y_b;
    ka.t.K1 = sp->a->d.k; ka.t.K2 = sp->b->d.k;
    ka.t.relu1 = sp->a->d.act == SABER_HIP_ACT_RELU; ka.t.relu2 = sp->b->d.act == SABER_HIP_ACT_RELU;
    ka.t.u8_1 = sp->a->d.out_dtype == SABER_HIP_U8; ka.t.u8_2 = sp->b->d.out_dtype == SABER_HIP_U8;
    HIP_TRY(saber_mi355x::launch_conv_stem_pool_pair(op->pre_quant ? 1 : 0, ka, (hipStream_t)stream));
    return SABER_HIP_OK;
}
saber_hip_chain::~saber_hip_chain() {
    if (h_coop_err) (void)hipHostFree(h_coop_err);
    delete stage1;
}
// ------------------------------------------------------------------------------------------------
// stage: a run of 3x3-led C = 256 chains in ONE persistent launch (conv_stage_coop.hip)
// ------------------------------------------------------------------------------------------------
static unsigned stage_magic(int d) { return d >= 2 ? (unsigned)((0x100000000ull + (unsigned)d - 1) / (unsigned)d) : 0u; }
static int stage_build(saber_hip_chain* const* chains, int n, bool per_image, saber_hip_chain_stage** out) {
    if (!chains || n <= 0 || n > saber_mi355x::STAGE4_LONG || !out) return fail(SABER_HIP_INVALID_VALUE, "stage: 1..24 chains");
    if (n > 1 && !per_image) return fail(SABER_HIP_INVALID_VALUE, "stage: several blocks need an image per XCD");
    if (n < 2 && chains[0] && chains[0]->c1 == 128) return fail(SABER_HIP_INVALID_VALUE, "stage: at C = 128 a stage is at least two blocks");
    const saber_hip_chain* c0 = chains[0];
    std::vector<saber_mi355x::StageBlk> blk;
    for (int k = 0; k < n; ++k) {
        const saber_hip_chain* ch = chains[k];
        const uint8_t* stream = !ch ? nullptr : (ch->c1 == 256 ? ch->d_stream_coop4.p : (ch->c1 == 128 ? ch->d_stream_stage1.p : nullptr));
        if (!ch || ch->c1 != c0->c1 || !ch->c3 || !ch->b || !stream || ch->c3->d.stride_h != 1)
            return fail(SABER_HIP_INVALID_VALUE, "stage: every block must be a conv3x3 (stride 1) + conv1x1 + conv1x1 chain at C = 256 (or all at C = 128)");
        const saber_hip_conv_desc& da = ch->a->d;
        if (da.n != c0->a->d.n || da.h != c0->a->d.h || da.w != c0->a->d.w) return fail(SABER_HIP_INVALID_VALUE, "stage: blocks of one tensor shape");
        if (k && ((ch->c3->x_dtype == DT_U8) != (chains[k - 1]->b->d.out_dtype == SABER_HIP_U8)))
            return fail(SABER_HIP_INVALID_VALUE, "stage: a block's 3x3 conv reads what the previous block's last conv writes");
        saber_mi355x::StageBlk B;
        std::memset(&B, 0, sizeof B);
        B.wstream = stream; B.prm0 = ch->d_prm0.p; B.prm1 = ch->d_prm1.p; B.prm2 = ch->d_prm2.p;
        B.coeff_conv = da.coeff_conv; B.scale_conv = ch->a->out_scale; B.coeff_res = da.coeff_res; B.scale_res = da.scale_res;
        B.in0_u8 = ch->c3->x_dtype == DT_U8; B.relu0 = ch->c3->d.act == SABER_HIP_ACT_RELU;
        B.in_u8 = ch->a->x_dtype == DT_U8; B.relu1 = da.act == SABER_HIP_ACT_RELU; B.res_relu = da.res_act == SABER_HIP_ACT_RELU;
        B.relu2 = ch->b->d.act == SABER_HIP_ACT_RELU; B.out_u8_2 = ch->b->d.out_dtype == SABER_HIP_U8;
        blk.push_back(B);
    }
    const saber_hip_conv_desc& d0 = c0->a->d;
    saber_hip_chain_stage* st = new saber_hip_chain_stage();
    st->chains.assign(chains, chains + n);
    st->c1 = c0->c1;
    st->n = d0.n; st->h = d0.h; st->w = d0.w;
    st->tiles_x = (d0.w + 15) / 16;
    st->tiles_per_img = st->tiles_x * ((d0.h + 1) / 2);
    st->per_image = per_image;
    // every workgroup of an image must hold a CU of its XCD at once (one workgroup per CU: 235 - 250 VGPRs); C = 256: four workgroups per
    // tile, edge counters for one column tile; C = 128: one per tile, 1 / 2 / 4 column tiles (arrivals per edge: a power of two)
    const int wg_per_tile = st->c1 == 256 ? 4 : 1;
    const bool fits = d0.n <= 8 && st->tiles_per_img * wg_per_tile <= 32 &&
                      (st->c1 == 256 ? st->tiles_x == 1 : (st->tiles_x == 1 || st->tiles_x == 2 || st->tiles_x == 4));
    if ((per_image && !fits) || (st->c1 == 128 && !per_image)) {
        delete st;
        return fail(SABER_HIP_INVALID_VALUE, "stage: an image per XCD needs batch <= 8 and <= 32 workgroups per image (C = 256: width <= 16, <= 8 tiles "
                    "of 2 x 16 pixels; C = 128: width <= 64, <= 32 tiles)");
    }
    const size_t tiles = (size_t)d0.n * st->tiles_per_img;
    hipError_t e = st->d_blk.upload(blk);
    if (e == hipSuccess && st->c1 == 256) e = st->d_grp_ctr.alloc_zero(tiles * 32);
    if (e == hipSuccess) e = st->d_img_ctr.alloc_zero((size_t)d0.n * (st->tiles_per_img + 1) * 16);
    if (e == hipSuccess) e = st->d_xcc.alloc_zero(tiles * 32);
    if (e == hipSuccess && st->c1 == 256) e = st->d_xch.alloc_zero(tiles * 32 * 256);
    if (e == hipSuccess) e = hipHostMalloc((void**)&st->h_err, sizeof(unsigned), hipHostMallocMapped);
    if (e != hipSuccess) {
        delete st;
        return hip_fail(e, "stage: device buffers");
    }
    *st->h_err = 0u;
    *out = st;
    return SABER_HIP_OK;
}
template <int MAXB>
static int stage_launch(saber_hip_chain_stage* st, const void* x, const void* res, void* const* y1, void* const* y2, hipStream_t s) {
    saber_mi355x::Stage4KArgs<MAXB> k;
    std::memset(&k, 0, sizeof k);
    k.x = x; k.res = res; k.zero = zero_page();
    if (!k.zero) return fail(SABER_HIP_RUNTIME_ERROR, "stage: zero page");
    k.blk = st->d_blk.p; k.grp_ctr = st->d_grp_ctr.p; k.img_ctr = st->d_img_ctr.p; k.xch = st->d_xch.p; k.xcc = st->d_xcc.p; k.err = st->h_err;
    k.nblk = (int)st->chains.size(); k.N = st->n; k.H = st->h; k.W = st->w;
    k.tiles_x = st->tiles_x; k.tiles_per_img = st->tiles_per_img;
    k.mg_tiles_x = stage_magic(st->tiles_x); k.mg_tpi = stage_magic(st->tiles_per_img); k.mg_wpi = stage_magic(st->tiles_per_img * 4);
    k.per_image = st->per_image;
    for (int i = 0; i < k.nblk; ++i) { k.y1[i] = y1[i]; k.y2[i] = y2[i]; }
    HIP_TRY(st->c1 == 256 ? saber_mi355x::launch_conv_stage4(k, s) : saber_mi355x::launch_conv_stage1_c128(k, s));
    return SABER_HIP_OK;
}
int stage_run(saber_hip_chain_stage* st, const void* x, const void* res, void* const* y1, void* const* y2, hipStream_t s) {
    if (*(volatile unsigned*)st->h_err) {      // an earlier launch found cooperating workgroups on different XCDs or timed out in a barrier
        *(volatile unsigned*)st->h_err = 0u;
        return fail(SABER_HIP_RUNTIME_ERROR, "cooperative stage: an earlier launch's workgroups did not share an XCD or timed out at a barrier "
                    "(its outputs are not valid)");
    }
    return (int)st->chains.size() <= saber_mi355x::STAGE4_SHORT ? stage_launch<saber_mi355x::STAGE4_SHORT>(st, x, res, y1, y2, s)
                                                                : stage_launch<saber_mi355x::STAGE4_LONG>(st, x, res, y1, y2, s);
}
int saber_hip_conv2d_stage_create(saber_hip_chain_t* const* chains, int n, saber_hip_chain_stage_t** out) {
    if (!xcd_round_robin()) return fail(SABER_HIP_UNIMPL, "stage: this device does not place workgroup b on XCD b % 8");
    return stage_build(chains, n, true, out);
}
void saber_hip_conv2d_stage_destroy(saber_hip_chain_stage_t* st) { delete st; }
int saber_hip_conv2d_stage_run(saber_hip_chain_stage_t* st, const void* x, const void* res, void* const* y1, void* const* y2, saber_hip_stream_t stream) {
    if (g_capture) return capture_unsupported("saber_hip_conv2d_stage_run (saber_hip_net_optimize forms stages itself)");
    if (!st || !x || !res || !y1 || !y2) return fail(SABER_HIP_INVALID_VALUE, "null argument");
    for (size_t i = 0; i < st->chains.size(); ++i)
        if (!y1[i] || !y2[i]) return fail(SABER_HIP_INVALID_VALUE, "stage: an output pointer per block");
    return stage_run(st, x, res, y1, y2, (hipStream_t)stream);
}
static int chain_build(saber_hip_conv* c3, saber_hip_conv* a, saber_hip_conv* b, saber_hip_chain_t** out, saber_hip_conv* b2 = nullptr) {
    // b == nullptr (with c3): conv3x3 + first 1x1 conv only; b2 (with c3 / stride 2 and b): b and b2 are the sibling pair that reads a's output
    if (!a || !out || (!b && !c3) || (b2 && (!b || !c3))) return fail(SABER_HIP_INVALID_VALUE, "null argument");
    const bool pair2 = b2 != nullptr;
    // a sub-sampled shortcut (saber_hip_conv_desc::res_stride) is read by the conv3x3 + conv1x1 form with a strided head only
    const bool strided = c3 && (!b || pair2) && c3->d.stride_h == 2 && c3->d.stride_w == 2;
    if (pair2 && !strided) return fail(SABER_HIP_INVALID_VALUE, "chain: a sibling pair follows the strided head (conv3x3 / stride 2 + conv1x1 + eltwise) only");
    if (!chain_1x1(a, strided) || (b && !chain_1x1(b)) || (b2 && !chain_1x1(b2)))
        return fail(SABER_HIP_INVALID_VALUE, "chain: both ops must be plain 1x1 stride-1 INT8 NHWC convs with weights set");
    if (strided != (a->d.res_stride > 1) || (strided && a->d.res_stride != 2))
        return fail(SABER_HIP_INVALID_VALUE, "chain: a stride-2 head goes with a shortcut sub-sampled by 2 (and only with one)");
    const saber_hip_conv_desc& da = a->d;
    if (da.res_mode != SABER_HIP_RES_ELTWISE || da.out_dtype != SABER_HIP_S8 || (da.res_has_dtype && da.res_dtype != SABER_HIP_S8))
        return fail(SABER_HIP_INVALID_VALUE, "chain: the first conv must carry the fused eltwise epilogue with s8 residual and output");
    if (b) {
        const saber_hip_conv_desc& db = b->d;
        if (db.res_mode != SABER_HIP_RES_NONE || b->x_dtype != DT_S8 || (db.out_dtype != SABER_HIP_S8 && db.out_dtype != SABER_HIP_U8))
            return fail(SABER_HIP_INVALID_VALUE, "chain: the second conv must be a plain s8-input conv with an 8-bit output");
        if (db.n != da.n || db.h != a->oh || db.w != a->ow || db.c != da.k || (!pair2 && db.k != da.c))
            return fail(SABER_HIP_INVALID_VALUE, "chain: shapes must be C -> 4C -> C with C in {64,128,256,512} on the same pixels");
    }
    if (pair2) {
        const saber_hip_conv_desc &db = b->d, &d2 = b2->d;
        if (d2.res_mode != SABER_HIP_RES_NONE || b2->x_dtype != DT_S8 || (d2.out_dtype != SABER_HIP_S8 && d2.out_dtype != SABER_HIP_U8) || d2.n != da.n ||
            d2.h != a->oh || d2.w != a->ow || d2.c != da.k || da.c != 64 || db.k % 32 || db.k + d2.k != 640 || d2.k <= 0)
            return fail(SABER_HIP_INVALID_VALUE, "chain: the sibling pair after a strided head: 64 -> 256 (+ eltwise) -> k_b | k_b2 with k_b + k_b2 = 640, k_b % 32 == 0, 8-bit outputs");
    }
    if (!conv1x1_chain_ok(da.c, da.k, da.c))
        return fail(SABER_HIP_INVALID_VALUE, "chain: shapes must be C -> 4C -> C with C in {64,128,256,512} on the same pixels");
    if (c3) {
        const saber_hip_conv_desc& d3 = c3->d;
        const bool ok = c3->is_i8 && c3->weights_set && c3->algo == ALGO_IGEMM_I8 && c3->epi == EPI_I8_CONV && d3.kh == 3 && d3.kw == 3 &&
                        d3.stride_h == (strided ? 2 : 1) && d3.stride_w == d3.stride_h && d3.pad_h == 1 && d3.pad_w == 1 && d3.dil_h == 1 && d3.dil_w == 1 &&
                        d3.group == 1 && !c3->pair_k2 && !c3->pool_fused && !c3->pool2 && !c3->pre_quant && !c3->pre_pad &&
                        c3->c_eff == d3.c && d3.act_negative_slope == 0.f && d3.in_layout == SABER_HIP_NHWC &&
                        d3.out_layout == SABER_HIP_NHWC && d3.res_mode == SABER_HIP_RES_NONE && d3.c == da.c && d3.k == da.c &&
                        d3.n == da.n && c3->oh == da.h && c3->ow == da.w && da.c <= 256 &&
                        (d3.out_dtype == SABER_HIP_S8 || d3.out_dtype == SABER_HIP_U8) &&
                        (d3.out_dtype == SABER_HIP_U8) == (a->x_dtype == DT_U8);
        if (!ok) return fail(SABER_HIP_INVALID_VALUE, "chain: the head must be the 3x3 pad-1 INT8 conv (C -> C, C <= 256; stride 1, or 2 in front of a lone 1x1 conv) whose 8-bit output the first 1x1 conv reads");
    }
    saber_hip_chain* ch = new saber_hip_chain();
    const int k2 = b ? b->d.k + (pair2 ? b2->d.k : 0) : 0, c2 = b ? b->d.c : 0;
    ch->c3 = c3; ch->a = a; ch->b = b; ch->b2 = b2; ch->c1 = da.c; ch->k1 = da.k; ch->k2 = k2;
    ch->tn = conv1x1_chain_tn(da.c, da.n * a->oh * a->ow);
    const int mfg2 = pair2 ? 2 : ((k2 / 4) / 16 >= 4 ? 4 : (k2 / 4) / 16);      // (pair: 160 channels per wave = 5 groups of 32)
    std::vector<int8_t> wcat;       // the pair's rows one after the other
    if (pair2) {
        wcat.assign(b->wq_oihw.begin(), b->wq_oihw.end());
        wcat.insert(wcat.end(), b2->wq_oihw.begin(), b2->wq_oihw.end());
    }
    std::vector<uint8_t> stream, p0, p1, p2;
    stream.reserve((size_t)da.k * da.c + (size_t)k2 * c2 + (c3 ? (size_t)9 * da.c * da.c : 0));
    for (int w = 0; w < 4; ++w) {
        if (c3) pack_chain_weights3(c3->wq_oihw.data(), da.c, w, stream);
        pack_chain_weights(a->wq_oihw.data(), da.k, da.c, 4, w, stream);
        if (b) pack_chain_weights(pair2 ? wcat.data() : b->wq_oihw.data(), k2, c2, mfg2, w, stream);
    }
    pack_chain_params(a, (size_t)da.k / 4 * 3, p1);
    if (b) pack_chain_params(b, ((size_t)k2 / 4 * 3 + 63) / 64 * 64 + 64, p2);      // (+ 64 chunks of slack: the cooperative kernel
                                                                                     // DMAs whole 64-chunk blocks from a half's offset)
    if (pair2) {                    // b2's constants behind b's (48 bytes per 4 channels)
        std::vector<uint8_t> pb2;
        pack_chain_params(b2, (size_t)b2->d.k / 4 * 3, pb2);
        std::memcpy(p2.data() + (size_t)b->d.k / 4 * 48, pb2.data(), pb2.size());
    }
    hipError_t e = ch->d_stream.upload(stream);
    if (e == hipSuccess && !c3 && da.c >= 256) {
        std::vector<uint8_t> sp;
        sp.reserve(2 * (size_t)da.k * da.c + (size_t)k2 * c2);
        const int k2w = k2 / 2, mfgw = (k2w / 4) / 16 >= 4 ? 4 : (k2w / 4) / 16;
        for (int half = 0; half < 2; ++half)
            for (int w = 0; w < 4; ++w) {
                pack_chain_weights(a->wq_oihw.data(), da.k, da.c, 4, w, sp);
                pack_chain_weights(b->wq_oihw.data(), k2w, c2, mfgw, w, sp, half * k2w);
            }
        e = ch->d_stream_split.upload(sp);
        if (e == hipSuccess && da.c == 256) {   // 8 waves: 128 first-conv channels (2 groups) and 16 second-conv channels per wave
            std::vector<uint8_t> s8;
            s8.reserve(sp.size());
            for (int half = 0; half < 2; ++half)
                for (int w = 0; w < 8; ++w) {
                    pack_chain_weights(a->wq_oihw.data(), da.k, da.c, 4, w, s8, 0, 8);
                    pack_chain_weights(b->wq_oihw.data(), k2w, c2, 1, w, s8, half * k2w, 8);
                }
            e = ch->d_stream_split8.upload(s8);
        }
    }
    if (e == hipSuccess && (da.c == 128 || (da.c == 256 && c3))) {
        // 8 waves: C = 128: 16 channels of the 3x3 / 64 of the first / 16 of the second 1x1 conv per wave; C = 256 (3x3-led forms only):
        // 32 / 128 / 32 (two accumulators per group of the second conv)
        std::vector<uint8_t> s8;
        s8.reserve(stream.size());
        for (int w = 0; w < 8; ++w) {
            if (c3) pack_chain_weights3(c3->wq_oihw.data(), da.c, w, s8, 8);
            pack_chain_weights(a->wq_oihw.data(), da.k, da.c, 4, w, s8, 0, 8);
            if (b) pack_chain_weights(b->wq_oihw.data(), k2, c2, da.c == 128 ? 1 : 2, w, s8, 0, 8);
        }
        e = ch->d_stream_w8.upload(s8);
    }
    if (e == hipSuccess && da.c == 256 && c3 && b && c3->d.stride_h == 1 && xcd_round_robin()) {
        // cooperative form (conv_chain_coop.hip): per (half, wave) [3x3: 16 channels][1x1: 64 channels][1x1: 16 channels]; tiles of one
        // row x 16 columns; needs the workgroup -> XCD placement the hand-off relies on (probed once per device)
        std::vector<uint8_t> sc;
        sc.reserve(stream.size());
        for (int half = 0; half < 2; ++half)
            for (int w = 0; w < 8; ++w) {
                pack_chain_weights3(c3->wq_oihw.data(), da.c, w, sc, 8, half * (da.c / 2), da.c / 2);
                pack_chain_weights(a->wq_oihw.data(), da.k / 2, da.c, 4, w, sc, half * (da.k / 2), 8, half * (da.c / 128));
                pack_chain_weights(b->wq_oihw.data(), k2 / 2, c2, 1, w, sc, half * (k2 / 2), 8, half * (c2 / 128));
            }
        ch->coop_tiles = da.n * da.h * ((da.w + 15) / 16);
        e = ch->d_stream_coop.upload(sc);
        if (e == hipSuccess) {
            std::vector<uint8_t> s4;
            s4.reserve(stream.size());
            pack_coop4_stream(c3, a, b, s4);
            e = ch->d_stream_coop4.upload(s4);
        }
        if (e == hipSuccess) e = ch->d_coop_ctr.alloc_zero((size_t)ch->coop_tiles * 32);      // a 128-byte line per (tile, barrier)
        if (e == hipSuccess) e = ch->d_coop_xcc.alloc_zero((size_t)ch->coop_tiles * 32);
        if (e == hipSuccess) e = ch->d_coop_xch.alloc_zero((size_t)ch->coop_tiles * 16 * da.c);
        if (e == hipSuccess) e = hipHostMalloc((void**)&ch->h_coop_err, sizeof(unsigned), hipHostMallocMapped);
        if (e == hipSuccess) *ch->h_coop_err = 0u;
    }
    if (e == hipSuccess && da.c == 128 && c3 && b && c3->d.stride_h == 1 && xcd_round_robin()) {
        std::vector<uint8_t> s1;
        s1.reserve(stream.size());
        pack_stage1_c128_stream(c3, a, b, s1);
        e = ch->d_stream_stage1.upload(s1);
    }
    if (e == hipSuccess) e = ch->d_prm1.upload(p1);
    if (e == hipSuccess && b) e = ch->d_prm2.upload(p2);
    if (e == hipSuccess && c3) {
        pack_chain_params(c3, ((size_t)da.c / 4 * 3 + 63) / 64 * 64 + 64, p0);
        e = ch->d_prm0.upload(p0);
    }
    if (e != hipSuccess) {
        delete ch;
        return hip_fail(e, "chain: device copies");
    }
    if (ch->d_stream_coop4.p) {      // tile 15: the four-workgroup form = a one-block stage, tiles spread over all XCDs
        saber_hip_chain* one[1] = {ch};
        int rc = stage_build(one, 1, false, &ch->stage1);
        if (rc != SABER_HIP_OK) {
            delete ch;
            return rc;
        }
    }
    *out = ch;
    return SABER_HIP_OK;
}
int saber_hip_conv2d_chain_create(saber_hip_conv_t* a, saber_hip_conv_t* b, saber_hip_chain_t** out) {
    if (!b) return fail(SABER_HIP_INVALID_VALUE, "null argument");
    return chain_build(nullptr, a, b, out);
}
int saber_hip_conv2d_chain_create3(saber_hip_conv_t* conv3x3, saber_hip_conv_t* a, saber_hip_conv_t* b, saber_hip_chain_t** out) {
    if (!conv3x3) return fail(SABER_HIP_INVALID_VALUE, "null argument");
    return chain_build(conv3x3, a, b, out);
}
int saber_hip_conv2d_chain_create3_pair(saber_hip_conv_t* conv3x3, saber_hip_conv_t* a, saber_hip_conv_t* pair_a, saber_hip_conv_t* pair_b,
                                        saber_hip_chain_t** out) {
    if (!conv3x3 || !pair_a || !pair_b) return fail(SABER_HIP_INVALID_VALUE, "null argument");
    return chain_build(conv3x3, a, pair_a, out, pair_b);
}
void saber_hip_conv2d_chain_destroy(saber_hip_chain_t* ch) { delete ch; }
int saber_hip_conv2d_chain_set_tile(saber_hip_chain_t* ch, int tn) {
    if (!ch) return fail(SABER_HIP_INVALID_VALUE, "null argument");
    if (ch->b2 && tn != 4 && tn != 2) return fail(SABER_HIP_INVALID_VALUE, "chain: the strided head + pair form has 2 or 4 tile rows");
    const bool ok = (ch->c1 == 64 && (tn == 4 || tn == 2)) || (ch->c1 == 128 && (tn == 2 || tn == 1)) || (ch->c1 >= 256 && tn == 1) ||
                    (ch->c1 == 128 && (tn == 6 || tn == 5) && ch->d_stream_w8.p) || (ch->c1 == 256 && tn == 3 && ch->c3 && ch->d_stream_w8.p) ||
                    (ch->c1 >= 256 && tn == 9 && ch->d_stream_split.p && ch->b) || (tn == 11 && ch->d_stream_split8.p && ch->b) ||
                    (ch->c1 == 256 && tn == 7 && ch->d_stream_coop.p) || (ch->c1 == 256 && tn == 15 && ch->stage1);
    if (!ok) return fail(SABER_HIP_INVALID_VALUE, "chain: no kernel with that many pixel fragments");
    ch->tn = tn;
    return SABER_HIP_OK;
}
int saber_hip_conv2d_chain_get_tile(const saber_hip_chain_t* ch) { return ch ? ch->tn : 0; }
int saber_hip_conv2d_chain_run(saber_hip_chain_t* ch, const void* x, const void* res, void* y_a, void* y_b,
                               saber_hip_stream_t stream) {
    return saber_hip_conv2d_chain_run3(ch, x, res, y_a, y_b, nullptr, stream);
}
int saber_hip_conv2d_chain_run3(saber_hip_chain_t* ch, const void* x, const void* res, void* y_a, void* y_b, void* y_c,
                                saber_hip_stream_t stream) {
    if (g_capture) return capture_unsupported("saber_hip_conv2d_chain_run (saber_hip_net_optimize forms chains itself)");
    if (!ch || !x || !res || !y_a || (ch->b && !y_b) || (ch->b2 && !y_c)) return fail(SABER_HIP_INVALID_VALUE, "null argument");
    const saber_hip_conv* a = ch->a;
    const saber_hip_conv* b = ch->b;
    ChainKArgs k;
    std::memset(&k, 0, sizeof k);
    k.x = x; k.res = res;
    k.wstream = ch->tn == 11 ? ch->d_stream_split8.p : ((ch->tn & 8) ? ch->d_stream_split.p : ch->d_stream.p);
    if ((ch->c1 == 128 && (ch->tn & 4)) || (ch->c1 == 256 && ch->tn == 3)) k.wstream = ch->d_stream_w8.p;
    k.prm1 = ch->d_prm1.p;
    k.prm2 = ch->d_prm2.p;
    k.y1 = y_a; k.y2 = y_b;
    k.M = a->d.n * a->oh * a->ow;
    k.in_u8 = a->x_dtype == DT_U8;
    k.relu1 = a->d.act == SABER_HIP_ACT_RELU;
    k.res_relu = a->d.res_act == SABER_HIP_ACT_RELU;
    k.coeff_conv = a->d.coeff_conv; k.scale_conv = a->out_scale; k.coeff_res = a->d.coeff_res; k.scale_res = a->d.scale_res;
    if (b) {
        k.relu2 = b->d.act == SABER_HIP_ACT_RELU;
        k.out_u8_2 = b->d.out_dtype == SABER_HIP_U8;
    }
    if (ch->b2) {
        k.y2b = y_c;
        k.k2_split = b->d.k;
        k.relu2b = ch->b2->d.act == SABER_HIP_ACT_RELU;
        k.out_u8_2b = ch->b2->d.out_dtype == SABER_HIP_U8;
    }
    if (ch->c3) {   // x is the 3x3 conv's input; tiles of tn rows x 16 columns
        auto magic = [](int d) { return d >= 2 ? (unsigned)((0x100000000ull + (unsigned)d - 1) / (unsigned)d) : 0u; };
        k.prm0 = ch->d_prm0.p;
        k.zero = zero_page();
        if (!k.zero) return fail(SABER_HIP_RUNTIME_ERROR, "chain: zero page");
        k.N = a->d.n; k.H = a->d.h; k.W = a->d.w;
        k.tiles_x = (k.W + 15) / 16;
        // tile rows (C = 128: bit 2 of the code = 8 waves; C = 256: one row, code 3 = 8 waves, 7 = two cooperating workgroups; 15 = four, two rows)
        const int rows = ch->c1 == 128 ? ch->tn & 3 : (ch->c1 == 256 ? (ch->tn == 15 ? 2 : 1) : ch->tn & 7);
        k.tiles_per_img = k.tiles_x * ((k.H + rows - 1) / rows);
        k.mg_tiles_x = magic(k.tiles_x);
        k.mg_tpi = magic(k.tiles_per_img);
        k.in0_u8 = ch->c3->x_dtype == DT_U8;
        k.relu0 = ch->c3->d.act == SABER_HIP_ACT_RELU;
        k.s0 = ch->c3->d.stride_h;
        k.H0 = ch->c3->d.h; k.W0 = ch->c3->d.w;
        if (a->d.res_stride > 1) { k.res_sub = a->d.res_stride; k.res_H = a->d.res_h; k.res_W = a->d.res_w; }
    }
    if (ch->c1 == 256 && ch->tn == 15) {      // four cooperating workgroups per tile of 2 x 16 pixels: a one-block stage
        void* y1s[1] = {y_a};
        void* y2s[1] = {y_b};
        const int rc = stage_run(ch->stage1, x, res, y1s, y2s, (hipStream_t)stream);
        if (rc != SABER_HIP_OK) ch->tn = 3;
        return rc;
    }
    if (ch->c1 == 256 && ch->tn == 7) {      // two cooperating workgroups per pixel tile
        if (*(volatile unsigned*)ch->h_coop_err) {      // an earlier launch found its halves on different XCDs or timed out in a barrier
            *(volatile unsigned*)ch->h_coop_err = 0u;
            ch->tn = 3;
            return fail(SABER_HIP_RUNTIME_ERROR, "cooperative chain: an earlier launch's workgroup pairs did not share an XCD or timed out at their "
                        "barrier (its outputs are not valid); the chain now runs the single-workgroup form");
        }
        CoopKArgs ck;
        ck.c = k;
        ck.c.wstream = ch->d_stream_coop.p;
        ck.coop_ctr = ch->d_coop_ctr.p;
        ck.coop_xch = ch->d_coop_xch.p;
        ck.coop_xcc = ch->d_coop_xcc.p;
        ck.coop_err = ch->h_coop_err;
        ck.n_tiles = ch->coop_tiles;
        HIP_TRY(launch_conv_chain_coop(ck, (hipStream_t)stream));
        return SABER_HIP_OK;
    }
    HIP_TRY(launch_conv1x1_chain(k, ch->c1, ch->k1, ch->k2, ch->tn, ch->c3 ? 1 : 0, (hipStream_t)stream));
    return SABER_HIP_OK;
}

