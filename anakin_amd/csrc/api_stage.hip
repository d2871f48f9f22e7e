// anakin_amd/csrc/api_stage.hip - saber_hip_stage_*: the XCD-resident stage (stage_xcd.hip): checks, weight repacking, launch.
#include "api_internal.h"

struct saber_hip_stage {
    std::vector<saber_hip_conv*> convs;
    std::vector<StagePhase> phases;
    DevBuf<uint8_t> d_phases, d_w, d_prm;
    DevBuf<unsigned long long> d_sync, d_trace;
    int n_barriers = 0, n_tensors = 0, n_img = 0, h = 0, w = 0;
    size_t lds_bytes = 0;
};

namespace {
constexpr int kSyncWords = 16 * 17 + 16;     // 8 registration + 8 arrival counters and the abort flag, one 128-byte line each

bool stage_conv_ok(const saber_hip_conv* o) {
    const saber_hip_conv_desc& d = o->d;
    const bool k1 = d.kh == 1 && d.kw == 1 && d.pad_h == 0 && d.pad_w == 0;
    const bool k3 = d.kh == 3 && d.kw == 3 && d.pad_h == 1 && d.pad_w == 1;
    const bool res_ok = d.res_mode == SABER_HIP_RES_NONE ||
                        (d.res_mode == SABER_HIP_RES_ELTWISE && d.out_dtype == SABER_HIP_S8 &&
                         (!d.res_has_dtype || d.res_dtype == SABER_HIP_S8) && d.res_stride <= 1);
    return o->is_i8 && o->weights_set && o->algo == ALGO_IGEMM_I8 && o->epi == EPI_I8_CONV && (k1 || k3) && d.stride_h == 1 &&
           d.stride_w == 1 && d.dil_h == 1 && d.dil_w == 1 && d.group == 1 && !o->pair_k2 && !o->pool_fused && !o->pool2 &&
           !o->pre_quant && !o->pre_pad && o->c_eff == d.c && d.act_negative_slope == 0.f && d.in_layout == SABER_HIP_NHWC &&
           d.out_layout == SABER_HIP_NHWC && (d.out_dtype == SABER_HIP_S8 || d.out_dtype == SABER_HIP_U8) && res_ok &&
           d.c % 64 == 0 && d.k % 16 == 0 && o->oh == d.h && o->ow == d.w;
}

// [cu 32][wave 4][k-step of the wave][tile][lane 64][16 bytes]: row = lane & 15 -> channel (cu * nt + tile) * 16 + row,
// k-group = lane >> 4; 3x3: k-steps ordered [tap][c / 64]
void pack_stage_weights(const saber_hip_conv* o, int nt, int kq, std::vector<uint8_t>& out) {
    const saber_hip_conv_desc& d = o->d;
    const int C = d.c, K = d.k, taps = d.kh * d.kw, kspt = C / 64;
    const int8_t* w = o->wq_oihw.data();
    for (int cu = 0; cu < 32; ++cu)
        for (int wave = 0; wave < 4; ++wave)
            for (int i = 0; i < kq; ++i)
                for (int j = 0; j < nt; ++j)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int ch = (cu * nt + j) * 16 + (lane & 15), ks = wave * kq + i;
                        const int tap = ks / kspt, c0 = (ks % kspt) * 64 + (lane >> 4) * 16;
                        for (int t = 0; t < 16; ++t)
                            out.push_back(ch < K ? (uint8_t)w[((size_t)ch * C + c0 + t) * taps + tap] : (uint8_t)0);
                    }
}
}  // namespace

namespace {
struct StageSpec {
    saber_hip_conv* conv;
    int in, out, res;
    int pool;              // >= 0: slot of the fused global average pooling's output
};
int stage_build(const StageSpec* ph, int n, bool xcd_resident, saber_hip_stage_t** out);
}  // namespace

int saber_hip_stage_create(const saber_hip_stage_phase* ph, int n, saber_hip_stage_t** out) {
    if (!ph || !out || n <= 0 || n > 64) return fail(SABER_HIP_INVALID_VALUE, "stage: bad argument");
    int dev = 0;
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDevice(&dev));
    HIP_TRY(hipGetDeviceProperties(&prop, dev));
    if (prop.multiProcessorCount != 256) return fail(SABER_HIP_UNIMPL, "stage: needs the 8 x 32 CU partition");
    std::vector<StageSpec> sp(n);
    for (int i = 0; i < n; ++i) sp[i] = {ph[i].conv, ph[i].in, ph[i].out, ph[i].res, -1};
    return stage_build(sp.data(), n, true, out);
}

namespace {
int stage_build(const StageSpec* ph, int n, bool xcd_resident, saber_hip_stage_t** out) {
    std::unique_ptr<saber_hip_stage> st(new saber_hip_stage());
    std::vector<uint8_t> wbytes, pbytes;
    std::vector<char> dirty(STAGE_MAX_TENSORS, 0), written(STAGE_MAX_TENSORS, 0);
    int prev_in = -1;
    size_t lds_chunks = 0;
    for (int i = 0; i < n; ++i) {
        saber_hip_conv* o = ph[i].conv;
        if (!o || !stage_conv_ok(o)) return fail(SABER_HIP_INVALID_VALUE, "stage: phase " + std::to_string(i) + " is not a plain 1x1 / 3x3 stride-1 INT8 NHWC conv with weights set");
        const saber_hip_conv_desc& d = o->d;
        const bool elt = d.res_mode == SABER_HIP_RES_ELTWISE;
        if (i == 0) { st->n_img = d.n; st->h = d.h; st->w = d.w; }
        if (d.n != st->n_img || d.h != st->h || d.w != st->w || d.h * d.w > 64) return fail(SABER_HIP_INVALID_VALUE, "stage: all phases run on the same n x h x w, h * w <= 64");
        const int slots[4] = {ph[i].in, ph[i].out, elt ? ph[i].res : 0, ph[i].pool >= 0 ? ph[i].pool : 0};
        for (int s : slots)
            if (s < 0 || s >= STAGE_MAX_TENSORS) return fail(SABER_HIP_INVALID_VALUE, "stage: tensor slot out of range");
        if (ph[i].out == ph[i].in || (elt && ph[i].out == ph[i].res) || written[ph[i].out])
            return fail(SABER_HIP_INVALID_VALUE, "stage: every phase writes a slot of its own (no in-place, no rewrite)");
        const int taps = d.kh * d.kw, ksteps = d.c * taps / 64;
        const int nt = (d.k / 16 + 31) / 32;
        StagePhase p;
        std::memset(&p, 0, sizeof p);
        if (ksteps % 4 || !stage_xcd_type(nt, ksteps / 4, taps == 9, &p.type))
            return fail(SABER_HIP_INVALID_VALUE, "stage: no kernel variant for " + std::to_string(d.c) + " -> " + std::to_string(d.k) + (taps == 9 ? " 3x3" : " 1x1"));
        p.cin = d.c; p.cout = d.k;
        p.in_t = ph[i].in; p.out_t = ph[i].out; p.res_t = elt ? ph[i].res : -1;
        p.in_u8 = o->x_dtype == DT_U8;
        p.relu = d.act == SABER_HIP_ACT_RELU;
        p.out_u8 = d.out_dtype == SABER_HIP_U8;
        p.elt = elt;
        p.res_relu = d.res_act == SABER_HIP_ACT_RELU;
        p.coeff_conv = d.coeff_conv; p.scale_conv = o->out_scale; p.coeff_res = d.coeff_res; p.scale_res = d.scale_res;
        p.barrier = dirty[p.in_t] || (elt && dirty[p.res_t]);
        if (p.barrier) {
            std::fill(dirty.begin(), dirty.end(), 0);
            st->n_barriers += 1;
        }
        p.reload = (p.barrier || p.in_t != prev_in || i == 0) ? 1 : 0;
        prev_in = p.in_t;
        dirty[p.out_t] = written[p.out_t] = 1;
        p.pch = (unsigned)(d.c / 16 + 1);
        p.mg_pch = (unsigned)((0x100000000ull + p.pch - 1) / p.pch);
        const size_t a64 = ((size_t)(d.h * d.w + 1) * p.pch + 63) / 64 * 64;
        p.red_chunk = (int)a64;
        lds_chunks = std::max(lds_chunks, a64 + (size_t)4 * nt * 4 * 64 + (size_t)4 * nt * 4);   // image, partial accumulators, pooling partials
        p.pool_t = ph[i].pool;
        p.pool_idiv = 1.0f / (float)(d.h * d.w);
        p.w_chunk = (unsigned)(wbytes.size() / 16);
        pack_stage_weights(o, nt, ksteps / 4, wbytes);
        // per 4 channels {scale[4], bias'[4], comp[4]}, padded to the 32 CUs' tiles
        p.prm_chunk = (unsigned)(pbytes.size() / 16);
        const int kpad = 32 * nt * 16;
        const size_t base = pbytes.size();
        pbytes.resize(base + (size_t)kpad / 4 * 48, 0);
        for (int k = 0; k < d.k; ++k) {
            uint8_t* q = pbytes.data() + base + (size_t)(k / 4) * 48;
            ((float*)q)[k % 4] = o->scale_host.empty() ? 1.f : o->scale_host[k];
            ((float*)q)[4 + k % 4] = (o->has_bias && !o->bias_p_host.empty()) ? o->bias_p_host[k] : 0.f;
            ((int*)q)[8 + k % 4] = o->comp_host.empty() ? 0 : o->comp_host[k];
        }
        st->n_tensors = std::max(st->n_tensors, std::max(std::max(p.in_t, p.pool_t), std::max(p.out_t, p.res_t)) + 1);
        st->phases.push_back(p);
        st->convs.push_back(o);
    }
    st->lds_bytes = xcd_resident ? std::max<size_t>(lds_chunks * 16, 81 * 1024) : lds_chunks * 16;     // stage: > 80 KB = one workgroup per CU
    if (st->lds_bytes > 160 * 1024 - 64) return fail(SABER_HIP_UNIMPL, "stage: the image does not fit in LDS");
    std::vector<uint8_t> pt((const uint8_t*)st->phases.data(), (const uint8_t*)st->phases.data() + st->phases.size() * sizeof(StagePhase));
    hipError_t e = st->d_phases.upload(pt);
    if (e == hipSuccess) e = st->d_w.upload(wbytes);
    if (e == hipSuccess) e = st->d_prm.upload(pbytes);
    if (e == hipSuccess && xcd_resident) e = st->d_sync.alloc_zero(kSyncWords);
    if (e != hipSuccess) return hip_fail(e, "stage: device copies");
    if (!zero_page()) return fail(SABER_HIP_RUNTIME_ERROR, "stage: zero page");
    *out = st.release();
    return SABER_HIP_OK;
}
}  // namespace

// ---- the same phase code as an ordinary kernel for ONE conv (a kernel variant of saber_hip_conv2d_run: image-resident) ----------
bool img_conv_ok(const saber_hip_conv* op) {
    if (!stage_conv_ok(op) || op->d.h * op->d.w > 64) return false;
    const int taps = op->d.kh * op->d.kw, ksteps = op->d.c * taps / 64;
    int type = 0;
    return ksteps % 4 == 0 && stage_xcd_type((op->d.k / 16 + 31) / 32, ksteps / 4, taps == 9, &type);
}
int img_conv_prepare(saber_hip_conv* op) {
    if (op->img_stage) return SABER_HIP_OK;
    if (!img_conv_ok(op)) return fail(SABER_HIP_INVALID_VALUE, "image-resident kernel: 1x1 / 3x3 stride-1 INT8 conv on <= 64 pixels per image with ResNet res5 channel shapes");
    const StageSpec sp = {op, 0, 1, op->d.res_mode == SABER_HIP_RES_ELTWISE ? 2 : -1, op->gpool ? 3 : -1};
    return stage_build(&sp, 1, false, &op->img_stage);
}
int img_conv_run(saber_hip_conv* op, const void* x, void* y, const void* res, void* y_pool, hipStream_t stream) {
    saber_hip_stage* st = op->img_stage;
    if (!st) return fail(SABER_HIP_INVALID_VALUE, "image-resident kernel selected without its buffers (set_tile / autotune build them)");
    const StagePhase& p = st->phases[0];
    if (!x || !y || (p.elt && !res) || (p.pool_t >= 0 && !y_pool)) return fail(SABER_HIP_INVALID_VALUE, "null tensor");
    StageKArgs k;
    std::memset(&k, 0, sizeof k);
    k.phases = (const StagePhase*)st->d_phases.p;
    k.weights = st->d_w.p; k.prm = st->d_prm.p; k.zero = zero_page();
    k.n_phases = 1;
    k.n_img = st->n_img; k.H = st->h; k.W = st->w;
    k.t[0] = const_cast<void*>(x); k.t[1] = y; k.t[2] = const_cast<void*>(res); k.t[3] = y_pool;
    HIP_TRY(launch_img_conv(k, st->lds_bytes, stream));
    return SABER_HIP_OK;
}
void img_conv_release(saber_hip_conv* op) {
    delete op->img_stage;
    op->img_stage = nullptr;
}

int saber_hip_stage_num_tensors(const saber_hip_stage_t* st) { return st ? st->n_tensors : 0; }

int saber_hip_stage_run(saber_hip_stage_t* st, void* const* tensors, int n_tensors, saber_hip_stream_t stream) {
    if (g_capture) return capture_unsupported("saber_hip_stage_run");
    if (!st || !tensors || n_tensors < st->n_tensors) return fail(SABER_HIP_INVALID_VALUE, "stage: tensor table too short");
    StageKArgs k;
    std::memset(&k, 0, sizeof k);
    k.phases = (const StagePhase*)st->d_phases.p;
    k.weights = st->d_w.p; k.prm = st->d_prm.p; k.zero = zero_page();
    k.sync = st->d_sync.p;
    k.n_phases = (int)st->phases.size(); k.n_barriers = st->n_barriers;
    k.n_img = st->n_img; k.H = st->h; k.W = st->w;
    k.trace = st->d_trace.p;
    for (int i = 0; i < st->n_tensors; ++i) k.t[i] = tensors[i];
    for (const StagePhase& p : st->phases)
        if (!k.t[p.in_t] || !k.t[p.out_t] || (p.elt && !k.t[p.res_t])) return fail(SABER_HIP_INVALID_VALUE, "stage: null tensor");
    HIP_TRY(launch_stage_xcd(k, st->lds_bytes, (hipStream_t)stream));
    return SABER_HIP_OK;
}

int saber_hip_stage_status(saber_hip_stage_t* st) {
    if (!st) return fail(SABER_HIP_INVALID_VALUE, "null argument");
    HIP_TRY(hipDeviceSynchronize());
    unsigned long long flag = 0;
    HIP_TRY(hipMemcpy(&flag, st->d_sync.p + 16 * 16, sizeof flag, hipMemcpyDeviceToHost));
    if (!flag) return SABER_HIP_OK;
    HIP_TRY(hipMemset(st->d_sync.p, 0, kSyncWords * sizeof(unsigned long long)));   // re-arm: counters back in step
    HIP_TRY(hipDeviceSynchronize());
    return fail(SABER_HIP_RUNTIME_ERROR, "stage: a workgroup gave up waiting for its XCD (the launch did not have the CUs to itself)");
}

int saber_hip_stage_trace(saber_hip_stage_t* st, unsigned long long* out, size_t cap) {
    if (!st) return fail(SABER_HIP_INVALID_VALUE, "null argument");
    const size_t n = (size_t)256 * st->phases.size() * 8;
    if (!out) {                       // arm: the following launches record their stamps
        HIP_TRY(st->d_trace.alloc_zero(n));
        return (int)n;
    }
    if (!st->d_trace.p || cap < n) return fail(SABER_HIP_INVALID_VALUE, "stage trace: not armed / buffer too short");
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, st->d_trace.p, n * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return (int)n;
}

void saber_hip_stage_destroy(saber_hip_stage_t* st) { delete st; }
