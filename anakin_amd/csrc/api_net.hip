// anakin_amd/csrc/api_net.hip - the op-list executor (saber_hip_net_*): arena, two lanes, hipGraph capture / replay, timing.
#include "api_internal.h"

#include <atomic>
static std::atomic<int> g_coop_fallbacks_total{0};      // every net of the process (saber_hip_coop_fallbacks_total)

int net_launch(saber_hip_net* net, const NetOp& o, hipStream_t s) {
    auto T = [&](int id) -> void* { return net->ptr(id); };
    void* ws = net->arena + net->ws_off;
    switch (o.kind) {
    case OP_CONV:
        if (o.skip) return SABER_HIP_OK;      // written by the previous op's chain launch
        if (o.stage && o.use_stage) {         // this op's and the next stage_n - 1 blocks' chains in one persistent launch
            void* y1[saber_mi355x::STAGE4_LONG];
            void* y2[saber_mi355x::STAGE4_LONG];
            const NetOp* ops = &o;            // (the ops of a net are contiguous: block k's 3x3 conv is ops[3 * k])
            for (int k = 0; k < o.stage_n; ++k) {
                y1[k] = T(ops[3 * k].chain3_y1);
                y2[k] = T(ops[3 * k].chain3_y2);
            }
            const int rc = stage_run(o.stage, T(o.in), T(o.chain3_res), y1, y2, s);
            if (rc == SABER_HIP_RUNTIME_ERROR) {      // an earlier launch of it timed out (conv_stage_coop.hip): block by block from now on
                ++net->coop_fallbacks;
                ++g_coop_fallbacks_total;
                net_set_stage(net, (int)(&o - net->ops.data()), false);
                if (net->exec) {      // (the captured graph holds the stage launch)
                    (void)hipGraphExecDestroy(net->exec);
                    (void)hipGraphDestroy(net->graph);
                    net->exec = nullptr;
                    net->graph = nullptr;
                }
            }
            return rc;
        }
        if (o.stem_pair) return saber_hip_conv2d_stem_pair_run(o.stem_pair, T(o.in), nullptr, T(o.stem_y1), T(o.stem_y2), ws, s);
        if (o.chain3 && o.use_chain3) {
            const int rc = saber_hip_conv2d_chain_run3(o.chain3, T(o.in), T(o.chain3_res), T(o.chain3_y1), T(o.chain3_y2), T(o.chain3_y3), s);
            if (rc == SABER_HIP_RUNTIME_ERROR) {      // (a cooperating-workgroup chain reporting its failed earlier launch)
                ++net->coop_fallbacks;
                ++g_coop_fallbacks_total;
            }
            return rc;
        }
        if (o.chain && o.use_chain) return saber_hip_conv2d_chain_run(o.chain, T(o.in), T(o.in2), T(o.out), T(o.chain_out), s);
        if (o.conv->gpool) return saber_hip_conv2d_run_gpool(o.conv, T(o.in), T(o.out), T(o.in2), T(o.out2), s);
        return saber_hip_conv2d_run(o.conv, T(o.in), T(o.out), T(o.in2), ws, s);
    case OP_CONV_PAIR:
        if (o.skip) return SABER_HIP_OK;      // written by the stem launch in front of it (flag 512)
        return saber_hip_conv2d_run_pair(o.conv, T(o.in), T(o.out), T(o.out2), s);
    case OP_FC:
        if (o.out2 >= 0) return fc_run_softmax(o.fc, T(o.in), (float*)T(o.out), (float*)T(o.out2), ws, s, false);      // flag 4096
        return saber_hip_fc_run(o.fc, T(o.in), (float*)T(o.out), ws, s);
    case OP_QUANT:
        return saber_hip_quantize_nchw_to_nhwc(o.p[0], o.p[1], o.p[2], o.p[3], o.p[4], o.p[5], o.f[0],
                                               (const float*)T(o.in), T(o.out), s);
    case OP_DEQUANT:
        return saber_hip_dequantize_nhwc_to_nchw(o.p[0], o.p[1], o.p[2], o.p[3], o.p[4], o.f[0], T(o.in),
                                                 (float*)T(o.out), s);
    case OP_TRANSPOSE_IN:
        return saber_hip_transpose_nchw_to_nhwc_f32(o.p[0], o.p[1], o.p[2], o.p[3], o.p[4], (const float*)T(o.in),
                                                    (float*)T(o.out), s);
    case OP_ELT_I8:
        return saber_hip_eltwise_sum_i8(o.count, (const int8_t*)T(o.in), (const int8_t*)T(o.in2), o.f[0], o.f[1],
                                        o.f[2], o.f[3], o.p[0], (int8_t*)T(o.out), s);
    case OP_ELT_F32:
        return saber_hip_eltwise_sum_f32(o.count, (const float*)T(o.in), (const float*)T(o.in2), o.f[0], o.f[1],
                                         o.p[0], (float*)T(o.out), s);
    case OP_POOL_I8:
        return saber_hip_pool2d_i8_nhwc(o.p[0], o.p[1], o.p[2], o.p[3], o.p[4], o.p[5], o.p[6], o.p[7], o.p[8],
                                        o.p[9], o.p[10], o.p[11], o.p[12], o.p[13], o.p[14], T(o.in), T(o.out), s);
    case OP_POOL_F32:
        return saber_hip_pool2d_f32(o.p[0], o.p[1], o.p[2], o.p[3], o.p[4], o.p[5], o.p[6], o.p[7], o.p[8], o.p[9],
                                    o.p[10], o.p[11], o.p[12], o.p[13], (const float*)T(o.in), (float*)T(o.out), s);
    case OP_POOL_F32_I8:
        return saber_hip_pool2d_f32_from_i8_q(o.p[0], o.p[1], o.p[2], o.p[3], o.p[4], o.p[5], o.p[6], o.p[7], o.p[8],
                                              o.p[9], o.p[10], o.p[11], o.p[12], o.p[13], o.f[0], T(o.in),
                                              (float*)T(o.out), o.f[1], (int8_t*)T(o.out2), s);
    case OP_FC_Q:
        if (o.out2 >= 0) return fc_run_softmax(o.fc, T(o.in), (float*)T(o.out), (float*)T(o.out2), ws, s, true);
        return saber_hip_fc_run_q(o.fc, (const int8_t*)T(o.in), (float*)T(o.out), s);
    case OP_SOFTMAX:
        if (o.skip) return SABER_HIP_OK;      // normalised by the fc launch in front of it (flag 4096)
        return saber_hip_softmax_f32(o.p[0], o.p[1], (const float*)T(o.in), (float*)T(o.out), s);
    case OP_RELU_F32: return saber_hip_relu_f32(o.count, (const float*)T(o.in), (float*)T(o.out), s);
    case OP_ACT_F32: return saber_hip_activation_f32(o.p[0], o.count, o.f[0], o.f[1], (const float*)T(o.in), (float*)T(o.out), s);
    }
    return SABER_HIP_UNIMPL;
}


int saber_hip_net_create(saber_hip_net_t** out) {
    *out = new saber_hip_net();
    return SABER_HIP_OK;
}
int saber_hip_net_add_tensor(saber_hip_net_t* net, size_t bytes) {
    net->tensor_bytes.push_back(bytes);
    net->tensor_ext.push_back(nullptr);
    return (int)net->tensor_bytes.size() - 1;
}
int saber_hip_net_num_tensors(const saber_hip_net_t* net) { return net ? (int)net->tensor_bytes.size() : 0; }
size_t saber_hip_net_tensor_bytes(const saber_hip_net_t* net, int id) {
    return (net && id >= 0 && id < (int)net->tensor_bytes.size()) ? net->tensor_bytes[id] : 0;
}
int saber_hip_net_bind_tensor(saber_hip_net_t* net, int id, void* ptr) {
    if (!net || id < 0 || id >= (int)net->tensor_bytes.size()) return fail(SABER_HIP_INVALID_VALUE, "bad tensor id");
    if (net->finalized && !ptr && !net->tensor_ext[id]) return SABER_HIP_OK;
    if (net->finalized && !net->tensor_ext[id]) return fail(SABER_HIP_INVALID_VALUE, "bind_tensor: an arena tensor cannot become external after finalize");
    if (net->finalized && !ptr) return fail(SABER_HIP_INVALID_VALUE, "bind_tensor: an external tensor has no arena slot to fall back to after finalize");
    net->tensor_ext[id] = ptr;
    if (net->exec) {   // a captured hipGraph holds the old address
        (void)hipGraphExecDestroy(net->exec);
        (void)hipGraphDestroy(net->graph);
        net->exec = nullptr;
        net->graph = nullptr;
    }
    return SABER_HIP_OK;
}
int saber_hip_net_tensor_of_ptr(const saber_hip_net_t* net, const void* ptr) {
    if (!net) return -1;
    for (const auto& pr : net->captured_ptr)
        if (pr.first == ptr) return pr.second;
    return -1;
}
static int push(saber_hip_net* net, NetOp&& o) {
    const int nt = (int)net->tensor_bytes.size();
    if (o.in >= nt || o.in2 >= nt || o.out >= nt || o.in < 0 || o.out < 0) return fail(SABER_HIP_INVALID_VALUE, "bad tensor id");
    net->ops.push_back(std::move(o));
    return (int)net->ops.size() - 1;
}
int saber_hip_net_add_conv(saber_hip_net_t* net, saber_hip_conv_t* op, int in_id, int out_id, int res_id) {
    NetOp o;
    o.kind = OP_CONV; o.conv = op; o.in = in_id; o.out = out_id; o.in2 = res_id;
    o.name = std::string("conv:") + op->algo_name;
    if (op->ws_bytes > net->ws_bytes) net->ws_bytes = op->ws_bytes;
    return push(net, std::move(o));
}
int saber_hip_net_add_conv_pair(saber_hip_net_t* net, saber_hip_conv_t* op, int in_id, int out_a_id, int out_b_id) {
    if (!op || !op->pair_k2) return fail(SABER_HIP_INVALID_VALUE, "not a sibling pair");
    if (out_b_id < 0 || out_b_id >= (int)net->tensor_bytes.size()) return fail(SABER_HIP_INVALID_VALUE, "bad tensor id");
    NetOp o;
    o.kind = OP_CONV_PAIR; o.conv = op; o.in = in_id; o.out = out_a_id; o.out2 = out_b_id;
    o.name = std::string("conv:") + op->algo_name;
    return push(net, std::move(o));
}
int saber_hip_net_add_fc(saber_hip_net_t* net, saber_hip_fc_t* op, int in_id, int out_id) {
    NetOp o;
    o.kind = OP_FC; o.fc = op; o.in = in_id; o.out = out_id;
    o.name = std::string("fc:") + op->conv->algo_name;
    const size_t w = saber_hip_fc_workspace_bytes(op);
    if (w > net->ws_bytes) net->ws_bytes = w;
    return push(net, std::move(o));
}
int saber_hip_net_add_quantize(saber_hip_net_t* net, int n, int c, int h, int w, int c_pad, int out_dtype, float scale,
                               int in_id, int out_id) {
    NetOp o;
    o.kind = OP_QUANT; o.in = in_id; o.out = out_id; o.name = "quantize_nchw_to_nhwc";
    o.p[0] = n; o.p[1] = c; o.p[2] = h; o.p[3] = w; o.p[4] = c_pad; o.p[5] = out_dtype; o.f[0] = scale;
    return push(net, std::move(o));
}
int saber_hip_net_add_dequantize(saber_hip_net_t* net, int n, int c, int h, int w, int in_dtype, float scale,
                                 int in_id, int out_id) {
    NetOp o;
    o.kind = OP_DEQUANT; o.in = in_id; o.out = out_id; o.name = "dequantize_nhwc_to_nchw";
    o.p[0] = n; o.p[1] = c; o.p[2] = h; o.p[3] = w; o.p[4] = in_dtype; o.f[0] = scale;
    return push(net, std::move(o));
}
int saber_hip_net_add_transpose_in_f32(saber_hip_net_t* net, int n, int c, int h, int w, int c_pad, int in_id,
                                       int out_id) {
    NetOp o;
    o.kind = OP_TRANSPOSE_IN; o.in = in_id; o.out = out_id; o.name = "transpose_nchw_to_nhwc_f32";
    o.p[0] = n; o.p[1] = c; o.p[2] = h; o.p[3] = w; o.p[4] = c_pad;
    return push(net, std::move(o));
}
int saber_hip_net_add_eltwise_i8(saber_hip_net_t* net, size_t count, float sa, float sb, float c0, float c1, int relu,
                                 int a_id, int b_id, int out_id) {
    NetOp o;
    o.kind = OP_ELT_I8; o.in = a_id; o.in2 = b_id; o.out = out_id; o.count = count; o.name = "eltwise_sum_i8";
    o.f[0] = sa; o.f[1] = sb; o.f[2] = c0; o.f[3] = c1; o.p[0] = relu;
    return push(net, std::move(o));
}
int saber_hip_net_add_eltwise_f32(saber_hip_net_t* net, size_t count, float c0, float c1, int relu, int a_id, int b_id,
                                  int out_id) {
    NetOp o;
    o.kind = OP_ELT_F32; o.in = a_id; o.in2 = b_id; o.out = out_id; o.count = count; o.name = "eltwise_sum_f32";
    o.f[0] = c0; o.f[1] = c1; o.p[0] = relu;
    return push(net, std::move(o));
}
int saber_hip_net_add_pool_i8(saber_hip_net_t* net, int n, int h, int w, int c, int oh, int ow, int kh, int kw, int sh,
                              int sw, int ph, int pw, int type, int in_dtype, int out_dtype, int in_id, int out_id) {
    NetOp o;
    o.kind = OP_POOL_I8; o.in = in_id; o.out = out_id; o.name = "pool2d_i8_nhwc";
    const int v[15] = {n, h, w, c, oh, ow, kh, kw, sh, sw, ph, pw, type, in_dtype, out_dtype};
    std::memcpy(o.p, v, sizeof v);
    return push(net, std::move(o));
}
int saber_hip_net_add_pool_f32(saber_hip_net_t* net, int n, int h, int w, int c, int oh, int ow, int kh, int kw, int sh,
                               int sw, int ph, int pw, int type, int layout, int in_id, int out_id) {
    NetOp o;
    o.kind = OP_POOL_F32; o.in = in_id; o.out = out_id; o.name = "pool2d_f32";
    const int v[14] = {n, h, w, c, oh, ow, kh, kw, sh, sw, ph, pw, type, layout};
    std::memcpy(o.p, v, sizeof v);
    return push(net, std::move(o));
}
int saber_hip_net_add_pool_f32_from_i8(saber_hip_net_t* net, int n, int h, int w, int c, int oh, int ow, int kh, int kw,
                                       int sh, int sw, int ph, int pw, int type, int in_dtype, float scale, int in_id,
                                       int out_id) {
    NetOp o;
    o.kind = OP_POOL_F32_I8; o.in = in_id; o.out = out_id; o.name = "pool2d_f32_from_i8";
    const int v[14] = {n, h, w, c, oh, ow, kh, kw, sh, sw, ph, pw, type, in_dtype};
    std::memcpy(o.p, v, sizeof v);
    o.f[0] = scale;
    return push(net, std::move(o));
}
int saber_hip_net_add_pool_f32_from_i8_q(saber_hip_net_t* net, int n, int h, int w, int c, int oh, int ow, int kh, int kw,
                                         int sh, int sw, int ph, int pw, int type, int in_dtype, float scale, int in_id,
                                         int out_id, float q_scale, int q_out_id) {
    if (q_out_id < 0 || q_out_id >= (int)net->tensor_bytes.size()) return fail(SABER_HIP_INVALID_VALUE, "bad tensor id");
    int idx = saber_hip_net_add_pool_f32_from_i8(net, n, h, w, c, oh, ow, kh, kw, sh, sw, ph, pw, type, in_dtype, scale,
                                                 in_id, out_id);
    if (idx < 0) return idx;
    net->ops[idx].out2 = q_out_id;
    net->ops[idx].f[1] = q_scale;
    net->ops[idx].name = "pool2d_f32_from_i8+quantize";
    return idx;
}
int saber_hip_net_add_fc_q(saber_hip_net_t* net, saber_hip_fc_t* op, int in_q_id, int out_id) {
    NetOp o;
    o.kind = OP_FC_Q; o.fc = op; o.in = in_q_id; o.out = out_id;
    o.name = std::string("fc:") + op->conv->algo_name;
    return push(net, std::move(o));
}
int saber_hip_net_add_relu_f32(saber_hip_net_t* net, size_t count, int in_id, int out_id) {
    NetOp o;
    o.kind = OP_RELU_F32; o.in = in_id; o.out = out_id; o.count = count; o.name = "relu_f32";
    return push(net, std::move(o));
}
int saber_hip_net_add_activation_f32(saber_hip_net_t* net, int active, size_t count, float negative_slope, float coef, int in_id, int out_id) {
    NetOp o;
    o.kind = OP_ACT_F32; o.in = in_id; o.out = out_id; o.count = count; o.name = "activation_f32";
    o.p[0] = active; o.f[0] = negative_slope; o.f[1] = coef;
    return push(net, std::move(o));
}
int saber_hip_net_add_softmax(saber_hip_net_t* net, int rows, int cols, int in_id, int out_id) {
    NetOp o;
    o.kind = OP_SOFTMAX; o.in = in_id; o.out = out_id; o.name = "softmax_f32";
    o.p[0] = rows; o.p[1] = cols;
    return push(net, std::move(o));
}

int saber_hip_net_finalize(saber_hip_net_t* net) {
    if (net->finalized) return SABER_HIP_OK;
    size_t off = 0;
    net->tensor_off.resize(net->tensor_bytes.size());
    for (size_t i = 0; i < net->tensor_bytes.size(); ++i) {
        net->tensor_off[i] = off;
        if (net->tensor_ext[i]) continue;      // caller-owned storage: no arena slot
        off += (net->tensor_bytes[i] + 255) / 256 * 256;
    }
    net->ws_off = off;
    off += (net->ws_bytes + 255) / 256 * 256;
    net->arena_bytes = off ? off : 256;
    HIP_TRY(hipMalloc((void**)&net->arena, net->arena_bytes));
    HIP_TRY(hipMemset(net->arena, 0, net->arena_bytes));
    HIP_TRY(hipStreamSynchronize(nullptr));   // the kernels run on non-blocking streams: not ordered after null-stream work
    net->finalized = true;
    return SABER_HIP_OK;
}
// ---- lifetime aliasing of the arena (round 6; the role framework/graph/llvm/optimizer/memory_scheduler.cpp plays for Net<>) ------------
// finalize gives EVERY edge its own slot (~166 MB of activations for a batch-8 ResNet50 INT8 net): that is what lets a test read any
// edge after a pass and lets the autotuner re-run any op on the operands a whole pass left behind. A net that only serves needs the
// edges alive from their producer to their last reader: compact_arena re-lays the arena out with tensors of disjoint lifetimes sharing
// memory, allocates the smaller arena and frees the old one. Three nets in flight then fit the 256 MB Infinity Cache together
// (multi_stream, Worker<MI355X>: the reference's own MemoryScheduler already does this for Net<MI355X>'s tensors; the plan behind
// prediction() was the part that did not).
//   * time = the op index; ops that MAY run as one launch in some selection (a chain head + its follower, a 3x3-led chain = 3 ops, a
//     stage = 3 x stage_n ops, the stem + its sibling pair, fc + softmax) count as ONE step: every tensor any of them touches is live
//     through the whole group - a persistent launch's workgroups are not in step with each other, and the selection may change later;
//   * never aliased: caller-owned tensors, tensors no op writes (the pass's inputs), tensors no op reads (its outputs), `keep` ids;
//   * a net with a side lane keeps the full arena (ops of the two lanes overlap in time);
//   * placement: largest first, lowest offset that overlaps no already placed tensor whose lifetime intersects (256-byte slots).
// Returns SABER_HIP_OK; saber_hip_net_arena_bytes reports the new size; a captured hipGraph is dropped (it holds the old addresses);
// tensor pointers handed out earlier are invalid. Edges between kept tensors hold garbage after a pass: call it AFTER the autotuner
// and after the last test that reads intermediate edges.
int saber_hip_net_compact_arena(saber_hip_net_t* net, const int* keep, int n_keep) {
    if (!net || !net->finalized) return fail(SABER_HIP_INVALID_VALUE, "compact_arena: net not finalized");
    const int nt = (int)net->tensor_bytes.size(), nops = (int)net->ops.size();
    for (const NetOp& o : net->ops)
        if (o.lane) return SABER_HIP_OK;      // two lanes: lifetimes by op index do not order the lanes against each other
    std::vector<int> first(nt, nops), last(nt, -1), writes(nt, 0), reads(nt, 0);
    std::vector<char> pinned(nt, 0);
    for (int i = 0; i < n_keep; ++i)
        if (keep && keep[i] >= 0 && keep[i] < nt) pinned[keep[i]] = 1;
    // group extents: g_lo[i] .. g_hi[i] = the widest run of ops op i can be launched together with
    std::vector<int> g_lo(nops), g_hi(nops);
    for (int i = 0; i < nops; ++i) g_lo[i] = g_hi[i] = i;
    auto span = [&](int i, int n) {
        const int hi = std::min(nops - 1, i + n - 1);
        int lo = i;
        for (int j = i; j <= hi; ++j) lo = std::min(lo, g_lo[j]);
        int h2 = hi;
        for (int j = i; j <= hi; ++j) h2 = std::max(h2, g_hi[j]);
        for (int j = lo; j <= h2; ++j) { g_lo[j] = std::min(g_lo[j], lo); g_hi[j] = std::max(g_hi[j], h2); }
    };
    for (int i = 0; i < nops; ++i) {
        const NetOp& o = net->ops[i];
        if (o.stage) span(i, 3 * o.stage_n);
        if (o.chain3) span(i, 3);
        if (o.chain) span(i, 2);
        if (o.stem_pair) span(i, 2);
        if ((o.kind == OP_FC || o.kind == OP_FC_Q) && o.out2 >= 0) span(i, 2);
        if (i + 1 < nops && net->ops[i + 1].skip) span(i, 2);      // (whatever else made the follower silent)
    }
    for (int i = 0; i < nops; ++i) {
        const NetOp& o = net->ops[i];
        const int rd[] = {o.in, o.in2, o.chain3_res};
        const int wr[] = {o.out, o.out2, o.chain_out, o.chain3_y1, o.chain3_y2, o.chain3_y3, o.stem_y1, o.stem_y2};
        auto touch = [&](int t) {
            if (t < 0 || t >= nt) return;
            first[t] = std::min(first[t], g_lo[i]);
            last[t] = std::max(last[t], g_hi[i]);
        };
        for (int t : rd) { touch(t); if (t >= 0 && t < nt) ++reads[t]; }
        for (int t : wr) { touch(t); if (t >= 0 && t < nt) ++writes[t]; }
    }
    for (int t = 0; t < nt; ++t)
        if (net->tensor_ext[t] || !writes[t] || !reads[t] || last[t] < 0) pinned[t] = 1;      // external / input / output / untouched
    for (int t = 0; t < nt; ++t)
        if (pinned[t]) { first[t] = -1; last[t] = nops; }
    std::vector<int> order;
    for (int t = 0; t < nt; ++t)
        if (!net->tensor_ext[t]) order.push_back(t);
    auto slot = [&](int t) { return (net->tensor_bytes[t] + 255) / 256 * 256; };
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return slot(a) > slot(b); });
    std::vector<size_t> off(nt, 0);
    std::vector<int> placed;
    size_t top = 0;
    for (int t : order) {
        const size_t sz = slot(t);
        if (!sz) { off[t] = 0; continue; }
        std::vector<std::pair<size_t, size_t>> busy;      // [begin, end) of placed tensors alive at the same time
        for (int u : placed)
            if (first[u] <= last[t] && first[t] <= last[u]) busy.push_back({off[u], off[u] + slot(u)});
        std::sort(busy.begin(), busy.end());
        size_t at = 0;
        for (const auto& b : busy) {
            if (at + sz <= b.first) break;
            at = std::max(at, b.second);
        }
        off[t] = at;
        top = std::max(top, at + sz);
        placed.push_back(t);
    }
    const size_t ws_off = top, total = std::max<size_t>(256, top + (net->ws_bytes + 255) / 256 * 256);
    if (total >= net->arena_bytes) return SABER_HIP_OK;      // nothing to gain
    char* fresh = nullptr;
    HIP_TRY(hipMalloc((void**)&fresh, total));
    HIP_TRY(hipMemset(fresh, 0, total));
    HIP_TRY(hipDeviceSynchronize());      // whatever still runs on the old arena
    // the pass's inputs keep their contents (a caller may have filled them already)
    for (int t = 0; t < nt; ++t)
        if (!net->tensor_ext[t] && !writes[t] && net->tensor_bytes[t])
            HIP_TRY(hipMemcpy(fresh + off[t], net->arena + net->tensor_off[t], net->tensor_bytes[t], hipMemcpyDeviceToDevice));
    HIP_TRY(hipDeviceSynchronize());
    (void)hipFree(net->arena);
    net->arena = fresh;
    net->arena_bytes = total;
    net->ws_off = ws_off;
    for (int t = 0; t < nt; ++t)
        if (!net->tensor_ext[t]) net->tensor_off[t] = off[t];
    net->compacted = true;
    if (net->exec) {
        (void)hipGraphExecDestroy(net->exec);
        (void)hipGraphDestroy(net->graph);
        net->exec = nullptr;
        net->graph = nullptr;
    }
    return SABER_HIP_OK;
}
int saber_hip_net_arena_compacted(const saber_hip_net_t* net) { return net && net->compacted ? 1 : 0; }
void* saber_hip_net_tensor_ptr(saber_hip_net_t* net, int id) {
    if (!net->finalized || id < 0 || id >= (int)net->tensor_off.size()) return nullptr;
    return net->ptr(id);
}
size_t saber_hip_net_arena_bytes(const saber_hip_net_t* net) { return net->arena_bytes; }
int saber_hip_net_num_ops(const saber_hip_net_t* net) { return (int)net->ops.size(); }
const char* saber_hip_net_op_name(const saber_hip_net_t* net, int i) {
    return (i >= 0 && i < (int)net->ops.size()) ? net->ops[i].name.c_str() : "";
}
static int net_prepare_lanes(saber_hip_net* net) {
    if (net->lanes_ready) return SABER_HIP_OK;
    const int nops = (int)net->ops.size();
    net->writer.assign(net->tensor_bytes.size(), -1);
    net->ev_op.assign(nops, nullptr);
    net->has_side = false;
    std::vector<int> w(net->tensor_bytes.size(), -1);
    std::vector<std::vector<int>> rd(net->tensor_bytes.size());   // ops that read a tensor since its last write
    for (int i = 0; i < nops; ++i) {
        NetOp& o = net->ops[i];
        if (o.lane) net->has_side = true;
        const int ins[3] = {o.in, o.in2, o.out};   // `out` counts as an input: in-place epilogues read it
        for (int t : ins)
            if (t >= 0 && w[t] >= 0 && net->ops[w[t]].lane != o.lane) net->ops[w[t]].record = true;   // RAW / WAW
        const int outs[2] = {o.out, o.out2};
        for (int t : outs) {
            if (t < 0) continue;
            for (int r : rd[t])
                if (net->ops[r].lane != o.lane) net->ops[r].record = true;   // WAR: a reader on the other lane must finish first
            rd[t].clear();
            w[t] = i;
        }
        if (o.in >= 0) rd[o.in].push_back(i);
        if (o.in2 >= 0) rd[o.in2].push_back(i);
    }
    if (net->has_side) {
        HIP_TRY(hipStreamCreateWithFlags(&net->side, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&net->ev_start, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&net->ev_join, hipEventDisableTiming));
        for (int i = 0; i < nops; ++i)
            if (net->ops[i].record) HIP_TRY(hipEventCreateWithFlags(&net->ev_op[i], hipEventDisableTiming));
    }
    net->lanes_ready = true;
    return SABER_HIP_OK;
}

int saber_hip_net_run(saber_hip_net_t* net, saber_hip_stream_t stream) {
    if (!net->finalized) return fail(SABER_HIP_INVALID_VALUE, "net not finalized");
    int rc = net_prepare_lanes(net);
    if (rc) return rc;
    hipStream_t main_s = (hipStream_t)stream;
    if (!net->has_side) {
        for (const NetOp& o : net->ops) {
            rc = net_launch(net, o, main_s);
            if (rc) return rc;
        }
        return SABER_HIP_OK;
    }
    // fork: the side lane starts after everything already queued on the caller's stream
    HIP_TRY(hipEventRecord(net->ev_start, main_s));
    HIP_TRY(hipStreamWaitEvent(net->side, net->ev_start, 0));
    std::fill(net->writer.begin(), net->writer.end(), -1);
    std::vector<std::vector<int>> readers(net->tensor_bytes.size());
    bool side_dirty = false;
    for (size_t i = 0; i < net->ops.size(); ++i) {
        const NetOp& o = net->ops[i];
        hipStream_t s = o.lane ? net->side : main_s;
        const int ins[3] = {o.in, o.in2, o.out};
        for (int t : ins) {
            if (t < 0) continue;
            const int wi = net->writer[t];
            if (wi >= 0 && net->ops[wi].lane != o.lane) HIP_TRY(hipStreamWaitEvent(s, net->ev_op[wi], 0));
        }
        const int outs[2] = {o.out, o.out2};
        for (int t : outs) {   // write-after-read across lanes: every reader of the old contents has to be done
            if (t < 0) continue;
            for (int r : readers[t])
                if (net->ops[r].lane != o.lane) HIP_TRY(hipStreamWaitEvent(s, net->ev_op[r], 0));
            readers[t].clear();
        }
        rc = net_launch(net, o, s);
        if (rc) return rc;
        if (o.record) HIP_TRY(hipEventRecord(net->ev_op[i], s));
        if (o.lane) side_dirty = true;
        net->writer[o.out] = (int)i;
        if (o.out2 >= 0) net->writer[o.out2] = (int)i;
        if (o.in >= 0) readers[o.in].push_back((int)i);
        if (o.in2 >= 0) readers[o.in2].push_back((int)i);
    }
    if (side_dirty) {   // join: required before a capture ends, and so that the caller sees one ordered stream
        HIP_TRY(hipEventRecord(net->ev_join, net->side));
        HIP_TRY(hipStreamWaitEvent(main_s, net->ev_join, 0));
    }
    return SABER_HIP_OK;
}
int saber_hip_net_set_lane(saber_hip_net_t* net, int index, int lane) {
    if (index < 0 || index >= (int)net->ops.size() || lane < 0 || lane > 1) return fail(SABER_HIP_INVALID_VALUE, "bad op index / lane");
    if (net->lanes_ready) return fail(SABER_HIP_INVALID_VALUE, "lanes are fixed after the first run");
    for (const NetOp& o : net->ops)
        if (lane && (o.chain || o.chain3 || o.stem_pair))
            return fail(SABER_HIP_INVALID_VALUE, "the net has conv1x1 chain launches: lanes must be assigned before saber_hip_net_optimize (a chain launch spans several ops' tensors)");
    if (lane) {   // both lanes share the arena's single workspace: an op that uses it stays on the main lane
        const NetOp& o = net->ops[index];
        const size_t ws = o.kind == OP_CONV && o.conv ? o.conv->ws_bytes : (o.kind == OP_FC && o.fc ? saber_hip_fc_workspace_bytes(o.fc) : 0);
        if (ws) return fail(SABER_HIP_INVALID_VALUE, "an op that needs the shared workspace cannot run on the side lane");
    }
    net->ops[index].lane = lane;
    return SABER_HIP_OK;
}
int saber_hip_net_run_op(saber_hip_net_t* net, int index, saber_hip_stream_t stream) {
    if (!net->finalized || index < 0 || index >= (int)net->ops.size()) return fail(SABER_HIP_INVALID_VALUE, "bad op index");
    return net_launch(net, net->ops[index], (hipStream_t)stream);
}
static bool net_coop_error_pending(const saber_hip_net* net) {      // host reads of the sites' pinned error words: a few nanoseconds each
    for (const NetOp& o : net->ops) {
        if (o.stage && o.stage->h_err && *(volatile unsigned*)o.stage->h_err) return true;
        if (const saber_hip_chain* ch = o.chain3)
            if ((ch->h_coop_err && *(volatile unsigned*)ch->h_coop_err) || (ch->stage1 && ch->stage1->h_err && *(volatile unsigned*)ch->stage1->h_err))
                return true;
    }
    return false;
}
int saber_hip_net_capture(saber_hip_net_t* net, saber_hip_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    if (net->exec) (void)hipGraphExecDestroy(net->exec);
    if (net->graph) (void)hipGraphDestroy(net->graph);
    net->exec = nullptr;
    net->graph = nullptr;
    // a cooperative launch of an EARLIER pass that failed is dealt with before anything is recorded (the sites fall back, the caller
    // hears about it): inside the capture the site's own check would switch kernels while the stream is capturing
    if (net_coop_error_pending(net)) return saber_hip_net_status(net);
    HIP_TRY(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    int rc = saber_hip_net_run(net, stream);
    hipError_t e = hipStreamEndCapture(s, &net->graph);
    if (rc == SABER_HIP_OK && e != hipSuccess) rc = hip_fail(e, "hipStreamEndCapture");
    if (rc == SABER_HIP_OK && (e = hipGraphInstantiate(&net->exec, net->graph, nullptr, nullptr, 0)) != hipSuccess) rc = hip_fail(e, "hipGraphInstantiate");
    if (rc != SABER_HIP_OK) {      // no half-built graph survives a failed capture
        if (net->exec) (void)hipGraphExecDestroy(net->exec);
        if (net->graph) (void)hipGraphDestroy(net->graph);
        net->exec = nullptr;
        net->graph = nullptr;
    }
    return rc;
}
int saber_hip_net_replay(saber_hip_net_t* net, saber_hip_stream_t stream) {
    if (!net->exec) return fail(SABER_HIP_INVALID_VALUE, "net not captured");
    // an eager launch of a cooperative site checks its error word itself (stage_run, chain_run3); a graph node cannot: a failed earlier
    // pass is reported here, before the next one is launched (saber_hip_net_status: the sites fall back, the graph is dropped)
    if (net_coop_error_pending(net)) return saber_hip_net_status(net);
    HIP_TRY(hipGraphLaunch(net->exec, (hipStream_t)stream));
    return SABER_HIP_OK;
}
int saber_hip_net_coop_fallbacks(const saber_hip_net_t* net) { return net ? net->coop_fallbacks : 0; }
int saber_hip_coop_fallbacks_total(void) { return g_coop_fallbacks_total.load(); }
// Algorithmic work of ONE launch of op `index` (every tensor touched once — input, output, residual — plus the
// weights once; MACs x 2), summed over the operators the launch covers (a chain head reports its followers' work too, the
// followers report 0). Streaming ops: bytes only.
static void conv_work(const saber_hip_conv* c, double& bytes, double& flops) {
    const saber_hip_conv_desc& d = c->d;
    const double esz_in = d.in_dtype == SABER_HIP_F32 ? 4 : 1, esz_out = d.out_dtype == SABER_HIP_F32 ? 4 : 1;
    const double esz_w = c->is_i8 ? 1 : 4;
    const double in_el = (double)d.n * d.h * d.w * d.c;
    const int oh = c->oh, ow = c->ow;
    // a sibling pair carries k = k1 + k2 output channels; a fused pooling writes the pooled tensor
    const double out_el = (double)d.n * ((c->pool_fused || c->pool2) ? c->pool_oh * c->pool_ow : oh * ow) * d.k;
    bytes += in_el * esz_in + out_el * esz_out + (double)d.k * (d.c / d.group) * d.kh * d.kw * esz_w;
    if (d.res_mode != SABER_HIP_RES_NONE) bytes += (double)d.n * oh * ow * d.k * esz_out;
    flops += 2.0 * d.n * oh * ow * (double)d.k * (d.c / d.group) * d.kh * d.kw;
}
int saber_hip_net_op_work(const saber_hip_net_t* net, int index, double* bytes, double* flops) {
    if (!net || index < 0 || index >= (int)net->ops.size() || !bytes || !flops) return fail(SABER_HIP_INVALID_VALUE, "bad argument");
    *bytes = 0; *flops = 0;
    const NetOp& o = net->ops[index];
    auto tb = [&](int t) { return t >= 0 ? (double)net->tensor_bytes[t] : 0.0; };
    switch (o.kind) {
    case OP_CONV:
        if (o.skip) return SABER_HIP_OK;
        conv_work(o.conv, *bytes, *flops);
        if (o.stem_pair) {      // + the pair's work (the algorithmic bytes of the separate ops, like the chains')
            conv_work(net->ops[index + 1].conv, *bytes, *flops);
        } else if (o.chain3 && o.use_chain3) {
            for (int j = index + 1; j < (int)net->ops.size() && net->ops[j].skip; ++j) conv_work(net->ops[j].conv, *bytes, *flops);
        } else if (o.chain && o.use_chain) {
            if (index + 1 < (int)net->ops.size() && net->ops[index + 1].skip) conv_work(net->ops[index + 1].conv, *bytes, *flops);
        }
        return SABER_HIP_OK;
    case OP_CONV_PAIR:
        if (!o.skip) conv_work(o.conv, *bytes, *flops);
        return SABER_HIP_OK;
    case OP_FC:
    case OP_FC_Q: {
        const saber_hip_fc_desc& d = o.fc->d;
        const double esz = d.int8_weights ? 1 : 4;
        *bytes = (double)d.m * d.k * (o.kind == OP_FC_Q || d.in_dtype != SABER_HIP_F32 ? 1 : 4) + (double)d.m * d.n * 4 + (double)d.n * d.k * esz;
        *flops = 2.0 * d.m * d.n * d.k;
        if (o.out2 >= 0) *bytes += 2.0 * tb(o.out2);      // + the softmax it runs (flag 4096): the logits read back, the probabilities written
        return SABER_HIP_OK;
    }
    default:
        if (o.skip) return SABER_HIP_OK;
        *bytes = tb(o.in) + tb(o.in2) + tb(o.out) + tb(o.out2);
        return SABER_HIP_OK;
    }
}
// Per-op time INSIDE a forward pass: one event after every launch of an eager pass, averaged over `iters` passes
// (out_us[i] = event[i] - event[i-1]; skipped ops report 0). Unlike saber_hip_net_time_ops (each op repeated back to back,
// operands warm) this is the op in its place in the pipeline, boundary included; the events themselves add to the pass, so
// use the SHARES and scale them to the untimed step time.
int saber_hip_net_time_pass(saber_hip_net_t* net, saber_hip_stream_t stream, int iters, float* out_us) {
    if (!net || !net->finalized || iters <= 0 || !out_us) return fail(SABER_HIP_INVALID_VALUE, "bad argument");
    hipStream_t s = (hipStream_t)stream;
    const size_t n = net->ops.size();
    struct Events {      // destroyed on every exit path
        std::vector<hipEvent_t> v;
        ~Events() {
            for (hipEvent_t e : v)
                if (e) (void)hipEventDestroy(e);
        }
    } evs;
    evs.v.assign(n + 1, nullptr);
    std::vector<hipEvent_t>& ev = evs.v;
    for (auto& e : ev) HIP_TRY(hipEventCreate(&e));
    std::vector<double> acc(n, 0.0);
    for (int it = 0; it < iters + 1; ++it) {      // the first pass warms up and is dropped
        HIP_TRY(hipEventRecord(ev[0], s));
        size_t last = 0;                          // index (into ev) of the newest recorded event
        std::vector<size_t> prev(n, 0);
        for (size_t i = 0; i < n; ++i) {
            const NetOp& o = net->ops[i];
            int rc = net_launch(net, o, s);
            if (rc) return rc;
            prev[i] = last;
            if (o.skip) continue;     // launches nothing: no event (a marker packet costs ~2.5 us itself)
            HIP_TRY(hipEventRecord(ev[i + 1], s));
            last = i + 1;
        }
        HIP_TRY(hipEventSynchronize(ev[last]));
        if (!it) continue;
        for (size_t i = 0; i < n; ++i) {
            const NetOp& o = net->ops[i];
            if (o.skip) continue;
            float ms = 0;
            HIP_TRY(hipEventElapsedTime(&ms, ev[prev[i]], ev[i + 1]));
            acc[i] += ms;
        }
    }
    for (size_t i = 0; i < n; ++i) out_us[i] = (float)(acc[i] * 1000.0 / iters);
    return SABER_HIP_OK;
}
// ONE op's launch duration inside a forward pass, undisturbed: whole eager passes with just two events, one in front of the op's
// launch and one behind it (both on the launch stream: the first is reached when the previous kernel has finished). The per-launch
// events of saber_hip_net_time_pass stretch a pass (a marker per launch), so its shares - scaled back to the untimed step -
// understate a long kernel among many short ones; this is the figure that agrees with a rocprofv3 kernel trace.
int saber_hip_net_time_op_in_pass(saber_hip_net_t* net, saber_hip_stream_t stream, int index, int iters, float* out_us) {
    if (!net || !net->finalized || iters <= 0 || !out_us || index < 0 || index >= (int)net->ops.size())
        return fail(SABER_HIP_INVALID_VALUE, "bad argument");
    hipStream_t s = (hipStream_t)stream;
    EventPair evp;
    HIP_TRY(evp.init());
    double acc = 0.0;
    for (int it = 0; it < iters + 1; ++it) {      // the first pass warms up and is dropped
        for (size_t i = 0; i < net->ops.size(); ++i) {
            if ((int)i == index) HIP_TRY(hipEventRecord(evp.e0, s));
            int rc = net_launch(net, net->ops[i], s);
            if (rc) return rc;
            if ((int)i == index) HIP_TRY(hipEventRecord(evp.e1, s));
        }
        HIP_TRY(hipStreamSynchronize(s));
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, evp.e0, evp.e1));
        if (it) acc += ms;
    }
    *out_us = (float)(acc * 1000.0 / iters);
    return SABER_HIP_OK;
}
int saber_hip_net_time_ops(saber_hip_net_t* net, saber_hip_stream_t stream, int iters, float* out_us) {
    hipStream_t s = (hipStream_t)stream;
    EventPair evp;
    HIP_TRY(evp.init());
    hipEvent_t e0 = evp.e0, e1 = evp.e1;
    for (size_t i = 0; i < net->ops.size(); ++i) {
        int rc = net_launch(net, net->ops[i], s);
        if (rc) return rc;
        HIP_TRY(hipEventRecord(e0, s));
        for (int it = 0; it < iters; ++it) net_launch(net, net->ops[i], s);
        HIP_TRY(hipEventRecord(e1, s));
        HIP_TRY(hipEventSynchronize(e1));
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
        out_us[i] = ms * 1000.f / iters;
    }
    return SABER_HIP_OK;
}
// 1 when tensor `id` is the output edge of a 3x3 conv that currently runs inside a conv3x3 + chain launch (not written)
int saber_hip_net_tensor_unwritten(const saber_hip_net_t* net, int id) {
    if (id < 0 || id >= (int)net->tensor_bytes.size()) return 0;
    if (net->tensor_bytes[id] == 0) return 1;      // the edge was removed by saber_hip_net_optimize (it has no storage)
    for (const NetOp& o : net->ops)
        if (((o.chain3 && o.use_chain3) || o.stem_pair) && o.out == id) return 1;
    return 0;
}
// After a pass has COMPLETED (the caller has synchronised): did one of its cooperative launches - a stage launch, a two-workgroup
// chain - find its workgroups on different XCDs or time out in a hand-off (its outputs are then not valid)? The pinned error
// words are read and cleared, the affected sites fall back to their single-workgroup launches for good, a captured graph is dropped.
// SABER_HIP_RUNTIME_ERROR tells the caller to run the pass again. (Without this call the next launch of the site reports it.)
static void net_drop_graph(saber_hip_net* net) {
    if (!net->exec) return;
    (void)hipGraphExecDestroy(net->exec);
    (void)hipGraphDestroy(net->graph);
    net->exec = nullptr;
    net->graph = nullptr;
}
int saber_hip_net_status(saber_hip_net_t* net) {
    if (!net || !net->finalized) return fail(SABER_HIP_INVALID_VALUE, "net not finalized");
    int bad = 0;
    for (size_t i = 0; i < net->ops.size(); ++i) {
        NetOp& o = net->ops[i];
        if (o.stage && o.stage->h_err && *(volatile unsigned*)o.stage->h_err) {
            *(volatile unsigned*)o.stage->h_err = 0u;
            if (o.use_stage) net_set_stage(net, (int)i, false);
            ++bad;
        }
        saber_hip_chain* ch = o.chain3;
        if (ch) {
            unsigned* words[2] = {ch->h_coop_err, ch->stage1 ? ch->stage1->h_err : nullptr};
            for (unsigned* w : words)
                if (w && *(volatile unsigned*)w) {
                    *(volatile unsigned*)w = 0u;
                    if (ch->tn == 7 || ch->tn == 15) (void)saber_hip_conv2d_chain_set_tile(ch, 3);
                    ++bad;
                }
        }
    }
    if (!bad) return SABER_HIP_OK;
    net->coop_fallbacks += bad;
    g_coop_fallbacks_total += bad;
    net_drop_graph(net);
    return fail(SABER_HIP_RUNTIME_ERROR, "a cooperative launch of the last pass did not complete (workgroups on different XCDs, or a hand-off "
                "timed out: another kernel held the CUs); its outputs are not valid - those sites now launch block by block: run the pass again");
}
// testing aid: makes the next saber_hip_net_status report a failed cooperative launch at the first site that has an error word
int saber_hip_net_inject_coop_error(saber_hip_net_t* net) {
    if (!net) return fail(SABER_HIP_INVALID_VALUE, "null argument");
    for (NetOp& o : net->ops) {
        if (o.stage && o.stage->h_err) { *(volatile unsigned*)o.stage->h_err = 1u; return SABER_HIP_OK; }
        if (o.chain3 && o.chain3->h_coop_err) { *(volatile unsigned*)o.chain3->h_coop_err = 1u; return SABER_HIP_OK; }
    }
    return fail(SABER_HIP_INVALID_VALUE, "the net has no cooperative launch site");
}
int saber_hip_net_num_launches(const saber_hip_net_t* net) {
    int n = 0;
    for (const NetOp& o : net->ops) n += o.skip ? 0 : 1;
    return n;
}
void saber_hip_net_destroy(saber_hip_net_t* net) {
    if (!net) return;
    if (net->exec) (void)hipGraphExecDestroy(net->exec);
    if (net->graph) (void)hipGraphDestroy(net->graph);
    if (net->arena) (void)hipFree(net->arena);
    for (saber_hip_stem_pair* sp : net->owned_stem_pairs) saber_hip_conv2d_stem_pair_destroy(sp);
    for (saber_hip_chain_stage* st : net->owned_stages) saber_hip_conv2d_stage_destroy(st);
    for (saber_hip_chain* c : net->owned_chains) saber_hip_conv2d_chain_destroy(c);
    for (saber_hip_conv* c : net->owned) saber_hip_conv2d_destroy(c);
    for (hipEvent_t e : net->ev_op)
        if (e) (void)hipEventDestroy(e);
    if (net->ev_start) (void)hipEventDestroy(net->ev_start);
    if (net->ev_join) (void)hipEventDestroy(net->ev_join);
    if (net->side) (void)hipStreamDestroy(net->side);
    delete net;
}

