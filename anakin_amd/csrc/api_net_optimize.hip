// anakin_amd/csrc/api_net_optimize.hip - saber_hip_net_optimize: executor-level fusions.
#include "api_internal.h"

// ------------------------------------------------------------------------------------------------
// Executor-level fusions, host side C++ (north star: "host side stays C++"): the reference's graph optimiser rewrites
// the operator graph before Net::init (framework/graph/llvm/fusion/fusion_op_register.cpp:45-175 is its pattern
// catalogue: Conv+Eltwise, Conv+Pooling, ...); this is the same step for an op list handed to the MI355X executor
// UNFUSED (one op per reference operator). Every rewrite keeps the bytes of every surviving edge identical:
//   1  conv (-> s8, single consumer) + INT8 eltwise sum        -> one conv with the RES_ELTWISE epilogue
//   2  two convs over the same tensor with the same geometry   -> one sibling-pair launch (saber_hip_conv2d_create_pair)
//   4  conv + max pooling (single consumer)                    -> SaberConv2DPooling where a fused kernel exists
//   8  global pooling feeding an INT8 fc that quantises on entry -> the pooling also writes the fc's s8 operand
//  16  1x1 conv with the fused eltwise epilogue + the 1x1 conv that follows it and reads its output (ResNet branch2c + sum
//      -> next branch2a) -> one conv1x1-chain launch (saber_hip_conv2d_chain_create); both ops stay in the list (the
//      second one launches nothing while the chain is selected), the autotuner keeps whichever form is faster
// New ops are re-created from the originals' quantised weights / scales / bias and owned by the net. Call before
// saber_hip_net_finalize. Returns the number of launches removed, or a negative status.
// ------------------------------------------------------------------------------------------------
static void net_name_chain(NetOp& A, NetOp& B) {
    if (A.skip) {
        A.name = B.name = "conv:(in the chain launch)";
    } else if (A.use_chain) {
        const int t = A.chain->tn, c = A.chain->c1;
        const bool w8 = t == 11 || (c == 128 && (t & 4));
        A.name = "conv:chain1x1_c" + std::to_string(c) + "_px" + std::to_string(t == 11 ? 16 : 16 * (c == 128 ? t & 3 : t & 7)) +
                 ((t & 8) ? "_split2" : "") + (w8 ? "_w8" : "");
        B.name = "conv:(in the chain launch)";
    } else {
        A.name = std::string("conv:") + A.conv->algo_name;
        B.name = std::string("conv:") + B.conv->algo_name;
    }
}
// mode of the ops around A = ops[ia] (a 1x1 conv with the fused eltwise): 0 separate launches, 1 A + B chained (A.chain),
// 2 the 3x3 conv ops[ia - 1] leads the launch (its chain3; with or without B)
static std::string stage_name(const NetOp& H0) {
    return "conv:stage_c" + std::to_string(H0.stage->c1) + "_" + std::to_string(H0.stage_n) + "x[conv3x3+chain1x1]_2x16" + (H0.stage->c1 == 256 ? "_coop4" : "");
}
void net_set_chain_mode(saber_hip_net* net, int ia, int mode) {
    NetOp& A = net->ops[ia];
    NetOp* H = (ia > 0 && net->ops[ia - 1].chain3) ? &net->ops[ia - 1] : nullptr;
    if (mode == 2 && !H) mode = 1;
    if (mode == 1 && !A.chain) mode = 0;
    NetOp* B = A.chain ? &net->ops[ia + 1] : nullptr;
    A.use_chain = mode == 1 || (mode == 2 && H->chain3->b && !H->chain3->b2);
    if (B) B->skip = mode == 1 || (mode == 2 && H->chain3->b);
    A.skip = mode == 2;
    if (H && H->chain3->b2) {      // strided head + sibling pair: the pair op behind A launches nothing while the chain runs
        NetOp& P = net->ops[ia + 1];
        P.skip = mode == 2;
        P.name = mode == 2 ? "conv:(in the chain launch)" : std::string("conv:") + P.conv->algo_name;
    }
    if (H) {
        H->use_chain3 = mode == 2;
        H->name = mode == 2 ? std::string("conv:conv3x3+") + (H->chain3->b2 ? "conv1x1+pair1x1_c" : (H->chain3->b ? "chain1x1_c" : "conv1x1_c")) + std::to_string(H->chain3->c1) +
                                  "_" + std::to_string(H->chain3->c1 == 128 ? H->chain3->tn & 3 : (H->chain3->c1 == 256 ? (H->chain3->tn == 15 ? 2 : 1) : H->chain3->tn)) + "x16" +
                                  ((H->chain3->c1 == 128 && (H->chain3->tn & 4)) || (H->chain3->c1 == 256 && H->chain3->tn == 3) ? "_w8" : "") +
                                  (H->chain3->c1 == 256 && H->chain3->tn == 7 ? "_coop2" : "") + (H->chain3->c1 == 256 && H->chain3->tn == 15 ? "_coop4" : "")
                            : std::string("conv:") + H->conv->algo_name;
    }
    if (B) net_name_chain(A, *B);
    else A.name = A.skip ? "conv:(in the chain launch)" : std::string("conv:") + A.conv->algo_name;
    if (H && H->stage && H->use_stage) H->name = stage_name(*H);      // (an active stage keeps its names: set_choice comes through here)
    else if (H && H->skip) H->name = "conv:(in the stage launch)";
}
// The stage headed by ops[i0] on / off. On: every block runs its 3x3-led chain form (mode 2), ops[i0] launches them all and the
// other blocks' 3x3 convs carry `skip` too. Off: the blocks' chains launch one by one again (mode 2, their own tile codes).
std::string stem_pair_name(const NetOp& o) {
    return std::string("conv:") + o.conv->algo_name + "+pair1x1_" + std::to_string(o.stem_pair->a->d.k) + "+" + std::to_string(o.stem_pair->b->d.k);
}
void net_set_stage(saber_hip_net* net, int i0, bool on) {
    NetOp& H0 = net->ops[i0];
    if (!H0.stage) return;
    for (int k = 0; k < H0.stage_n; ++k) {
        if (on || H0.use_stage) net_set_chain_mode(net, i0 + 3 * k + 1, 2);
        NetOp& Hk = net->ops[i0 + 3 * k];
        if (k) {
            Hk.skip = on;
            if (on) Hk.name = "conv:(in the stage launch)";
        }
    }
    H0.use_stage = on;
    if (on) H0.name = stage_name(H0);
}
int net_chain_mode(const saber_hip_net* net, int ia) {
    const NetOp& A = net->ops[ia];
    return A.skip ? 2 : (A.use_chain ? 1 : 0);
}
static int clone_conv_i8(const saber_hip_conv* src, const saber_hip_conv_desc& d, saber_hip_conv** out) {
    int rc = saber_hip_conv2d_create(&d, out);
    if (rc) return rc;
    rc = saber_hip_conv2d_set_weights(*out, src->wq_oihw.data(), SABER_HIP_S8, src->w_scale.data(),
                                      src->has_bias ? src->bias_host.data() : nullptr, src->in_scale, src->out_scale);
    if (rc) {
        saber_hip_conv2d_destroy(*out);
        *out = nullptr;
    }
    return rc;
}

int saber_hip_net_optimize(saber_hip_net_t* net, int flags) {
    if (!net) return fail(SABER_HIP_INVALID_VALUE, "null argument");
    if (net->finalized) return fail(SABER_HIP_INVALID_VALUE, "optimize must run before finalize");
    if (flags & 2048) {      // the net does not own the device: decided HERE, at selection time, not when a launch has failed
        net->shared_device = true;
        flags &= ~256;       // no persistent stage launch
    }
    if (net->shared_device) flags &= ~256;
    if (flags & SABER_HIP_NET_REPRODUCIBLE_FP32) net->reproducible_fp32 = true;      // sticks to the net (api_net_autotune.hip)
    std::vector<NetOp>& ops = net->ops;
    const int nt = (int)net->tensor_bytes.size();
    std::vector<char> dead(ops.size(), 0);
    int removed = 0;
    auto consumers = [&](int t) {
        int c = 0;
        for (size_t i = 0; i < ops.size(); ++i)
            if (!dead[i]) c += (ops[i].in == t) + (ops[i].in2 == t);
        return c;
    };
    auto producer = [&](int t, int before) {   // last live op before `before` that writes tensor t
        for (int i = before - 1; i >= 0; --i)
            if (!dead[i] && (ops[i].out == t || ops[i].out2 == t)) return i;
        return -1;
    };
    auto plain_i8_conv = [&](const NetOp& o) {
        return o.kind == OP_CONV && o.conv && o.conv->is_i8 && o.conv->weights_set && o.conv->epi == EPI_I8_CONV &&
               o.conv->d.res_mode == SABER_HIP_RES_NONE && !o.conv->pair_k2 && !o.conv->pool_fused && o.lane == 0;
    };
    // ---- 1: conv + eltwise -----------------------------------------------------------------------------------
    if (flags & 1) {
        for (size_t i = 0; i < ops.size(); ++i) {
            if (dead[i] || ops[i].kind != OP_ELT_I8) continue;
            NetOp& e = ops[i];
            for (int side = 0; side < 2; ++side) {
                const int tc = side == 0 ? e.in : e.in2, tr = side == 0 ? e.in2 : e.in;   // conv-side / residual-side tensors
                const float s_conv = e.f[side], s_res = e.f[1 - side], c_conv = e.f[2 + side], c_res = e.f[3 - side];
                const int p = producer(tc, (int)i);
                if (p < 0 || !plain_i8_conv(ops[p]) || consumers(tc) != 1 || tc == tr) continue;
                const saber_hip_conv* src = ops[p].conv;
                if (src->d.out_dtype != SABER_HIP_S8 || src->out_scale != s_conv) continue;
                if (net->tensor_bytes[tc] != net->tensor_bytes[e.out] || e.count != net->tensor_bytes[tc]) continue;
                const int pr = producer(tr, (int)i);
                if (pr >= p) continue;   // the residual must exist when the conv runs (external tensors: pr == -1)
                // the fused conv writes e.out at the CONV's position: no live op in (p, i) may read or write that tensor,
                // and it must not alias the residual (in-place eltwise)
                bool clash = e.out == tr;
                for (int j = p + 1; j < (int)i && !clash; ++j)
                    if (!dead[j]) clash = ops[j].in == e.out || ops[j].in2 == e.out || ops[j].out == e.out || ops[j].out2 == e.out;
                if (clash) continue;
                saber_hip_conv_desc d = src->d;
                d.res_mode = SABER_HIP_RES_ELTWISE;
                d.res_act = e.p[0] ? SABER_HIP_ACT_RELU : SABER_HIP_ACT_NONE;
                d.coeff_conv = c_conv; d.coeff_res = c_res; d.scale_res = s_res;
                saber_hip_conv* fused = nullptr;
                int rc = clone_conv_i8(src, d, &fused);
                if (rc) return rc;
                net->owned.push_back(fused);
                ops[p].conv = fused;
                ops[p].in2 = tr;
                ops[p].out = e.out;
                ops[p].name = std::string("conv:") + fused->algo_name;
                dead[i] = 1;
                net->tensor_bytes[tc] = 0;   // the conv's own output edge no longer exists
                ++removed;
                break;
            }
        }
    }
    // ---- 64: a shortcut's 1x1 / stride-s max pooling read by a fused eltwise epilogue only -> folded into that read ----
    // (the pooling graph_strategy::apply_stride_up inserts, optimize_strategy.h:213-248: one element per window, so the op is
    // a spatial subsampling; the conv then reads the pooling's SOURCE at (oy * s, ox * s): saber_hip_conv_desc::res_stride)
    if (flags & 64) {
        for (size_t i = 0; i < ops.size(); ++i) {
            if (dead[i] || ops[i].kind != OP_POOL_I8) continue;
            NetOp& q = ops[i];   // p[] = n,h,w,c,oh,ow,kh,kw,sh,sw,ph,pw,type,in_dtype,out_dtype
            if (q.p[6] != 1 || q.p[7] != 1 || q.p[8] != q.p[9] || q.p[8] < 2 || q.p[10] || q.p[11] || q.p[12] != SABER_HIP_POOL_MAX ||
                q.p[13] != SABER_HIP_S8 || q.p[14] != SABER_HIP_S8 || (q.p[1] - 1) / q.p[8] + 1 != q.p[4] ||
                (q.p[2] - 1) / q.p[8] + 1 != q.p[5] || consumers(q.out) != 1)
                continue;
            int c = -1;
            for (size_t j = i + 1; j < ops.size() && c < 0; ++j) {
                if (dead[j]) continue;
                if (ops[j].out == q.in || ops[j].out2 == q.in) break;            // the pooling's source is rewritten first
                if (ops[j].in2 == q.out && ops[j].kind == OP_CONV && ops[j].conv && !ops[j].chain && !ops[j].chain3 &&
                    ops[j].conv->d.res_mode == SABER_HIP_RES_ELTWISE && ops[j].conv->d.res_stride <= 1 && ops[j].lane == q.lane)
                    c = (int)j;
                else if (ops[j].in == q.out || ops[j].in2 == q.out) break;       // some other reader
            }
            if (c < 0) continue;
            const saber_hip_conv* src = ops[c].conv;
            if (src->oh != q.p[4] || src->ow != q.p[5] || src->d.k != q.p[3] || src->d.n != q.p[0]) continue;
            saber_hip_conv_desc d = src->d;
            d.res_stride = q.p[8]; d.res_h = q.p[1]; d.res_w = q.p[2];
            saber_hip_conv* fused = nullptr;
            if (clone_conv_i8(src, d, &fused) != SABER_HIP_OK) continue;       // (the direct fallback kernel cannot: keep the op)
            net->owned.push_back(fused);
            ops[c].conv = fused;
            ops[c].in2 = q.in;
            ops[c].name = std::string("conv:") + fused->algo_name + "+res/" + std::to_string(q.p[8]);
            dead[i] = 1;
            net->tensor_bytes[q.out] = 0;
            ++removed;
        }
    }
    // ---- 4: conv + max pooling ----------------------------------------------------------------------------------
    if (flags & 4) {
        for (size_t i = 0; i < ops.size(); ++i) {
            if (dead[i] || ops[i].kind != OP_POOL_I8) continue;
            NetOp& q = ops[i];
            const int p = producer(q.in, (int)i);
            if (p < 0 || !plain_i8_conv(ops[p]) || consumers(q.in) != 1) continue;
            const saber_hip_conv* src = ops[p].conv;
            saber_hip_conv* fused = nullptr;
            int rc = clone_conv_i8(src, src->d, &fused);
            if (rc) return rc;
            // p[] = n,h,w,c,oh,ow,kh,kw,sh,sw,ph,pw,type,in_dtype,out_dtype (saber_hip_net_add_pool_i8)
            const int floor_mode = saber_hip_pool_out_dim(q.p[1], q.p[10], q.p[6], q.p[8], 0) == q.p[4] ? 0 : 1;
            rc = saber_hip_conv2d_set_pooling(fused, q.p[12], q.p[6], q.p[7], q.p[8], q.p[9], q.p[10], q.p[11], floor_mode);
            int oh = 0, ow = 0;
            if (rc == SABER_HIP_OK) saber_hip_conv2d_out_shape(fused, &oh, &ow);
            if (rc != SABER_HIP_OK || oh != q.p[4] || ow != q.p[5] || q.p[13] != q.p[14]) {   // no fused kernel: keep the two ops
                saber_hip_conv2d_destroy(fused);
                continue;
            }
            net->owned.push_back(fused);
            ops[p].conv = fused;
            const int dead_t = ops[p].out;
            ops[p].out = q.out;
            ops[p].name = std::string("conv:") + fused->algo_name;
            dead[i] = 1;
            net->tensor_bytes[dead_t] = 0;
            ++removed;
        }
    }
    // ---- 2: sibling pairs -------------------------------------------------------------------------------------
    if (flags & 2) {
        for (size_t i = 0; i < ops.size(); ++i) {
            if (dead[i] || !plain_i8_conv(ops[i])) continue;
            for (size_t j = i + 1; j < ops.size(); ++j) {
                if (dead[j]) continue;
                if (ops[j].out == ops[i].in || ops[j].out2 == ops[i].in) break;   // the shared input is rewritten: stop
                if (!plain_i8_conv(ops[j]) || ops[j].in != ops[i].in) continue;
                {   // op j's output is written at position i instead: nothing in [i, j) may touch that tensor
                    bool clash = false;
                    for (size_t m = i; m < j && !clash; ++m)
                        if (!dead[m]) clash = ops[m].in == ops[j].out || ops[m].in2 == ops[j].out || ops[m].out == ops[j].out || ops[m].out2 == ops[j].out;
                    if (clash) continue;
                }
                saber_hip_conv* pair = nullptr;
                bool swapped = false;
                if (saber_hip_conv2d_create_pair(ops[i].conv, ops[j].conv, &pair) != SABER_HIP_OK) {
                    // the first rows must be a multiple of the largest block tile: try the other order
                    if (saber_hip_conv2d_create_pair(ops[j].conv, ops[i].conv, &pair) != SABER_HIP_OK) continue;
                    swapped = true;
                }
                // hoisting op j to position i is safe: it only reads the shared input, and nothing between reads its output
                // before j (a consumer of j's output cannot precede j in a valid list)
                net->owned.push_back(pair);
                ops[i].kind = OP_CONV_PAIR;
                ops[i].conv = pair;
                ops[i].out2 = ops[j].out;
                if (swapped) std::swap(ops[i].out, ops[i].out2);
                ops[i].name = std::string("conv:") + pair->algo_name;
                dead[j] = 1;
                ++removed;
                break;
            }
        }
    }
    // ---- 8: global pooling writes the fc's quantised operand ---------------------------------------------------------
    if (flags & 8) {
        for (size_t i = 0; i < ops.size(); ++i) {
            if (dead[i] || ops[i].kind != OP_FC || !ops[i].fc->pre_quant) continue;
            const int p = producer(ops[i].in, (int)i);
            if (p < 0 || ops[p].kind != OP_POOL_F32_I8 || ops[p].out2 >= 0) continue;
            const saber_hip_fc* fc = ops[i].fc;
            // the pooled tensor must be exactly the fc's [m, k] operand
            if ((size_t)ops[p].p[0] * ops[p].p[3] * ops[p].p[4] * ops[p].p[5] != (size_t)fc->d.m * fc->d.k) continue;
            const int qt = saber_hip_net_add_tensor(net, (size_t)fc->d.m * fc->d.k);
            ops[p].out2 = qt;
            ops[p].f[1] = fc->in_scale;
            ops[p].name = "pool2d_f32_from_i8+quantize";
            ops[i].kind = OP_FC_Q;
            ops[i].in = qt;
            ++removed;   // the fc's quantise-on-entry kernel
        }
    }
    // ---- 128: conv + global average pooling (INT8): the pooling's launch disappears, both tensors are written -------
    if (flags & 128) {
        for (size_t i = 0; i < ops.size(); ++i) {
            if (dead[i] || ops[i].kind != OP_POOL_I8) continue;
            NetOp& q = ops[i];   // p[] = n,h,w,c,oh,ow,kh,kw,sh,sw,ph,pw,type,in_dtype,out_dtype
            if (q.p[4] != 1 || q.p[5] != 1 || q.p[6] != q.p[1] || q.p[7] != q.p[2] || q.p[10] || q.p[11] ||
                q.p[12] == SABER_HIP_POOL_MAX || q.p[13] != q.p[14])
                continue;
            const int p = producer(q.in, (int)i);
            if (p < 0 || ops[p].kind != OP_CONV || !ops[p].conv || ops[p].lane != q.lane || ops[p].out2 >= 0) continue;
            const saber_hip_conv* src = ops[p].conv;
            if (!src->is_i8 || src->pair_k2 || src->pool_fused || src->gpool || !img_conv_ok(src) || src->d.n != q.p[0] ||
                src->d.k != q.p[3] || src->oh != q.p[1] || src->ow != q.p[2] || src->d.out_dtype != q.p[13])
                continue;
            {   // the pooled tensor is written at the CONV's position instead: nothing in (p, i) may read or write it
                bool clash = false;
                for (int j = p + 1; j < (int)i && !clash; ++j)
                    if (!dead[j]) clash = ops[j].in == q.out || ops[j].in2 == q.out || ops[j].out == q.out || ops[j].out2 == q.out;
                if (clash) continue;
            }
            saber_hip_conv* fused = nullptr;
            if (clone_conv_i8(src, src->d, &fused) != SABER_HIP_OK) continue;
            if (saber_hip_conv2d_set_global_pooling(fused) != SABER_HIP_OK) {
                saber_hip_conv2d_destroy(fused);
                continue;
            }
            net->owned.push_back(fused);
            ops[p].conv = fused;
            ops[p].out2 = q.out;
            ops[p].name = std::string("conv:") + fused->algo_name;
            dead[i] = 1;
            ++removed;
        }
    }
    (void)nt;
    std::vector<NetOp> live;
    for (size_t i = 0; i < ops.size(); ++i)
        if (!dead[i]) live.push_back(std::move(ops[i]));
    ops.swap(live);
    // ---- 512: the fused stem conv + pooling runs the sibling pair that reads the pooled tensor (on the compacted list) --------
    bool side_lane = false;      // (like the chains: a launch that writes other ops' tensors is not ordered across lanes)
    for (const NetOp& o : ops) side_lane |= o.lane != 0;
    if ((flags & 512) && !side_lane) {
        for (size_t i = 0; i + 1 < ops.size(); ++i) {
            NetOp& S = ops[i];
            NetOp& P = ops[i + 1];
            if (S.kind != OP_CONV || !S.conv || !S.conv->pool_fused || S.lane || S.in2 >= 0 || S.out2 >= 0 || S.stem_pair) continue;
            if (P.kind != OP_CONV_PAIR || !P.conv || P.lane || P.in != S.out || !P.conv->pair_src_a || !P.conv->pair_src_b) continue;
            int readers = 0;
            for (const NetOp& o : ops) readers += (o.in == S.out) + (o.in2 == S.out);
            if (readers != 1) continue;
            saber_hip_stem_pair* sp = nullptr;
            if (saber_hip_conv2d_stem_pair_create(S.conv, P.conv->pair_src_a, P.conv->pair_src_b, &sp) != SABER_HIP_OK) continue;
            net->owned_stem_pairs.push_back(sp);
            S.stem_pair = sp;
            S.stem_y1 = P.out;
            S.stem_y2 = P.out2;
            S.name = stem_pair_name(S);
            P.skip = true;
            P.name = "conv:(in the stem launch)";
            ++removed;
        }
    }
    // A chain launch reads and writes the tensors of SEVERAL ops (chain3_res / chain_out / chain3_y1 / chain3_y2), while the
    // cross-lane event ordering of saber_hip_net_run only follows the launching op's own in / in2 / out / out2: in a
    // two-lane net a chained launch could read a residual produced on the other lane before its event. The two executor
    // options are therefore exclusive: no chains once any op sits on the side lane (and saber_hip_net_set_lane refuses
    // a lane change once chains exist).
    bool two_lanes = false;
    for (const NetOp& o : ops) two_lanes |= o.lane != 0;
    if (two_lanes) flags &= ~(16 | 32);
    // ---- 16: conv1x1 chains (on the compacted list: the pair must be adjacent) --------------------------------
    if (flags & 16) {
        for (size_t i = 0; i + 1 < ops.size(); ++i) {
            NetOp& A = ops[i];
            NetOp& B = ops[i + 1];
            if (A.kind != OP_CONV || B.kind != OP_CONV || !A.conv || !B.conv || A.chain || A.skip || B.chain || A.lane || B.lane ||
                A.chain3 || B.chain3)
                continue;
            if (A.conv->d.res_mode != SABER_HIP_RES_ELTWISE || B.in != A.out || A.in2 < 0 || B.in2 >= 0) continue;
            saber_hip_chain* ch = nullptr;
            if (saber_hip_conv2d_chain_create(A.conv, B.conv, &ch) != SABER_HIP_OK) continue;   // not a chainable shape
            net->owned_chains.push_back(ch);
            A.chain = ch;
            A.chain_out = B.out;
            net_set_chain_mode(net, (int)i, ch->c1 <= 256 ? 1 : 0);      // default until the autotuner has timed both forms
            if (A.use_chain) ++removed;
        }
    }
    // ---- 32: the block's 3x3 conv in front of a chain head, when the head is its only consumer -----------------------
    if (flags & 32) {
        for (size_t i = 0; i + 2 < ops.size(); ++i) {
            NetOp& Hd = ops[i];
            NetOp& A = ops[i + 1];
            NetOp& B = ops[i + 2];
            if (!A.chain || Hd.kind != OP_CONV || !Hd.conv || Hd.chain3 || Hd.chain || Hd.skip || Hd.lane || Hd.in2 >= 0 ||
                A.in != Hd.out)
                continue;
            int readers = 0;
            for (const NetOp& o : ops) readers += (o.in == Hd.out) + (o.in2 == Hd.out);
            if (readers != 1) continue;
            saber_hip_chain* ch = nullptr;
            if (saber_hip_conv2d_chain_create3(Hd.conv, A.conv, B.conv, &ch) != SABER_HIP_OK) continue;
            net->owned_chains.push_back(ch);
            Hd.chain3 = ch;
            Hd.chain3_res = A.in2; Hd.chain3_y1 = A.out; Hd.chain3_y2 = B.out;
            const bool was = A.use_chain;
            net_set_chain_mode(net, (int)i + 1, ch->c1 <= 128 ? 2 : (was ? 1 : 0));
            if (Hd.use_chain3) removed += was ? 1 : 2;
        }
        // ... and in front of a fused-eltwise 1x1 conv that heads no chain (the last block of a stage): conv3x3 + conv1x1
        for (size_t i = 0; i + 1 < ops.size(); ++i) {
            NetOp& Hd = ops[i];
            NetOp& A = ops[i + 1];
            if (A.chain || A.skip || A.kind != OP_CONV || !A.conv || A.conv->d.res_mode != SABER_HIP_RES_ELTWISE || A.in2 < 0 || A.lane ||
                Hd.kind != OP_CONV || !Hd.conv || Hd.chain3 || Hd.chain || Hd.skip || Hd.lane || Hd.in2 >= 0 || A.in != Hd.out)
                continue;
            int readers = 0;
            for (const NetOp& o : ops) readers += (o.in == Hd.out) + (o.in2 == Hd.out);
            if (readers != 1) continue;
            saber_hip_chain* ch = nullptr;
            // 1024: ... and the sibling pair that is the only reader of A's output (the next stage's branch1 / branch2a) joins the launch
            NetOp* P = (flags & 1024) && i + 2 < ops.size() ? &ops[i + 2] : nullptr;
            if (P && (P->kind != OP_CONV_PAIR || P->lane || P->skip || P->in != A.out || !P->conv || !P->conv->pair_src_a || !P->conv->pair_src_b)) P = nullptr;
            if (P) {
                int rd = 0;
                for (const NetOp& o : ops) rd += (o.in == A.out) + (o.in2 == A.out);
                if (rd != 1 || saber_hip_conv2d_chain_create3_pair(Hd.conv, A.conv, const_cast<saber_hip_conv*>(P->conv->pair_src_a),
                                                                   const_cast<saber_hip_conv*>(P->conv->pair_src_b), &ch) != SABER_HIP_OK)
                    P = nullptr;
            }
            if (!P && saber_hip_conv2d_chain_create3(Hd.conv, A.conv, nullptr, &ch) != SABER_HIP_OK) continue;
            net->owned_chains.push_back(ch);
            Hd.chain3 = ch;
            Hd.chain3_res = A.in2; Hd.chain3_y1 = A.out; Hd.chain3_y2 = P ? P->out : -1; Hd.chain3_y3 = P ? P->out2 : -1;
            net_set_chain_mode(net, (int)i + 1, ch->c1 <= 128 ? 2 : 0);
            if (Hd.use_chain3) removed += P ? 2 : 1;
        }
    }
    // ---- 256: runs of 3x3-led C = 256 (or C = 128) chains whose blocks feed each other (ResNet's res4 / res3 stage) -> one persistent launch
    // NOT part of 255: the launch needs every workgroup of an image resident on its XCD at once - fine for one net on the GPU
    // (the latency path), not for several nets in flight on their own streams (two such launches can each hold half of the CUs
    // and wait for the other half: they time out, report an error and fall back; conv_stage_coop.hip)
    if ((flags & 256) && (flags & 32) && !two_lanes) {
        for (size_t i = 0; i + 2 < ops.size();) {
            auto block_ok = [&](size_t j) {
                return j + 2 < ops.size() && ops[j].chain3 && ops[j].chain3->b && !ops[j].stage && ops[j].chain3_y2 >= 0 &&
                       ((ops[j].chain3->c1 == 256 && ops[j].chain3->stage1) || (ops[j].chain3->c1 == 128 && ops[j].chain3->d_stream_stage1.p));
            };
            if (!block_ok(i)) { ++i; continue; }
            std::vector<saber_hip_chain*> run{ops[i].chain3};
            while ((int)run.size() < saber_mi355x::STAGE4_LONG) {
                const size_t p = i + 3 * (run.size() - 1), j = p + 3;
                if (!block_ok(j) || ops[j].chain3->c1 != run[0]->c1 || ops[j].in != ops[p].chain3_y2 || ops[j].chain3_res != ops[p].chain3_y1) break;
                run.push_back(ops[j].chain3);
            }
            if (run.size() >= 2) {
                saber_hip_chain_stage* st = nullptr;
                if (saber_hip_conv2d_stage_create(run.data(), (int)run.size(), &st) == SABER_HIP_OK) {
                    net->owned_stages.push_back(st);
                    ops[i].stage = st;
                    ops[i].stage_n = (int)run.size();
                    int before = 0;
                    for (size_t k = 0; k < run.size(); ++k) before += 3 - (ops[i + 3 * k].skip + ops[i + 3 * k + 1].skip + ops[i + 3 * k + 2].skip);
                    // default until the autotuner has timed both forms: the res4 stage (C = 256) on from batch 4 (an image per XCD: fewer
                    // images leave XCDs idle); the res3 stage (C = 128) off - measured slower than its chain launches
                    const bool on = run[0]->a->d.n >= 4 && run[0]->c1 == 256;
                    net_set_stage(net, (int)i, on);
                    if (on) removed += before - 1;
                }
            }
            i += 3 * run.size();
        }
    }
    // ---- 4096: fc + the softmax over its output -> one launch (fc_small.hip: the last-arriving workgroup normalises) ----------------
    if (flags & 4096) {
        for (size_t i = 0; i + 1 < ops.size(); ++i) {
            NetOp& F = ops[i];
            if (dead[i] || (F.kind != OP_FC && F.kind != OP_FC_Q) || !F.fc || F.out2 >= 0 || F.lane) continue;
            size_t j = i + 1;
            while (j < ops.size() && dead[j]) ++j;
            if (j >= ops.size()) break;
            NetOp& S = ops[j];
            if (S.kind != OP_SOFTMAX || S.in != F.out || S.skip || S.lane || S.p[0] != F.fc->d.m || S.p[1] != F.fc->d.n) continue;
            if (F.kind == OP_FC && F.fc->pre_quant) continue;      // (a quantise-on-entry pre-pass: two launches already)
            if (!fc_softmax_ok(F.fc, F.kind == OP_FC_Q) || fc_softmax_prepare(F.fc) != SABER_HIP_OK) continue;
            F.out2 = S.out;
            F.name = std::string("fc:") + F.fc->conv->algo_name + "+softmax";
            S.skip = true;
            S.name = "softmax_f32 (in the fc launch)";
            ++removed;
        }
    }
    if (net->shared_device) {      // no variant that relies on workgroup placement: split-K off, cooperating chains back to one workgroup per tile
        for (NetOp& o : ops) {
            saber_hip_conv* c = (o.kind == OP_FC || o.kind == OP_FC_Q) ? (o.fc ? o.fc->conv : nullptr) : o.conv;
            if (c) {
                c->no_placement = true;
                if (c->ksplit) { c->ksplit = 0; name_algo(c); }
            }
            for (saber_hip_chain* ch : {o.chain, o.chain3})
                if (ch && (ch->tn == 7 || ch->tn == 15)) (void)saber_hip_conv2d_chain_set_tile(ch, 3);
        }
    }
    // the shared workspace only has to cover the surviving ops
    net->ws_bytes = 0;
    for (const NetOp& o : ops) {
        size_t w = 0;
        if ((o.kind == OP_CONV) && o.conv) w = o.conv->ws_bytes;
        if (o.kind == OP_FC && o.fc) w = saber_hip_fc_workspace_bytes(o.fc);
        if (w > net->ws_bytes) net->ws_bytes = w;
    }
    return removed;
}

