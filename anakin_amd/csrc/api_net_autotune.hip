// anakin_amd/csrc/api_net_autotune.hip - whole-net autotuner and the kernel-selection save / restore entry points.
#include "api_internal.h"

// The kernel selection of op `index` in saber_hip_conv2d_get_tile / set_tile encoding (0 for ops without one): lets a caller
// carry an autotuned selection from one process to the next (bench.py --tune-cache: every profiling pass runs the SAME
// kernels).
static saber_hip_conv* net_op_conv(saber_hip_net* net, int index) {
    if (index < 0 || index >= (int)net->ops.size()) return nullptr;
    NetOp& o = net->ops[index];
    if (o.kind == OP_CONV || o.kind == OP_CONV_PAIR) return o.conv;
    if (o.kind == OP_FC || o.kind == OP_FC_Q) return o.fc ? o.fc->conv : nullptr;
    return nullptr;
}
// bits 0..23: saber_hip_conv2d_get_tile of the op; chain heads add bit 28 (a chain decision is recorded) and the chain's
// pixel fragments in bits 24..27 (0: run as two launches)
int saber_hip_net_get_choice(saber_hip_net_t* net, int index) {
    saber_hip_conv* c = net_op_conv(net, index);
    int choice = (c && !c->pool_fused && c->algo <= ALGO_IGEMM_F32) ? saber_hip_conv2d_get_tile(c) : 0;
    if (c && net->ops[index].chain) choice |= (1 << 28) | ((net->ops[index].use_chain ? net->ops[index].chain->tn : 0) << 24);
    if (c && net->ops[index].chain3) choice |= (1 << 29) | ((net->ops[index].use_chain3 ? net->ops[index].chain3->tn : 0) << 24);
    if (c && net->ops[index].stage && net->ops[index].use_stage) choice |= 1 << 30;      // this op launches its whole stage
    return choice;
}
int saber_hip_net_stage_blocks(const saber_hip_net_t* net, int index) {
    if (!net || index < 0 || index >= (int)net->ops.size()) return 0;
    return net->ops[index].stage ? net->ops[index].stage_n : 0;
}
int saber_hip_net_set_choice(saber_hip_net_t* net, int index, int choice) {
    saber_hip_conv* c = net_op_conv(net, index);
    if (!c || !choice || c->pool_fused || c->algo > ALGO_IGEMM_F32) return SABER_HIP_OK;
    if (net->reproducible_fp32 && !c->is_i8) return SABER_HIP_OK;      // flag 8192: a restored selection does not move FP32 ops either
    if (net->ops[index].kind == OP_CONV_PAIR && index > 0 && net->ops[index - 1].stem_pair) return SABER_HIP_OK;      // no kernel of its own (flag 512)
    int chain_bits = (choice >> 24) & 63;
    const bool stage_on = ((choice >> 30) & 1) && !net->shared_device;
    choice &= 0xffffff;
    if (net->shared_device) {      // a selection tuned on a net that owned its device: placement-dependent variants are mapped to their plain forms
        if (((choice >> 16) & 0xff) == 11) choice &= ~(0xf << 12);                           // bf16-plane kernel: split-K off
        if ((chain_bits & 15) == 7 || (chain_bits & 15) == 15) chain_bits = (chain_bits & ~15) | 3;   // cooperating chains -> one workgroup per tile, 8 waves
    }
    int rc = choice ? saber_hip_conv2d_set_tile(c, choice) : SABER_HIP_OK;
    if (rc) return rc;
    NetOp& o = net->ops[index];
    o.name = std::string(o.kind == OP_FC || o.kind == OP_FC_Q ? "fc:" : "conv:") + c->algo_name;
    if ((o.kind == OP_FC || o.kind == OP_FC_Q) && o.out2 >= 0) o.name += fc_softmax_ok(o.fc, o.kind == OP_FC_Q) ? "+softmax" : " | softmax_f32";
    // chain decisions: a 3x3 head (bit 29) is restored before its chain head (bit 28, the next op): set_choices runs in op order
    if (o.chain3 && (chain_bits & 32) && index + 1 < (int)net->ops.size()) {
        const int tn = chain_bits & 15;
        if (tn && (rc = saber_hip_conv2d_chain_set_tile(o.chain3, tn)) != SABER_HIP_OK) return rc;
        net_set_chain_mode(net, index + 1, tn ? 2 : net_chain_mode(net, index + 1) == 2 ? 1 : net_chain_mode(net, index + 1));
    } else if (o.chain && (chain_bits & 16) && index + 1 < (int)net->ops.size()) {
        const int tn = chain_bits & 15;
        if (tn && (rc = saber_hip_conv2d_chain_set_tile(o.chain, tn)) != SABER_HIP_OK) return rc;
        net_set_chain_mode(net, index, net_chain_mode(net, index) == 2 ? 2 : (tn ? 1 : 0));   // (also restores the names)
    }
    if (o.stage) net_set_stage(net, index, stage_on);      // (a stage head comes before its blocks: set_choices runs in op order)
    if (o.skip) o.name = (o.chain3 && o.use_chain3) ? "conv:(in the stage launch)" : "conv:(in the chain launch)";
    if (o.skip && o.kind == OP_CONV_PAIR) o.name = (index > 0 && net->ops[index - 1].stem_pair) ? "conv:(in the stem launch)" : "conv:(in the chain launch)";
    if (o.stem_pair) o.name = stem_pair_name(o);
    if (net->exec) {
        (void)hipGraphExecDestroy(net->exec);
        (void)hipGraphDestroy(net->graph);
        net->exec = nullptr;
        net->graph = nullptr;
    }
    return SABER_HIP_OK;
}

// End-to-end refinement after the per-op tuning: an implicit-GEMM kernel function used by exactly ONE op of the pass is a
// body of code fetched cold once per forward for one launch. For each such op try the functions other ops (of the same
// epilogue class) already run and keep a switch only if the WHOLE forward pass gets faster by >= 0.4 % against two
// measurements of the incumbent - which it does when cold code is expensive on this box (pool's slow boxes: 3-9 us per
// first use) and not when it is cheap (0.3-0.6 us).
static int net_consolidate_kernels(saber_hip_net* net, hipStream_t s) {
    if (net->has_side) return SABER_HIP_OK;
    if (const char* e = std::getenv("SABER_HIP_NO_CONSOLIDATE"))
        if (e[0] == '1') return SABER_HIP_OK;
    struct Site { int op; unsigned long long key; ConvChoice choice; };
    auto conv_of = [&](const NetOp& o) -> saber_hip_conv* {
        if (o.skip || (o.chain && o.use_chain) || (o.chain3 && o.use_chain3) || (o.stage && o.use_stage)) return nullptr;
        if (o.kind != OP_CONV && o.kind != OP_CONV_PAIR) return nullptr;
        return (o.conv && !o.conv->pool_fused && o.conv->algo <= ALGO_IGEMM_F32) ? o.conv : nullptr;
    };
    auto collect = [&]() {
        std::vector<Site> v;
        for (int i = 0; i < (int)net->ops.size(); ++i)
            if (saber_hip_conv* c = conv_of(net->ops[i])) v.push_back({i, kernel_key(c, get_choice(c)), get_choice(c)});
        return v;
    };
    EventPair ev;
    HIP_TRY(ev.init());
    auto forward_ms = [&](float* ms) -> int {   // 3 warm-up + 30 timed eager forwards
        int rc = 0;
        for (int i = 0; i < 3 && !rc; ++i) rc = saber_hip_net_run(net, s);
        if (rc) return rc;
        HIP_TRY(hipEventRecord(ev.e0, s));
        for (int i = 0; i < 30 && !rc; ++i) rc = saber_hip_net_run(net, s);
        HIP_TRY(hipEventRecord(ev.e1, s));
        HIP_TRY(hipEventSynchronize(ev.e1));
        HIP_TRY(hipEventElapsedTime(ms, ev.e0, ev.e1));
        return rc;
    };
    std::vector<Site> sites = collect();
    for (size_t si = 0; si < sites.size(); ++si) {
        const Site cur = sites[si];
        if ((cur.key >> 8 & 0xff) != 5) continue;                      // implicit-GEMM tile kernels only
        int uses = 0;
        for (const Site& t : sites) uses += t.key == cur.key;
        if (uses != 1) continue;
        saber_hip_conv* c = conv_of(net->ops[cur.op]);
        std::vector<Site> alts;                                        // distinct functions of the same class in use elsewhere
        for (const Site& t : sites) {
            if (t.op == cur.op || (t.key & 0xff) != (cur.key & 0xff) || (t.key >> 8 & 0xff) != 5 || t.key == cur.key) continue;
            if ((net->ops[t.op].kind == OP_CONV_PAIR) != (net->ops[cur.op].kind == OP_CONV_PAIR)) continue;
            bool dup = false;
            for (const Site& a : alts) dup |= a.key == t.key;
            if (!dup) alts.push_back(t);
        }
        if (alts.empty()) continue;
        float base = 0.f, base2 = 0.f;
        int rc = forward_ms(&base);
        if (rc) return rc;
        ConvChoice best_c = cur.choice;
        float best = base;
        for (const Site& a : alts) {
            ConvChoice cc = cur.choice;
            cc.tile = a.choice.tile; cc.ks = a.choice.ks; cc.dma = a.choice.dma;
            set_choice(c, cc);
            float ms = 0.f;
            if (forward_ms(&ms) != SABER_HIP_OK) { (void)hipGetLastError(); continue; }   // not launchable for this shape
            if (ms < best) { best = ms; best_c = cc; }
        }
        set_choice(c, cur.choice);
        if (best < base * 0.996f) {                                    // confirm against a second look at the incumbent
            rc = forward_ms(&base2);
            if (rc) return rc;
            if (best < base2 * 0.996f) {
                set_choice(c, best_c);
                name_algo(c);
                net->ops[cur.op].name = std::string("conv:") + c->algo_name;
                sites = collect();
            }
        }
    }
    return saber_hip_net_run(net, s);   // every tensor holds the final selection's result
}

int saber_hip_net_autotune(saber_hip_net_t* net, saber_hip_stream_t stream, int iters) {
    if (net->inplace_external)
        return fail(SABER_HIP_INVALID_VALUE, "autotune: an in-place sum of this captured list accumulates into a tensor of the caller's that the "
                    "list itself never writes - timing passes would change it (run the pass that writes it inside the capture)");
    if (net->exec) {   // a captured graph holds the OLD kernel selections: drop it, the caller captures again
        (void)hipGraphExecDestroy(net->exec);
        (void)hipGraphDestroy(net->graph);
        net->exec = nullptr;
        net->graph = nullptr;
    }
    auto T = [&](int id) -> void* { return net->ptr(id); };
    ColdScope scope;
    // flush size between timed repetitions: a net whose tensor arena exceeds the 256 MB Infinity Cache finds its weights in no cache
    // from one forward pass to the next - flush that much (up to 512 MB); smaller nets keep the 64 MB L2-only flush
    HIP_TRY(scope.enter(iters < 7 ? 7 : (iters > 15 ? 15 : iters), net->arena_bytes > ((size_t)256 << 20) ? net->arena_bytes : 0));
    std::vector<unsigned long long> used_kernels;
    struct UsedScope {
        UsedScope(std::vector<unsigned long long>* v) { g_used_kernels = g_cold ? v : nullptr; }
        ~UsedScope() { g_used_kernels = nullptr; }
    } used_scope(&used_kernels);
    for (NetOp& o : net->ops) {
        // a pair that runs inside the stem launch (flag 512) has no kernel of its own (one absorbed by a chain launch - flag 1024 -
        // keeps its own for the mode in which the chain is off: it is tuned like any other)
        if (o.kind == OP_CONV_PAIR && o.skip && &o != net->ops.data() && (&o)[-1].stem_pair) continue;
        if (net->reproducible_fp32 && o.conv && !o.conv->is_i8) continue;      // flag 8192: FP32 convs / pairs keep the static selection
        if (net->reproducible_fp32 && o.fc && o.fc->conv && !o.fc->conv->is_i8) continue;
        if (o.kind == OP_CONV_PAIR) {
            int rc = saber_hip_conv2d_autotune_pair(o.conv, T(o.in), T(o.out), T(o.out2), stream, iters);
            if (rc) return rc;
            o.name = std::string("conv:") + o.conv->algo_name;
            continue;
        }
        saber_hip_conv* c = o.kind == OP_CONV ? o.conv : ((o.kind == OP_FC || o.kind == OP_FC_Q) ? o.fc->conv : nullptr);
        if (!c) continue;
        const void* xin = T(o.in);
        if (o.kind == OP_FC && o.fc->pre_quant) {   // the GEMM reads the quantised copy in the workspace
            int rq = net_launch(net, o, (hipStream_t)stream);
            if (rq) return rq;
            xin = net->arena + net->ws_off;
        }
        int rc = saber_hip_conv2d_autotune(c, xin, T(o.out), T(o.in2), net->arena + net->ws_off, stream, iters);
        if (rc) return rc;
        o.name = std::string(o.kind == OP_CONV ? "conv:" : "fc:") + c->algo_name;
        if ((o.kind == OP_FC || o.kind == OP_FC_Q) && o.out2 >= 0) o.name += fc_softmax_ok(o.fc, o.kind == OP_FC_Q) ? "+softmax" : " | softmax_f32";
        if (o.stem_pair) o.name = stem_pair_name(o);
    }
    // conv1x1 chains: the tuned separate launches against the chain launch (every pixel-tile size) and, where the block's
    // 3x3 conv can lead the chain, against that single launch too - on the real tensors
    for (size_t i = 0; i < net->ops.size(); ++i)
        if (net->ops[i].stage) net_set_stage(net, (int)i, false);      // (block by block first; the stages after this loop)
    for (size_t i = 0; i < net->ops.size(); ++i) {
        NetOp& A = net->ops[i];
        const int ia = (int)i;
        NetOp* H = (ia > 0 && net->ops[ia - 1].chain3) ? &net->ops[ia - 1] : nullptr;
        if (!A.chain && !H) continue;
        const int first = H ? ia - 1 : ia;
        const int last = (A.chain || (H && H->chain3->b2)) ? ia + 1 : ia;      // (a head + pair chain also replaces the pair op behind A)
        hipStream_t s = (hipStream_t)stream;
        auto run_all = [&]() -> int {
            int rc = 0;
            for (int k = first; k <= last; ++k) rc |= net_launch(net, net->ops[k], s);
            return rc;
        };
        auto timed = [&](float* us) -> int {
            if (g_cold) {
                *us = g_cold->run(s, run_all);
                return *us < 0.f ? SABER_HIP_RUNTIME_ERROR : SABER_HIP_OK;
            }
            EventPair ev;                 // SABER_HIP_AUTOTUNE_WARM: 20 back-to-back repetitions
            HIP_TRY(ev.init());
            int rc = run_all();
            if (rc) return rc;
            HIP_TRY(hipEventRecord(ev.e0, s));
            for (int it = 0; it < 20; ++it) rc |= run_all();
            HIP_TRY(hipEventRecord(ev.e1, s));
            HIP_TRY(hipEventSynchronize(ev.e1));
            HIP_TRY(hipEventElapsedTime(us, ev.e0, ev.e1));
            return rc;
        };
        float best = 0.f;
        int best_mode = 0, best_tn = 0;
        net_set_chain_mode(net, ia, 0);
        int rc = timed(&best);
        if (rc) return rc;
        const int c1 = A.chain ? A.chain->c1 : H->chain3->c1;
        const int tns[6] = {c1 == 64 ? 4 : (c1 == 128 ? 2 : 1), c1 == 64 ? 2 : (c1 == 128 ? 1 : 9), c1 == 256 ? 11 : (c1 == 128 ? 6 : 0),
                            c1 == 128 ? 5 : (c1 == 256 ? 3 : 0),       // (C = 256, code 3: the 3x3-led forms with 8 waves; refused elsewhere)
                            c1 == 256 ? 7 : 0,                         // (code 7: two cooperating workgroups per tile, 3x3-led with a second 1x1 conv)
                            c1 == 256 ? 15 : 0};                       // (code 15: four per tile of two rows)
        for (int mode = A.chain ? 1 : 2; mode <= (H ? 2 : 1); ++mode) {
            saber_hip_chain* ch = mode == 2 ? H->chain3 : A.chain;
            for (int tn : tns) {
                if (!tn) continue;
                if (net->shared_device && (tn == 7 || tn == 15)) continue;      // cooperating workgroups: not on a shared device
                float ms = 0.f;
                if (saber_hip_conv2d_chain_set_tile(ch, tn) != SABER_HIP_OK) continue;
                net_set_chain_mode(net, ia, mode);
                if (timed(&ms) != SABER_HIP_OK) continue;
                if (ms < best) { best = ms; best_mode = mode; best_tn = tn; }
            }
        }
        if (best_mode) (void)saber_hip_conv2d_chain_set_tile(best_mode == 2 ? H->chain3 : A.chain, best_tn);
        net_set_chain_mode(net, ia, best_mode);
        rc = run_all();   // every written output holds the selected form's result
        if (rc) return rc;
    }
    // stages: the blocks' tuned launches one after the other against the one persistent launch
    for (size_t i = 0; i < net->ops.size(); ++i) {
        NetOp& H0 = net->ops[i];
        if (!H0.stage || net->shared_device) continue;
        const int first = (int)i, last = first + 3 * H0.stage_n - 1;
        hipStream_t s = (hipStream_t)stream;
        auto run_all = [&]() -> int {
            int rc = 0;
            for (int k = first; k <= last; ++k) rc |= net_launch(net, net->ops[k], s);
            return rc;
        };
        auto timed = [&](float* us) -> int {
            if (g_cold) {
                *us = g_cold->run(s, run_all);
                return *us < 0.f ? SABER_HIP_RUNTIME_ERROR : SABER_HIP_OK;
            }
            EventPair ev;
            HIP_TRY(ev.init());
            int rc = run_all();
            if (rc) return rc;
            HIP_TRY(hipEventRecord(ev.e0, s));
            for (int it = 0; it < 20; ++it) rc |= run_all();
            HIP_TRY(hipEventRecord(ev.e1, s));
            HIP_TRY(hipEventSynchronize(ev.e1));
            HIP_TRY(hipEventElapsedTime(us, ev.e0, ev.e1));
            return rc;
        };
        std::vector<int> modes(H0.stage_n), tns(H0.stage_n);
        for (int k = 0; k < H0.stage_n; ++k) {
            modes[k] = net_chain_mode(net, first + 3 * k + 1);
            tns[k] = net->ops[first + 3 * k].chain3->tn;
        }
        float sep = 0.f, one = 0.f;
        int rc = timed(&sep);
        if (rc) return rc;
        net_set_stage(net, first, true);
        const bool ok = timed(&one) == SABER_HIP_OK && hipStreamSynchronize(s) == hipSuccess && !*(volatile unsigned*)H0.stage->h_err;
        if (!ok || one >= sep) {
            *(volatile unsigned*)H0.stage->h_err = 0u;
            net_set_stage(net, first, false);
            for (int k = 0; k < H0.stage_n; ++k) {
                (void)saber_hip_conv2d_chain_set_tile(net->ops[first + 3 * k].chain3, tns[k]);
                net_set_chain_mode(net, first + 3 * k + 1, modes[k]);
            }
        }
        rc = run_all();   // every written output holds the selected form's result
        if (rc) return rc;
    }
    return g_cold ? net_consolidate_kernels(net, (hipStream_t)stream) : SABER_HIP_OK;
}
