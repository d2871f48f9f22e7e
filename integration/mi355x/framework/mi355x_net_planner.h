// framework/core/net/mi355x_net_planner.h - builds and runs the MI355XNetPlan of a Net (see mi355x_net_plan.h for the design).
// Included at the end of net.h, after class Net (whose friend MI355XPlanner is).
#ifndef ANAKIN_FRAMEWORK_CORE_NET_MI355X_NET_PLANNER_H
#define ANAKIN_FRAMEWORK_CORE_NET_MI355X_NET_PLANNER_H

namespace anakin {

// primary template: every other target keeps the reference's loop
template <typename Ttype, Precision Ptype, OpRunType RunType>
struct MI355XPlanner {
    static bool run(Net<Ttype, Ptype, RunType>&) { return false; }
    static void prepare(Net<Ttype, Ptype, RunType>&) {}
};

#ifdef USE_MI355X_PLACE
template <Precision Ptype, OpRunType RunType>
struct MI355XPlanner<saber::MI355X, Ptype, RunType> {
    typedef Net<saber::MI355X, Ptype, RunType> net_t;
    typedef saber::TargetWrapper<saber::MI355X> API;

    static bool env_on(const char* name, bool dflt) {
        const char* e = std::getenv(name);
        return e ? e[0] != '0' : dflt;
    }
    static std::vector<int> dims(Tensor4dPtr<saber::MI355X> t) {
        saber::Shape s = t->valid_shape();
        std::vector<int> v;
        for (int i = 0; i < s.dims(); ++i) v.push_back(s[i]);
        return v;
    }

    // Runs the executors once with the C ABI in capture mode and turns the recorded list into a runnable plan.
    static void prepare(net_t& net) {
        MI355XNetPlan& plan = net._mi355x_plan;
        if (plan.net || plan.tried) return;
        plan.tried = true;
        plan.why.clear();
        if (!plan.enabled || !env_on("SABER_MI355X_NET_PLAN", true)) { plan.why = "switched off"; return; }
        if (net._exec_funcs.empty()) { plan.why = "no executors"; return; }
        if (saber_hip_capture_begin() != SABER_HIP_OK) { plan.why = saber_hip_last_error(); return; }
        for (auto& executer : net._exec_funcs) {      // the loop body of prediction(), net.cpp:426-456, nothing is launched
            if (executer.op_name != "Input" && executer.op_name != "Output") {
                executer.infer_shape();
                executer.launch();
            }
        }
        saber_hip_net_t* n = nullptr;
        if (saber_hip_capture_end(&n) != SABER_HIP_OK) {
            plan.why = saber_hip_last_error();
            LOG(WARNING) << "MI355X net plan: " << plan.why << " - Net::prediction keeps the operator loop";
            return;
        }
        plan.captured_ops = saber_hip_net_num_ops(n);
        plan.in_t.clear(); plan.out_t.clear(); plan.in_ptr.clear(); plan.in_shape.clear(); plan.out_ptr.clear();
        for (auto& t : net.get_in_list()) {
            plan.in_t.push_back((void*)t);
            plan.in_ptr.push_back(t->data());
            plan.in_shape.push_back(dims(t));
        }
        bool ok = plan.captured_ops > 0;
        for (auto& t : net.get_out_list()) {          // the graph outputs stay the Net's tensors
            const int id = saber_hip_net_tensor_of_ptr(n, t->data());
            ok = ok && id >= 0 && saber_hip_net_bind_tensor(n, id, t->mutable_data()) == SABER_HIP_OK;
            plan.out_t.push_back((void*)t);
            plan.out_ptr.push_back(t->data());
        }
        plan.stream = (void*)net._exec_funcs[0].ctx_p->get_compute_stream();
        {
            // a stream of this plan's own: the Worker shape (MI355XNetPlanDefaults::worker_threads) - or the A/B switch in the environment.
            // A plan that does not run on the context's shared compute stream runs BESIDE other plans: it does not own the device.
            const char* own = std::getenv("SABER_MI355X_NET_PLAN_STREAM");
            if (own && std::string(own) == "own") plan.want_own_stream = true;
            if (plan.want_own_stream) plan.shared_device = true;
            if (plan.want_own_stream && !plan.own_stream) {
                API::stream_t s;
                API::event_t e;
                API::create_stream_with_flag(&s, 1);
                API::owner_syncs_stream(s);      // run() synchronises the plan's outputs itself: not part of a device-to-host copy's drain
                API::create_event(&e, false);
                plan.own_stream = (void*)s;
                plan.own_event = (void*)e;
            }
        }
        void* const ctx_stream = plan.stream;
        if (plan.own_stream) plan.stream = plan.own_stream;
        plan.ctx_stream = ctx_stream;
        // 256 (a run of res4 blocks as one persistent launch) only for a plan that owns the device; a shared device is declared to the
        // executor (SABER_HIP_NET_SHARED_DEVICE), which then excludes every placement-dependent variant when kernels are SELECTED
        // (SABER_MI355X_NET_STAGE=0 remains as an A/B switch for the owning case)
        const bool stage = !plan.shared_device && env_on("SABER_MI355X_NET_STAGE", true);
        // 512: conv1 + pool1 also run the sibling pair reading pool1 (one launch fewer; pool1's own tensor is not written by the plan)
        const bool stem_pair = env_on("SABER_MI355X_NET_STEM_PAIR", true);
        // 1024: res2c's strided-head chain launch also runs the res3a sibling pair (opt-in: measured no faster than the two launches)
        const bool head_pair = env_on("SABER_MI355X_NET_HEAD_PAIR", false);
        // 4096: the fc and the Softmax over its output as one launch
        const bool fc_softmax = env_on("SABER_MI355X_NET_FC_SOFTMAX", true);
        if (ok) ok = saber_hip_net_optimize(n, 255 | (stage ? 256 : 0) | (stem_pair ? 512 : 0) | (head_pair ? 1024 : 0) | (fc_softmax ? 4096 : 0) |
                                               (plan.shared_device ? SABER_HIP_NET_SHARED_DEVICE : 0) |
                                               (MI355XNetPlanDefaults::reproducible_fp32() ? SABER_HIP_NET_REPRODUCIBLE_FP32 : 0)) >= 0;
        if (ok) ok = saber_hip_net_finalize(n) == SABER_HIP_OK;
        if (ok && plan.builds == 0 && env_on("SABER_MI355X_NET_PLAN_TUNE", true))
            ok = saber_hip_net_autotune(n, plan.stream, 9) == SABER_HIP_OK;
        // A plan that shares its device (the Worker shape: one Net + plan per pool thread) lays its arena out with lifetime aliasing -
        // what the reference's MemoryScheduler does for the Net's own tensors; ResNet50 INT8 batch 8: 72 -> 24 MB per plan. A plan
        // that owns the device keeps every edge in its own slot: measured 3 % faster single-stream (profiles/r06/compact_ab.txt).
        // Inputs / outputs are the Net's tensors either way. SABER_MI355X_NET_PLAN_COMPACT=0|1 forces it.
        if (ok && env_on("SABER_MI355X_NET_PLAN_COMPACT", plan.shared_device)) {
            plan.arena_bytes_full = saber_hip_net_arena_bytes(n);
            ok = saber_hip_net_compact_arena(n, nullptr, 0) == SABER_HIP_OK;
        }
        plan.arena_bytes = saber_hip_net_arena_bytes(n);
        if (!plan.arena_bytes_full) plan.arena_bytes_full = plan.arena_bytes;
        if (ok) ok = choose_launch_form(n, plan);
        if (!ok) {
            plan.why = saber_hip_last_error();
            LOG(WARNING) << "MI355X net plan: " << plan.why << " - Net::prediction keeps the operator loop";
            saber_hip_net_destroy(n);
            return;
        }
        plan.net = n;
        plan.launches = saber_hip_net_num_launches(n);
        ++plan.builds;
        LOG(INFO) << "MI355X net plan: " << net._exec_funcs.size() << " executors -> " << plan.captured_ops << " captured ops -> "
                  << plan.launches << " launches per prediction (" << (plan.use_graph ? "hipGraph replay" : "eager") << "), arena "
                  << (plan.arena_bytes >> 20) << " MiB (every edge materialised: " << (plan.arena_bytes_full >> 20) << " MiB)";
    }

    // eager launches from the C++ op loop against one hipGraph replay: whichever is faster on this host
    static bool choose_launch_form(saber_hip_net_t* n, MI355XNetPlan& plan) {
        const char* force = std::getenv("SABER_MI355X_NET_PLAN_GRAPH");
        if (force && force[0] == '0') { plan.use_graph = false; return true; }
        if (saber_hip_net_capture(n, plan.stream) != SABER_HIP_OK) { plan.use_graph = false; return true; }
        if (force) { plan.use_graph = true; return true; }
        API::event_t e0, e1;
        API::create_event(&e0, true);
        API::create_event(&e1, true);
        float ms[2] = {0.f, 0.f};
        bool ok = true;
        for (int form = 0; form < 2 && ok; ++form) {
            for (int it = 0; it < 3 && ok; ++it)
                ok = (form ? saber_hip_net_replay(n, plan.stream) : saber_hip_net_run(n, plan.stream)) == SABER_HIP_OK;
            API::record_event(e0, (API::stream_t)plan.stream);
            for (int it = 0; it < 20 && ok; ++it)
                ok = (form ? saber_hip_net_replay(n, plan.stream) : saber_hip_net_run(n, plan.stream)) == SABER_HIP_OK;
            API::record_event(e1, (API::stream_t)plan.stream);
            API::sync_event(e1);
            (void)hipEventElapsedTime(&ms[form], e0, e1);
        }
        API::destroy_event(e0);
        API::destroy_event(e1);
        plan.eager_ms = ms[0] / 20.f;
        plan.graph_ms = ms[1] / 20.f;
        plan.use_graph = ok && ms[1] < ms[0];
        return ok;
    }

    static bool still_valid(net_t& net, MI355XNetPlan& plan) {
        for (size_t i = 0; i < plan.in_t.size(); ++i) {
            Tensor4dPtr<saber::MI355X> t = (Tensor4dPtr<saber::MI355X>)plan.in_t[i];
            if (t->data() != plan.in_ptr[i]) return false;
            const saber::Shape s = t->valid_shape();
            if ((size_t)s.dims() != plan.in_shape[i].size()) return false;
            for (int d = 0; d < s.dims(); ++d)
                if (s[d] != plan.in_shape[i][d]) return false;
        }
        for (size_t i = 0; i < plan.out_t.size(); ++i)
            if (((Tensor4dPtr<saber::MI355X>)plan.out_t[i])->data() != plan.out_ptr[i]) return false;
        return true;
    }

    // true: the forward pass has been enqueued and its outputs are complete (prediction() returns)
    static bool run(net_t& net) {
        MI355XNetPlan& plan = net._mi355x_plan;
        if (plan.net && !still_valid(net, plan)) plan.drop();      // reshaped / re-allocated: the impls re-create under the next capture
        if (!plan.net) {
            if (plan.tried) return false;
            prepare(net);
            if (!plan.net) return false;
        }
        if (plan.own_stream) {      // after whatever the caller queued on the context's stream (asynchronous input copies)
            API::record_event((API::event_t)plan.own_event, (API::stream_t)plan.ctx_stream);
            API::sync_stream((API::event_t)plan.own_event, (API::stream_t)plan.own_stream);
        }
        for (int attempt = 0; attempt < 2; ++attempt) {
            const auto t0 = std::chrono::steady_clock::now();
            const int rc = plan.use_graph ? saber_hip_net_replay(plan.net, plan.stream) : saber_hip_net_run(plan.net, plan.stream);
            CHECK_EQ(rc, (int)SABER_HIP_OK) << "MI355X net plan: " << saber_hip_last_error();
            const auto t1 = std::chrono::steady_clock::now();
            for (void* p : plan.out_t) {              // what the Output executors' ins[i]->sync() does (net.cpp:427-432)
                Tensor4dPtr<saber::MI355X> t = (Tensor4dPtr<saber::MI355X>)p;
                t->record_event((API::stream_t)plan.stream);
                t->sync();
            }
            MI355XNetPlanStats::enqueue_ns() += std::chrono::duration_cast<std::chrono::nanoseconds>(t1 - t0).count();
            MI355XNetPlanStats::wait_ns() += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t1).count();
            ++MI355XNetPlanStats::runs();
            // the pass has completed: a cooperative launch that could not (another kernel held the CUs its workgroups wait for) reports
            // itself here, those sites now launch block by block, and the pass runs once more - the caller never sees its outputs
            if (saber_hip_net_status(plan.net) == SABER_HIP_OK) break;
            plan.coop_fallbacks = saber_hip_net_coop_fallbacks(plan.net);
            LOG(WARNING) << "MI355X net plan: " << saber_hip_last_error();
            CHECK_EQ(attempt, 0) << "MI355X net plan: the fallback launches failed as well";
            plan.use_graph = false;
            plan.launches = saber_hip_net_num_launches(plan.net);
        }
        return true;
    }
};
#endif  // USE_MI355X_PLACE

}  // namespace anakin
#endif
