// framework/core/net/mi355x_net_plan.h - Net<MI355X, P, R>::prediction() as ONE executor call.
//
// The reference's prediction() (framework/core/net/net.cpp:417-509) walks ~94 executors per ResNet50 pass: Operator::operator()
// -> BaseFunc::operator() (shape compare, std::vector copies) -> ImplBase::dispatch -> one kernel launch, plus a
// Tensor::record_event per output. On an MI355X a batch-8 INT8 pass is 0.23 ms of GPU time; that loop alone costs ~1 ms of
// host time (profiles/r03/bench_b8_int8.json: net_prediction 0.97 ms) and, being one launch per operator, it can never use the
// executor-level fusions of the MI355X library (fused eltwise epilogues, sibling pairs, conv1x1 chains).
// The reference already has the concept this file implements: Net::fusion_prediction() (net.cpp:511-519) hands the whole
// network to the device as one unit on the MLU / BM targets. For MI355X:
//
//   * after Net::init the op loop is run ONCE under saber_hip_capture_begin / _end (include/saber_hip.h): every Saber impl's
//     dispatch reaches its saber_hip_*_run entry point as usual, which records itself instead of launching. The result is the
//     net's op list as a saber_hip_net - the operators are the Net's own (weights already folded / quantised by
//     WeightsFusion and the adaptors' create()), the tensors are renamed out of the memory planner's aliasing;
//   * the net's input tensors stay where they are (read-before-written addresses are bound to the caller's memory), the graph
//     outputs are bound to the Net's output tensors, every other edge moves into the executor's arena;
//   * saber_hip_net_optimize (the executor's counterpart of fusion_op_register.cpp's pattern catalogue), finalize, autotune
//     (BaseFunc's RUNTIME strategy, base.h:194-247, over the whole list), and the faster of eager launches / hipGraph replay;
//   * prediction() then is: check that the input / output tensors still are the ones captured (address + shape) ->
//     saber_hip_net_run / saber_hip_net_replay -> record + sync the outputs' events (what the Output executors' sync does).
// A shape or address change drops the plan; the next prediction() runs through the op loop under capture again (the impls
// re-create themselves there, base.h:151-161) and builds a new one. Anything the capture cannot express (SABER_HIP_UNIMPL)
// leaves the reference's loop in charge. SABER_MI355X_NET_PLAN=0 in the environment switches the plan off,
// SABER_MI355X_NET_PLAN_TUNE=0 keeps the static kernel selection, SABER_MI355X_NET_PLAN_GRAPH=0|1 forces eager / hipGraph,
// SABER_MI355X_NET_PLAN_STREAM=own gives every Net's plan a stream of its own (all Nets of a device otherwise share the
// Context's compute stream `lane`, as on every target of the reference: Worker threads would serialise on the GPU).
// WHO OWNS THE DEVICE is a property of the plan, decided before kernels are selected (round-4 verdict item 5), not an environment
// variable read at failure time: MI355XNetPlan::shared_device says that other Nets run on the GPU at the same time. It is passed to
// saber_hip_net_optimize as SABER_HIP_NET_SHARED_DEVICE, which keeps every placement-dependent kernel variant (the persistent res4
// stage launch, cooperating-workgroup chains, FP32 split-K through one XCD's L2) out of the static selection, the autotuner and a
// restored selection. Worker<MI355X, ...> (one Net per pool thread) sets the process default from its constructor
// (MI355XNetPlanDefaults::worker_threads: more than one thread -> a stream per Net + shared_device); a single Net on the context's
// compute stream owns the device. The environment variable above remains as an A/B aid and implies shared_device.
// New code of this repository (reference-side glue of the MI355X target).
#ifndef ANAKIN_FRAMEWORK_CORE_NET_MI355X_NET_PLAN_H
#define ANAKIN_FRAMEWORK_CORE_NET_MI355X_NET_PLAN_H

#include <atomic>
#include <chrono>
#include <cstdlib>
#include <string>
#include <vector>

#include "saber_hip.h"

namespace anakin {

// Process-wide defaults a plan takes when it is built (function-local statics: header-only). Set BEFORE the Nets are initialised.
struct MI355XNetPlanDefaults {
    static int& shared_device() { static int v = 0; return v; }   // other Nets / streams run on the device concurrently
    static int& own_stream() { static int v = 0; return v; }      // every plan gets a non-blocking stream of its own
    // FP32 plans keep the executor's STATIC kernel selection (saber_hip_net_optimize flag SABER_HIP_NET_REPRODUCIBLE_FP32): every Net of
    // the process - e.g. the Nets of a Worker's pool threads - then answers a request with the same bits. Off by default (the tuned
    // selection is faster); set before the Nets are initialised. The reference's FP32 answers do not depend on which Worker thread serves.
    static int& reproducible_fp32() { static int v = 0; return v; }
    // Worker<MI355X, P, R>(model, n): n pool threads = n Nets in flight on one GPU (framework/core/net/worker.cpp:59)
    static void worker_threads(int n) {
        if (n > 1) { shared_device() = 1; own_stream() = 1; }
    }
};

// nanoseconds this process spent inside plans (all threads): enqueueing a pass, and waiting for its outputs - for the Worker driver's breakdown
struct MI355XNetPlanStats {
    static std::atomic<long long>& enqueue_ns() { static std::atomic<long long> v{0}; return v; }
    static std::atomic<long long>& wait_ns() { static std::atomic<long long> v{0}; return v; }
    static std::atomic<long long>& runs() { static std::atomic<long long> v{0}; return v; }
};

struct MI355XNetPlan {
    saber_hip_net_t* net = nullptr;
    bool shared_device = MI355XNetPlanDefaults::shared_device() != 0;   // see the header comment; may be set on net.mi355x_plan() before init()
    bool want_own_stream = MI355XNetPlanDefaults::own_stream() != 0;
    int coop_fallbacks = 0;      // cooperative launches that failed a pass and fell back (saber_hip_net_coop_fallbacks); 0 expected
    bool tried = false;          // a build was attempted for the current shapes (successful or not)
    bool enabled = true;
    bool use_graph = false;
    int builds = 0;
    int captured_ops = 0, launches = 0;
    size_t arena_bytes = 0, arena_bytes_full = 0;      // the plan's device footprint: after / before saber_hip_net_compact_arena
    float eager_ms = 0.f, graph_ms = 0.f;
    std::string why;             // why there is no plan (capture refused, switched off ...)
    std::vector<void*> in_t, out_t;      // the Net's input / output Tensor objects (edge tensors: stable for the Net's lifetime)
    std::vector<const void*> in_ptr, out_ptr;
    std::vector<std::vector<int> > in_shape;
    void* stream = nullptr;
    void* own_stream = nullptr;  // SABER_MI355X_NET_PLAN_STREAM=own: a stream of this plan's (serving: one Net per Worker thread, each
    void* own_event = nullptr;   // pass then overlaps the others on the GPU); ordered after the context's compute stream by an event
    void* ctx_stream = nullptr;
    MI355XNetPlan() {}
    MI355XNetPlan(const MI355XNetPlan&) {}                  // a plan belongs to ONE Net: a copy starts without one
    MI355XNetPlan& operator=(const MI355XNetPlan&) { drop(); return *this; }
    void drop() {
        if (net) saber_hip_net_destroy(net);
        net = nullptr;
        tried = false;
    }
    ~MI355XNetPlan() {
        if (net) saber_hip_net_destroy(net);
    }
};

template <typename Ttype, Precision Ptype, OpRunType RunType>
struct MI355XPlanner;      // framework/core/net/mi355x_net_planner.h (a friend of Net)

}  // namespace anakin
#endif
