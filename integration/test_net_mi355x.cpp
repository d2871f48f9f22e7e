// integration/test_net_mi355x.cpp — the reference's OWN framework driving the MI355X target:
//
//     Graph<MI355X, P>::AddOp / AddOpAttr / Freeze          framework/graph/graph.h:97-139, graph.cpp:88-330
//         (a Caffe-topology network given as ORIGINAL operators: Convolution, BatchNorm, Scale, ReLU, Pooling, Eltwise,
//          Dense, Softmax — the programmatic route of test/framework/net/net_subgraph_test.cpp:34-63; no protobuf)
//     Graph::SetOpPrec / SetVarScale                         graph.cpp:108-180   (what load_calibrator_config would set)
//     Graph::Optimize()                                      graph.cpp:351-477: the fusion pass
//         (fusion_op_register.cpp:45-175), graph_strategy::apply_stride_up (optimize_strategy.h:41-47), Scheduler,
//         ParallScheduler, MemoryScheduler (buffer aliasing) — ALL the reference's own, unmodified code
//     Net<MI355X, P>::init(graph, auto_config_layout)        framework/core/net/net.cpp:215-408: calibrator_op per node,
//         BindParam / InitParam (WeightsFusion::update_weights folds BN+Scale), edge dtype / layout / scale rules
//         (calibrator_parse.cpp), InferShape, Init -> BaseFunc::init -> Saber*<MI355X, OpDtype>::init, init_memory
//     Net::prediction()                                      net.cpp:417-509 -> Operator::operator() -> BaseFunc::operator()
//         -> Saber*<MI355X,...>::dispatch -> integration/saber_mi355x_adaptor.h -> include/saber_hip.h -> HIP kernels
//
// The program is a file-driven harness (the Python test writes the model and checks the dumps against the oracle):
//     test_net_mi355x.bin <model.txt> <weights.bin> <input.bin> <outdir> [iters | dry]
// model.txt (one record per line; the network BEFORE any fusion):
//     precision int8|fp32
//     input  <name> n c h w
//     conv   <name> <src> cin cout k stride pad relu(0|1) bn(0|1)     weights.bin: w[cout,cin,k,k]; bn=0: bias[cout];
//                                                                      bn=1: mean[cout] var[cout] gamma[cout] beta[cout]
//     pool   <name> <src> MAX|AVG win stride pad global(0|1)
//     eltwise <name> <a> <b> relu(0|1) coeff_a coeff_b
//     fc     <name> <src> cin cout relu(0|1)                            weights.bin: w[cout,cin] bias[cout]
//     softmax <name> <src>
//     prec   <node> int8|fp32          (after Freeze: Graph::SetOpPrec)
//     precsplit <node> int8|fp32       (the Split node Graph::Freeze inserts behind <node>'s output when it has >1 readers)
//     scale  <node> <float>            (Graph::SetVarScale on <node>'s output variable)
//     calibrator <net_config.txt> <calibrator.txt>   (instead of prec / scale records: Graph::load_calibrator_config reads both
//                                      text files - integration/net_model.py: calibrator_files writes them from the topology)
// Outputs in <outdir>:  plan.txt (the captured plan behind prediction(): ops / launches / launch form), out_<o>_oploop.bin
// (prediction() with the plan switched off), oplist.txt (the op list the reference's optimiser produced, with every edge's dtype / layout /
// shape / scale / sharing), out_<graph output>.bin (after Net::prediction()), step_<i>_<j>.bin (output j of op i, dumped
// right after that op ran in a second, op-by-op pass — the memory planner aliases edge buffers, so intermediate edges only
// exist at that moment), timing.txt (Net::prediction() wall time per call, hipEvents through SaberTimer<MI355X>).
//
// The op-by-op pass reads Net's private `_exec_funcs`; this TU (and only it) is compiled with the access specifier opened.
// TEST INFRASTRUCTURE of integration/; built by integration/build_mi355x_test.sh, run on the GPU by tests/test_gpu_net.py.
#include "anakin_config.h"

#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <vector>
#include <unordered_map>
#include <unordered_set>
#include <functional>
#include <mutex>
#include <thread>
#include <algorithm>

#include "saber/core/context.h"
#include "saber/core/tensor.h"
#include "saber/funcs/timer.h"
#include "framework/graph/graph.h"
#include "framework/core/operator/operator.h"
#include "framework/core/net/calibrator_parse.h"
#define private public
#include "framework/core/net/net.h"
#undef private

using namespace anakin;
using namespace anakin::saber;
using anakin::graph::Graph;

static std::vector<float> g_weights;
static size_t g_wpos = 0;

static const float* take(size_t n) {
    if (g_wpos + n > g_weights.size()) { fprintf(stderr, "weights.bin too short\n"); exit(2); }
    const float* p = g_weights.data() + g_wpos;
    g_wpos += n;
    return p;
}

// A weight block as the model parser creates it: host + device copy, registered with the graph's global weight guard
// (Graph::RegistBlock, graph.h:108; GraphGlobalMem::apply — WeightsFusion::update_weights, trans_weights — looks it up).
static PBlock<MI355X> block(const std::vector<int>& shape4, const float* src) {
    Shape sh(shape4);
    PBlock<MI355X> b(sh);
    memcpy(b.h_tensor().mutable_data(), src, sizeof(float) * sh.count());
    b.d_tensor().set_shape(sh);
    b.d_tensor().copy_from(b.h_tensor());     // TargetWrapper<MI355X>::sync_memcpy(..., __HtoD)
    graph::GraphGlobalMem<MI355X>::Global().register_block(&b);
    return b;
}

static const char* dtype_name(DataType t) {
    return t == AK_FLOAT ? "f32" : (t == AK_INT8 ? "s8" : (t == AK_UINT8 ? "u8" : (t == AK_INT32 ? "s32" : "?")));
}
static const char* layout_name(LayoutType l) { return l == Layout_NHWC ? "nhwc" : (l == Layout_NCHW ? "nchw" : "other"); }

struct Record { std::vector<std::string> f; };

template <Precision P>
static int run(const std::vector<Record>& recs, const std::vector<float>& input, const std::string& outdir, int iters) {
    typedef Graph<MI355X, P> graph_t;
    std::unique_ptr<graph_t> graph(new graph_t());
    std::string in_name;
    std::vector<int> in_shape;
    std::vector<std::pair<std::string, std::string> > precs;
    std::vector<std::pair<std::string, float> > scales;
    std::string cal_config, cal_table;      // `calibrator` record: Graph::load_calibrator_config instead of SetOpPrec / SetVarScale
    auto I = [](const std::string& s) { return atoi(s.c_str()); };
    // variable names live in their own namespace but Graph::Freeze names the Input / Output / Split nodes after them
    // (graph.cpp:237-296), so a layer's output variable must not be called like the layer's node
    auto V = [&](const std::string& layer) { return layer == in_name ? layer : layer + "_out"; };

    for (const Record& r : recs) {
        const std::vector<std::string>& f = r.f;
        const std::string& kind = f[0];
        if (kind == "input") {
            in_name = f[1];
            in_shape = {I(f[2]), I(f[3]), I(f[4]), I(f[5])};
        } else if (kind == "conv") {
            const std::string name = f[1], src = V(f[2]);
            const int cin = I(f[3]), cout = I(f[4]), k = I(f[5]), stride = I(f[6]), pad = I(f[7]);
            const bool relu = I(f[8]) != 0, bn = I(f[9]) != 0;
            std::string top = name;
            graph->AddOp(name, "Convolution", {src}, {bn || relu ? name + "_conv" : V(name)});
            graph->AddOpAttr(name, "group", 1);
            graph->AddOpAttr(name, "bias_term", !bn);
            graph->AddOpAttr(name, "padding", PTuple<int>(pad, pad));
            graph->AddOpAttr(name, "strides", PTuple<int>(stride, stride));
            graph->AddOpAttr(name, "dilation_rate", PTuple<int>(1, 1));
            graph->AddOpAttr(name, "filter_num", cout);
            graph->AddOpAttr(name, "kernel_size", PTuple<int>(k, k));
            graph->AddOpAttr(name, "axis", 1);
            graph->AddOpAttr(name, "weight_1", block({cout, cin, k, k}, take((size_t)cout * cin * k * k)));
            std::string cur = name + "_conv";
            if (!bn) {
                graph->AddOpAttr(name, "weight_2", block({1, cout, 1, 1}, take(cout)));
            } else {
                // Caffe: BatchNorm (mean, variance, moving-average factor) then Scale (gamma, beta)
                const std::string bnn = "bn_" + name, scn = "scale_" + name;
                graph->AddOp(bnn, "BatchNorm", {cur}, {name + "_bn"});
                graph->AddOpAttr(bnn, "epsilon", 1e-5f);
                graph->AddOpAttr(bnn, "momentum", 0.999f);
                graph->AddOpAttr(bnn, "weight_1", block({1, cout, 1, 1}, take(cout)));
                graph->AddOpAttr(bnn, "weight_2", block({1, cout, 1, 1}, take(cout)));
                const float one = 1.f;
                graph->AddOpAttr(bnn, "weight_3", block({1, 1, 1, 1}, &one));
                const std::string sc_out = relu ? name + "_scale" : V(name);
                graph->AddOp(scn, "Scale", {name + "_bn"}, {sc_out});
                graph->AddOpAttr(scn, "num_axes", 1);
                graph->AddOpAttr(scn, "bias_term", true);
                graph->AddOpAttr(scn, "axis", 1);
                graph->AddOpAttr(scn, "weight_1", block({1, cout, 1, 1}, take(cout)));
                graph->AddOpAttr(scn, "weight_2", block({1, cout, 1, 1}, take(cout)));
                cur = sc_out;
            }
            if (relu) {
                const std::string rn = name + "_relu";
                graph->AddOp(rn, "ReLU", {cur}, {V(name)});
                graph->AddOpAttr(rn, "alpha", 0.0f);
            }
        } else if (kind == "pool") {
            const std::string name = f[1];
            graph->AddOp(name, "Pooling", {V(f[2])}, {V(name)});
            graph->AddOpAttr(name, "method", f[3]);
            graph->AddOpAttr(name, "pool_size", PTuple<int>(I(f[4]), I(f[4])));
            graph->AddOpAttr(name, "strides", PTuple<int>(I(f[5]), I(f[5])));
            graph->AddOpAttr(name, "padding", PTuple<int>(I(f[6]), I(f[6])));
            graph->AddOpAttr(name, "global_pooling", I(f[7]) != 0);
            graph->AddOpAttr(name, "cmp_out_shape_floor_as_conv", false);     // Caffe: ceil mode
        } else if (kind == "eltwise") {
            const std::string name = f[1];
            const bool relu = I(f[4]) != 0;
            graph->AddOp(name, "Eltwise", {V(f[2]), V(f[3])}, {relu ? name + "_sum" : V(name)});
            graph->AddOpAttr(name, "type", std::string("Add"));
            graph->AddOpAttr(name, "coeff", PTuple<float>((float)atof(f[5].c_str()), (float)atof(f[6].c_str())));
            if (relu) {
                graph->AddOp(name + "_relu", "ReLU", {name + "_sum"}, {V(name)});
                graph->AddOpAttr(name + "_relu", "alpha", 0.0f);
            }
        } else if (kind == "fc") {
            const std::string name = f[1];
            const int cin = I(f[3]), cout = I(f[4]);
            const bool relu = I(f[5]) != 0;
            graph->AddOp(name, "Dense", {V(f[2])}, {relu ? name + "_fc" : V(name)});
            graph->AddOpAttr(name, "out_dim", cout);
            graph->AddOpAttr(name, "bias_term", true);
            graph->AddOpAttr(name, "axis", 1);
            graph->AddOpAttr(name, "weight_1", block({1, 1, cout, cin}, take((size_t)cout * cin)));
            graph->AddOpAttr(name, "weight_2", block({1, cout, 1, 1}, take(cout)));
            if (relu) {
                graph->AddOp(name + "_relu", "ReLU", {name + "_fc"}, {V(name)});
                graph->AddOpAttr(name + "_relu", "alpha", 0.0f);
            }
        } else if (kind == "softmax") {
            graph->AddOp(f[1], "Softmax", {V(f[2])}, {V(f[1])});
            graph->AddOpAttr(f[1], "axis", 1);
        } else if (kind == "prec") {
            precs.push_back({f[1], f[2]});
        } else if (kind == "precsplit") {       // the Split node Freeze inserts behind a variable with several readers
            precs.push_back({V(f[1]) + "split", f[2]});
        } else if (kind == "scale") {
            scales.push_back({V(f[1]), (float)atof(f[2].c_str())});
        } else if (kind == "calibrator") {
            cal_config = f[1];
            cal_table = f[2];
        } else if (kind != "precision") {
            fprintf(stderr, "unknown record %s\n", kind.c_str());
            return 2;
        }
    }
    if (g_wpos != g_weights.size()) { fprintf(stderr, "weights.bin: %zu floats left over\n", g_weights.size() - g_wpos); return 2; }

    if (!graph->Freeze()) { fprintf(stderr, "Freeze failed\n"); return 2; }
    for (auto& p : precs)
        if (!graph->SetOpPrec(p.first, p.second == "int8" ? AK_INT8 : AK_FLOAT)) {
            fprintf(stderr, "SetOpPrec: no node %s\n", p.first.c_str());
            return 2;
        }
    for (auto& s : scales) graph->SetVarScale(s.first, s.second);
    // the text-file route of a deployed model: node precisions from the net config, edge scales from the calibration table
    // (Graph::load_calibrator_config, graph.cpp:555-571 -> CalibratorParser::parse_from_file, calibrator_parse.cpp:338-460)
    if (!cal_config.empty()) graph->load_calibrator_config(cal_config, cal_table);
    graph->AddOpAttr(in_name, "input_shape", PTuple<int>(in_shape[0], in_shape[1], in_shape[2], in_shape[3]));

    graph->Optimize();      // the reference's fusion pass + stride-up + schedulers + memory planner

    // Nodes the optimiser CREATED (apply_stride_up inserts 1x1 / stride-s max poolings on a shortcut,
    // optimize_strategy.h:213-248) carry no precision and their new edges no scale: give them their producer's, as a user
    // would in the calibrator config of the optimised model.
    if (P == Precision::INT8) {
        auto fix = [&](graph::NodePtr& node_p) {
            if (node_p->bit_type() != AK_INVALID) return;
            auto& ins = graph->get_in_arc_its(node_p->name());
            if (ins.empty()) return;
            graph::NodePtr src = (*graph)[ins[0]->bottom()];
            if (src->bit_type() != AK_INT8 && src->bit_type() != AK_UINT8) return;
            node_p->set_bit_type(AK_INT8);
            std::vector<float> sc;
            for (auto& e : graph->get_in_arc_its(src->name())) if (e->scale().size()) sc = e->scale();
            if (sc.empty()) return;
            for (auto& e : graph->get_in_arc_its(node_p->name())) e->set_scale(sc);
            for (auto& e : graph->get_out_arc_its(node_p->name())) e->set_scale(sc);
            // the same Edge objects are reachable from the other end's arc list
            for (auto& e : graph->get_out_arc_its(src->name())) if (e->top() == node_p->name()) e->set_scale(sc);
            for (auto& e : graph->get_out_arc_its(node_p->name()))
                for (auto& e2 : graph->get_in_arc_its(e->top())) if (e2->bottom() == node_p->name()) e2->set_scale(sc);
        };
        graph->Scanner->BFS(fix);
    }

    Net<MI355X, P> net(true);
    net.init(*graph, /*auto_config_layout=*/P == Precision::INT8);

    // ---- the op list the reference produced ----------------------------------------------------------------------
    FILE* fo = fopen((outdir + "/oplist.txt").c_str(), "w");
    auto describe = [&](const char* tag, Tensor4dPtr<MI355X> t, const std::string& edge, bool shared, const std::string& from) {
        Shape s = t->valid_shape();
        fprintf(fo, "  %s %s %s %s [", tag, edge.c_str(), dtype_name(t->get_dtype()), layout_name(t->get_layout()));
        for (int d = 0; d < s.dims(); ++d) fprintf(fo, d ? ",%d" : "%d", s[d]);
        fprintf(fo, "] scale %.9g ptr %p%s%s\n", t->get_scale().size() ? t->get_scale()[0] : 0.f, t->data(),
                shared ? " shared_from " : "", shared ? from.c_str() : "");
    };
    for (size_t i = 0; i < net._exec_funcs.size(); ++i) {
        auto& ex = net._exec_funcs[i];
        std::string prec = net._calibrator_parser.get_precision(ex.name);
        fprintf(fo, "op %zu %s %s %s lane %d\n", i, ex.name.c_str(), ex.op_name.c_str(), prec.c_str(), (int)ex.current_lane);
        auto& in_arcs = net._graph_p->get_in_arc_its(ex.name);
        for (size_t j = 0; j < ex.ins.size(); ++j) describe("in", ex.ins[j], in_arcs[j]->name(), in_arcs[j]->shared(), in_arcs[j]->share_from());
        auto& out_arcs = net._graph_p->get_out_arc_its(ex.name);
        for (size_t j = 0; j < ex.outs.size(); ++j) describe("out", ex.outs[j], out_arcs[j]->name(), out_arcs[j]->shared(), out_arcs[j]->share_from());
    }
    fclose(fo);

    auto feed = [&]() {
        Tensor4dPtr<MI355X> din = net.get_in(in_name);
        Tensor<X86> hin(din->valid_shape(), AK_FLOAT);
        if ((size_t)hin.valid_size() != input.size()) { fprintf(stderr, "input.bin: %zu floats, the net wants %lld\n", input.size(), (long long)hin.valid_size()); exit(2); }
        memcpy(hin.mutable_data(), input.data(), input.size() * sizeof(float));
        din->copy_from(hin);
    };
    auto dump = [&](Tensor4dPtr<MI355X> t, const std::string& path) {
        Tensor<X86> h(t->valid_shape(), t->get_dtype());
        h.copy_from(*t);
        FILE* f = fopen(path.c_str(), "wb");
        fwrite(h.data(), h.get_dtype_size(), h.valid_size(), f);
        fclose(f);
    };

    // ---- the plan behind prediction() (mi355x_net_plan.h): what the capture recorded and what the executor made of it --------
    {
        MI355XNetPlan& plan = net.mi355x_plan();
        FILE* fp = fopen((outdir + "/plan.txt").c_str(), "w");
        fprintf(fp, "plan %d captured_ops %d launches %d graph %d eager_ms %.6f graph_ms %.6f why %s\n", plan.net ? 1 : 0, plan.captured_ops,
                plan.launches, plan.use_graph ? 1 : 0, plan.eager_ms, plan.graph_ms, plan.why.empty() ? "-" : plan.why.c_str());
        if (plan.net) {
            fprintf(fp, "tensors %d arena_bytes %zu\n", saber_hip_net_num_tensors(plan.net), saber_hip_net_arena_bytes(plan.net));
            for (int i = 0; i < saber_hip_net_num_ops(plan.net); ++i) fprintf(fp, "op %d %s\n", i, saber_hip_net_op_name(plan.net, i));
        }
        fclose(fp);
    }

    if (iters < 0) {      // "dry": graph + optimiser + Net::init only (runs on the mock HIP runtime without a GPU)
        printf("net dry run ok: %zu ops after Graph::Optimize + Net::init\n", net._exec_funcs.size());
        return 0;
    }

    // ---- 1. Net::prediction(): through the captured plan (the default), then with the plan switched off = the reference's
    //         operator loop untouched (all buffers aliased by its memory planner) ------------------------------------------
    const bool planned = net.mi355x_plan().net != nullptr;
    feed();
    net.prediction();
    TargetWrapper<MI355X>::device_sync();
    for (auto& o : graph->get_outs()) dump(net.get_out(o), outdir + "/out_" + o + ".bin");
    auto time_predictions = [&](int n) {
        Context<MI355X> ctx(0, 0, 0);
        for (int i = 0; i < 10; ++i) net.prediction();
        TargetWrapper<MI355X>::device_sync();
        SaberTimer<MI355X> timer;
        timer.start(ctx);
        for (int i = 0; i < n; ++i) net.prediction();
        timer.end(ctx);
        return timer.get_average_ms() / n;
    };
    const double ms_plan = (iters > 0 && planned) ? time_predictions(iters) : 0.0;
    net.mi355x_plan().enabled = false;
    net.mi355x_plan().drop();
    feed();
    net.prediction();
    TargetWrapper<MI355X>::device_sync();
    for (auto& o : graph->get_outs()) dump(net.get_out(o), outdir + "/out_" + o + "_oploop.bin");

    // ---- 2. timing of Net::prediction() ---------------------------------------------------------------------------
    if (iters > 0) {
        const double ms_loop = time_predictions(iters);
        FILE* ft = fopen((outdir + "/timing.txt").c_str(), "w");
        fprintf(ft, "ops %zu iters %d ms_per_prediction %.6f ms_per_prediction_op_loop %.6f planned %d\n", net._exec_funcs.size(), iters,
                planned ? ms_plan : ms_loop, ms_loop, planned ? 1 : 0);
        fclose(ft);
    }

    // ---- 3. op by op (the loop body of Net::prediction, net.cpp:426-456), every output edge dumped when it is fresh --
    feed();
    for (size_t i = 0; i < net._exec_funcs.size(); ++i) {
        auto& ex = net._exec_funcs[i];
        for (size_t j = 0; j < ex.ins.size(); ++j) ex.ins[j]->sync();
        if (ex.op_name != "Input" && ex.op_name != "Output") {
            ex.infer_shape();
            ex.launch();
        }
        for (size_t j = 0; j < ex.outs.size(); ++j) ex.outs[j]->record_event(ex.ctx_p->get_compute_stream());
        TargetWrapper<MI355X>::device_sync();
        if (ex.op_name == "Split" || ex.op_name == "Output") continue;
        for (size_t j = 0; j < ex.outs.size(); ++j) dump(ex.outs[j], outdir + "/step_" + std::to_string(i) + "_" + std::to_string(j) + ".bin");
    }
    printf("net ok: %zu ops executed through Net<MI355X>::prediction\n", net._exec_funcs.size());
    return 0;
}

int main(int argc, char** argv) {
    if (argc < 5) {
        fprintf(stderr, "usage: %s model.txt weights.bin input.bin outdir [timing iters]\n", argv[0]);
        return 2;
    }
    logger::init(argv[0]);
    std::ifstream fm(argv[1]);
    std::vector<Record> recs;
    std::string line, precision = "int8";
    while (std::getline(fm, line)) {
        std::istringstream is(line);
        Record r;
        std::string tok;
        while (is >> tok) r.f.push_back(tok);
        if (r.f.empty() || r.f[0][0] == '#') continue;
        if (r.f[0] == "precision") precision = r.f[1];
        recs.push_back(r);
    }
    auto slurp = [](const char* path) {
        std::ifstream f(path, std::ios::binary | std::ios::ate);
        if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
        std::vector<float> v((size_t)f.tellg() / sizeof(float));
        f.seekg(0);
        f.read((char*)v.data(), v.size() * sizeof(float));
        return v;
    };
    g_weights = slurp(argv[2]);
    std::vector<float> input = slurp(argv[3]);
    const int iters = argc > 5 ? (std::string(argv[5]) == "dry" ? -1 : atoi(argv[5])) : 0;
    Env<MI355X>::env_init();
    if (precision == "int8") return run<Precision::INT8>(recs, input, argv[4], iters);
    return run<Precision::FP32>(recs, input, argv[4], iters);
}
