// framework/model_parser/parser/parser.cpp REPLACEMENT for builds without protobuf (this container: no protoc, no libprotobuf;
// the reference's parser.cpp is the only protobuf consumer of framework/ besides the nanopb copy): a loader for a TEXT model
// format carrying what an `.anakin.bin` carries for the ResNet / VGG family - the network as ORIGINAL operators (Convolution,
// BatchNorm, Scale, ReLU, Pooling, Eltwise, Dense, Softmax) with their attributes and raw weight blobs - so that
// Graph<T,P>::load(path) (framework/graph/graph.cpp:16-40), Worker<T,P,R>(model_path, threads) (framework/core/net/worker.cpp:17-42)
// and every caller of the model-parser interface work on this target's build the way they do with the protobuf parser:
//     graph.load("model.txt"); graph.Optimize(); net.init(graph);
// Format (one record per line, `#` comments; written by integration/net_model.py):
//     precision int8|fp32
//     weights <file>                    f32 blobs in record order (relative to the model file's directory)
//     input  <name> n c h w
//     conv   <name> <src> cin cout k stride pad relu(0|1) bn(0|1)     blobs: w[cout,cin,k,k]; bn=0: bias[cout];
//                                                                      bn=1: mean[cout] var[cout] gamma[cout] beta[cout]
//     pool   <name> <src> MAX|AVG win stride pad global(0|1)
//     eltwise <name> <a> <b> relu(0|1) coeff_a coeff_b
//     fc     <name> <src> cin cout relu(0|1)                            blobs: w[cout,cin] bias[cout]
//     softmax <name> <src>
//     prec   <node> int8|fp32          (Graph::SetOpPrec after Freeze)
//     precsplit <node> int8|fp32       (the Split node Graph::Freeze inserts behind <node>'s output when it has several readers)
//     scale  <node> <float>            (Graph::SetVarScale on <node>'s output variable)
//     calibrator <net_config.txt> <calibrator.txt>     (Graph::load_calibrator_config, graph.cpp:555-571, instead of prec / scale)
// The graph is built with the reference's own public construction API (Graph::AddOp / AddOpAttr / Freeze, graph.h:97-139 - the
// route of test/framework/net/net_subgraph_test.cpp); weight blocks come from GraphGlobalMem::new_block as in the protobuf parser
// (parser.cpp / model_io.cpp). Every other file is taken for an `.anakin.bin`: load / save / load(buffer) / InspectAnakin at the end of this
// file hand it to anakin_bin_parser.cpp (the protobuf wire format without a protobuf library).
// Reference-side glue of the MI355X target's TEST BUILD (integration/), not part of the product library.
#include "framework/model_parser/parser/parser.h"

#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <vector>

#include "framework/graph/graph.h"
#include "framework/graph/graph_global_mem.h"
#include "framework/model_parser/parser/anakin_bin_model.h"

namespace anakin {
namespace parser {

using namespace anakin::saber;

static Status unsupported() { return Status::ANAKINFAIL("text model parser: a text model is loaded from its path only"); }

template <typename Ttype, Precision Ptype>
Status load_text_model(graph::Graph<Ttype, Ptype>* graph, const std::string& path) {
    std::ifstream fm(path);
    if (!fm.is_open()) return Status::ANAKINFAIL(("text model: cannot open " + path).c_str());
    const size_t slash = path.find_last_of('/');
    const std::string dir = slash == std::string::npos ? std::string() : path.substr(0, slash + 1);
    struct Record { std::vector<std::string> f; };
    std::vector<Record> recs;
    std::string line, wpath = dir + "weights.bin";
    while (std::getline(fm, line)) {
        std::istringstream is(line);
        Record r;
        std::string tok;
        while (is >> tok) r.f.push_back(tok);
        if (r.f.empty() || r.f[0][0] == '#') continue;
        if (r.f[0] == "weights") wpath = r.f[1][0] == '/' ? r.f[1] : dir + r.f[1];
        recs.push_back(r);
    }
    std::vector<float> weights;
    {
        std::ifstream f(wpath, std::ios::binary | std::ios::ate);
        if (!f) return Status::ANAKINFAIL(("text model: cannot open weights " + wpath).c_str());
        weights.resize((size_t)f.tellg() / sizeof(float));
        f.seekg(0);
        f.read((char*)weights.data(), weights.size() * sizeof(float));
    }
    size_t wpos = 0;
    bool short_weights = false;
    auto take = [&](size_t n) -> const float* {
        static const float zero = 0.f;
        if (wpos + n > weights.size()) { short_weights = true; return &zero; }
        const float* p = weights.data() + wpos;
        wpos += n;
        return p;
    };
    // a weight block as the protobuf parser creates it: owned by the graph's global memory, host + device copy
    auto blk = [&](const std::vector<int>& shape4, const float* src) -> PBlock<Ttype>* {
        Shape sh(shape4);
        PBlock<Ttype>* b = graph::GraphGlobalMem<Ttype>::Global().template new_block<AK_FLOAT>(sh);
        if (!short_weights) memcpy(b->h_tensor().mutable_data(), src, sizeof(float) * sh.count());
        b->d_tensor().set_shape(sh);
        b->d_tensor().copy_from(b->h_tensor());
        return b;
    };
    std::string in_name;
    std::vector<int> in_shape;
    std::vector<std::pair<std::string, std::string> > precs;
    std::vector<std::pair<std::string, float> > scales;
    std::string cal_config, cal_table;
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
            graph->AddOpAttr(name, "weight_1", *blk({cout, cin, k, k}, take((size_t)cout * cin * k * k)));
            std::string cur = name + "_conv";
            if (!bn) {
                graph->AddOpAttr(name, "weight_2", *blk({1, cout, 1, 1}, take(cout)));
            } else {
                // Caffe: BatchNorm (mean, variance, moving-average factor) then Scale (gamma, beta)
                const std::string bnn = "bn_" + name, scn = "scale_" + name;
                graph->AddOp(bnn, "BatchNorm", {cur}, {name + "_bn"});
                graph->AddOpAttr(bnn, "epsilon", 1e-5f);
                graph->AddOpAttr(bnn, "momentum", 0.999f);
                graph->AddOpAttr(bnn, "weight_1", *blk({1, cout, 1, 1}, take(cout)));
                graph->AddOpAttr(bnn, "weight_2", *blk({1, cout, 1, 1}, take(cout)));
                const float one = 1.f;
                graph->AddOpAttr(bnn, "weight_3", *blk({1, 1, 1, 1}, &one));
                const std::string sc_out = relu ? name + "_scale" : V(name);
                graph->AddOp(scn, "Scale", {name + "_bn"}, {sc_out});
                graph->AddOpAttr(scn, "num_axes", 1);
                graph->AddOpAttr(scn, "bias_term", true);
                graph->AddOpAttr(scn, "axis", 1);
                graph->AddOpAttr(scn, "weight_1", *blk({1, cout, 1, 1}, take(cout)));
                graph->AddOpAttr(scn, "weight_2", *blk({1, cout, 1, 1}, take(cout)));
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
            graph->AddOpAttr(name, "weight_1", *blk({1, 1, cout, cin}, take((size_t)cout * cin)));
            graph->AddOpAttr(name, "weight_2", *blk({1, cout, 1, 1}, take(cout)));
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
        } else if (kind == "weights") {
            wpath = f[1][0] == '/' ? f[1] : dir + f[1];
        } else if (kind != "precision") {
            return Status::ANAKINFAIL(("text model: unknown record " + kind).c_str());
        }
    }
    if (short_weights) return Status::ANAKINFAIL("text model: the weights file is too short");
    if (wpos != weights.size()) return Status::ANAKINFAIL("text model: floats left over in the weights file");
    if (in_name.empty() || in_shape.size() != 4) return Status::ANAKINFAIL("text model: no input record");
    if (!graph->Freeze()) return Status::ANAKINFAIL("text model: Graph::Freeze failed");
    for (auto& p : precs)
        if (!graph->SetOpPrec(p.first, p.second == "int8" ? AK_INT8 : AK_FLOAT))
            return Status::ANAKINFAIL(("text model: SetOpPrec on an unknown node " + p.first).c_str());
    for (auto& s : scales) graph->SetVarScale(s.first, s.second);
    // the text-file route of a deployed model: node precisions from the net config, edge scales from the calibration table
    // (Graph::load_calibrator_config, graph.cpp:555-571 -> CalibratorParser::parse_from_file, calibrator_parse.cpp:338-460)
    if (!cal_config.empty()) graph->load_calibrator_config(cal_config[0] == '/' ? cal_config : dir + cal_config,
                                                           cal_table[0] == '/' ? cal_table : dir + cal_table);
    graph->AddOpAttr(in_name, "input_shape", PTuple<int>(in_shape[0], in_shape[1], in_shape[2], in_shape[3]));
    return Status::OK();
}

// ---- the interface of parser.h: a file that starts with the text form's `precision` record goes to the loader above, anything else is an
// `.anakin.bin` (protobuf wire format; anakin_bin_parser.cpp, in model_io.cpp's place) ---------------------------------------------------
template <typename Ttype, Precision Ptype>
Status load_anakin_bin(graph::Graph<Ttype, Ptype>* graph, const char* data, size_t len);
template <typename Ttype, Precision Ptype>
Status save_anakin_bin(graph::Graph<Ttype, Ptype>* graph, const char* path);

static bool is_text_model(const char* data, size_t len) {
    size_t i = 0;
    while (i < len && (data[i] == ' ' || data[i] == '\n' || data[i] == '\t' || data[i] == '\r')) ++i;
    if (i < len && data[i] == '#') return true;
    return len - i >= 9 && std::memcmp(data + i, "precision", 9) == 0;
}
static bool read_file(const std::string& path, std::string& out) {
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f) return false;
    out.resize((size_t)f.tellg());
    f.seekg(0);
    f.read(&out[0], (std::streamsize)out.size());
    return f.good() || out.empty();
}

template <typename Ttype, Precision Ptype>
Status load(graph::Graph<Ttype, Ptype>* graph, const char* model_path) {
    std::string head;
    {
        std::ifstream f(model_path, std::ios::binary);
        if (!f) return Status::ANAKINFAIL((std::string("model parser: cannot open ") + model_path).c_str());
        head.resize(64);
        f.read(&head[0], 64);
        head.resize((size_t)f.gcount());
    }
    if (is_text_model(head.data(), head.size())) return load_text_model(graph, std::string(model_path));
    std::string bytes;
    if (!read_file(model_path, bytes)) return Status::ANAKINFAIL((std::string("model parser: cannot read ") + model_path).c_str());
    return load_anakin_bin(graph, bytes.data(), bytes.size());
}
template <typename Ttype, Precision Ptype>
Status load(graph::Graph<Ttype, Ptype>* graph, std::string& model_path) { return load(graph, model_path.c_str()); }
template <typename Ttype, Precision Ptype>
Status load(graph::Graph<Ttype, Ptype>* graph, const char* buffer, size_t len) {      // an `.anakin.bin` in memory (parser.cpp:244-249)
    if (is_text_model(buffer, len)) return unsupported();
    return load_anakin_bin(graph, buffer, len);
}
template <typename Ttype, Precision Ptype>
Status save(graph::Graph<Ttype, Ptype>* graph, const char* model_path) { return save_anakin_bin(graph, model_path); }
template <typename Ttype, Precision Ptype>
Status save(graph::Graph<Ttype, Ptype>* graph, std::string& model_path) { return save_anakin_bin(graph, model_path.c_str()); }

bool InspectAnakin(const std::string& path) {
    std::string bytes;
    if (!read_file(path, bytes)) return false;
    return InspectAnakin(bytes.data(), bytes.size());
}
bool InspectAnakin(const char* buffer, size_t len) {
    if (is_text_model(buffer, len)) return true;
    ::anakin_bin::Graph g;
    return ::anakin_bin::decode((const uint8_t*)buffer, len, g) && !g.nodes.empty();
}

#define MI355X_PARSER_INSTANCE(T, P)                                              \
    template Status load<T, P>(graph::Graph<T, P>*, std::string&);                \
    template Status load<T, P>(graph::Graph<T, P>*, const char*);                 \
    template Status load<T, P>(graph::Graph<T, P>*, const char*, size_t);         \
    template Status save<T, P>(graph::Graph<T, P>*, std::string&);                \
    template Status save<T, P>(graph::Graph<T, P>*, const char*);

MI355X_PARSER_INSTANCE(X86, Precision::FP32)
MI355X_PARSER_INSTANCE(X86, Precision::FP16)
MI355X_PARSER_INSTANCE(X86, Precision::INT8)
#ifdef USE_MI355X_PLACE
MI355X_PARSER_INSTANCE(MI355X, Precision::FP32)
MI355X_PARSER_INSTANCE(MI355X, Precision::FP16)
MI355X_PARSER_INSTANCE(MI355X, Precision::INT8)
#endif

}  // namespace parser
}  // namespace anakin
