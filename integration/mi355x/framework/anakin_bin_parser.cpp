// framework/model_parser/parser/model_io.cpp REPLACEMENT for builds without protobuf: Graph<T,P> <-> `.anakin.bin` through the wire-format
// reader / writer of anakin_bin_model.h (no protoc-generated classes, no libprotobuf / nanopb runtime - neither exists in this image).
// Behaviour restated from the reference's protobuf route, which stays the parser of a maintainer's build that HAS protobuf:
//   load:  parser.cpp:131-240 generate_graph_with_graph_proto (name, ins / outs, nodes, edges_in / edges_out with the per-target scale and
//          layout, edges_info's shared / share_from, the optimisation summary) and model_io.cpp:9-300 NodeIO::operator>>(NodeProto) (node
//          name / lane / need_wait / bit_type, one attribute per valueType by its DateTypeProto tag, weight tensors into
//          GraphGlobalMem blocks - float and int8 payloads, real and valid shape, per-channel scale, weights shared from another node);
//   save:  parser.cpp:262-375 and model_io.cpp:330-470 (nodes in execution order, the ones sharing weights last; every arc as a target of
//          edges_in[top] and edges_out[bottom] plus its edges_info entry).
// Differences, on purpose: a malformed file or an unsupported attribute type is a failure Status with the reason, not LOG(FATAL); a weight
// payload shorter than its shape is refused (the reference reads past the repeated field's end).
// `load` / `save` themselves (the interface of parser.h) are in text_model_parser.cpp, which hands every file that is not the text form here.
// Reference-side glue of the MI355X target's TEST BUILD (integration/), not part of the product library.
#include <cstdio>
#include <fstream>
#include <unordered_map>

#include "framework/graph/graph.h"
#include "framework/graph/graph_global_mem.h"
#include "framework/model_parser/parser/anakin_bin_model.h"
#include "framework/model_parser/parser/parser.h"

namespace anakin {
namespace parser {

using namespace anakin::saber;
namespace ab = ::anakin_bin;

namespace {
Status bad(const std::string& why) { return Status::ANAKINFAIL(("anakin.bin: " + why).c_str()); }

template <typename Ttype, typename V>
bool weight_block(const ab::Tensor& t, DataType dt, const V* src, size_t have, graph::NodePtr& node_p, const std::string& key, std::string* why) {
    if (t.shape.value.size() != 4) { *why = "weight " + key + " of " + node_p->name() + ": the shape must have 4 dims"; return false; }
    Shape real({1, 1, 1, 1});
    for (int i = 0; i < 4; ++i) real[i] = t.shape.value[i];
    const size_t count = (size_t)real.count();
    if ((int64_t)count != t.data.size || have < count) {
        *why = "weight " + key + " of " + node_p->name() + ": " + std::to_string(have) + " values for a shape of " + std::to_string(count);
        return false;
    }
    std::vector<float> scale(t.scale.f.begin(), t.scale.f.end());
    PBlock<Ttype>* block = dt == AK_FLOAT ? graph::GraphGlobalMem<Ttype>::Global().template new_block<AK_FLOAT>(real)
                                          : graph::GraphGlobalMem<Ttype>::Global().template new_block<AK_INT8>(real);
    std::memcpy(block->h_tensor().mutable_data(), src, count * sizeof(V));
    block->d_tensor().set_scale(scale);
    block->h_tensor().set_scale(scale);
    block->d_tensor().set_shape(real);
    block->d_tensor().copy_from(block->h_tensor());
    if (!t.valid_shape.value.empty()) {
        if (t.valid_shape.value.size() != 4) { *why = "weight " + key + " of " + node_p->name() + ": the valid shape must have 4 dims"; return false; }
        Shape valid({1, 1, 1, 1});
        for (int i = 0; i < 4; ++i) valid[i] = t.valid_shape.value[i];
        block->d_tensor().set_shape(valid);
        block->h_tensor().set_shape(valid);
    }
    node_p->set_attr(key, *block);
    return true;
}
}  // namespace

template <typename Ttype, Precision Ptype>
Status load_anakin_bin(graph::Graph<Ttype, Ptype>* graph, const char* data, size_t len) {
    ab::Graph g;
    if (!ab::decode((const uint8_t*)data, len, g)) return bad("not a well-formed GraphProto");
    if (g.nodes.empty()) return bad("no nodes");
    graph->set_name(g.name);
    for (auto& s : g.ins) graph->add_in(s);
    for (auto& s : g.outs) graph->add_out(s);

    std::unordered_map<std::string, graph::NodePtr> by_name;
    for (const ab::Node& n : g.nodes) {
        graph::NodePtr node_p = std::make_shared<graph::Node>();
        if (!by_name.count(n.name)) by_name[n.name] = node_p;
        node_p->name() = n.name;
        node_p->need_wait() = n.need_wait;
        node_p->lane() = n.lane;
        node_p->bit_type() = n.bit_type == ab::DT_INT8 ? AK_INT8 : (n.bit_type == ab::DT_FLOAT ? AK_FLOAT : AK_INVALID);
        for (const auto& kv : n.attr) {
            const std::string& key = kv.first;
            const ab::Value& v = kv.second;
            switch (v.type) {
            case ab::DT_STR: node_p->set_attr(key, v.s); break;
            case ab::DT_INT32: node_p->set_attr(key, (int)v.i); break;
            case ab::DT_FLOAT: case ab::DT_DOUBLE: node_p->set_attr(key, v.f); break;
            case ab::DT_BOOLEN: node_p->set_attr(key, v.b); break;
            case ab::DT_CACHE_LIST: {
                const ab::Cache& c = v.cache_list;
                const size_t n_el = (size_t)c.size;
                if (c.type == ab::DT_FLOAT) {
                    if (c.f.size() < n_el) return bad("attribute " + key + " of " + n.name + ": short float list");
                    PTuple<float> l;
                    for (size_t i = 0; i < n_el; ++i) l.push_back(c.f[i]);
                    node_p->set_attr(key, l);
                } else if (c.type == ab::DT_BOOLEN) {
                    if (c.b.size() < n_el) return bad("attribute " + key + " of " + n.name + ": short bool list");
                    PTuple<bool> l;
                    for (size_t i = 0; i < n_el; ++i) l.push_back(c.b[i] != 0);
                    node_p->set_attr(key, l);
                } else if (c.type == ab::DT_INT32) {
                    if (c.i.size() < n_el) return bad("attribute " + key + " of " + n.name + ": short int list");
                    PTuple<int> l;
                    for (size_t i = 0; i < n_el; ++i) l.push_back(c.i[i]);
                    node_p->set_attr(key, l);
                } else if (c.type == ab::DT_STR) {
                    if (c.s.size() < n_el) return bad("attribute " + key + " of " + n.name + ": short string list");
                    PTuple<std::string> l;
                    for (size_t i = 0; i < n_el; ++i) l.push_back(c.s[i]);
                    node_p->set_attr(key, l);
                } else if (c.type == ab::DT_CACHE_LIST) {      // lists of int lists only, as in the reference
                    PTuple<PTuple<int> > ll;
                    for (const ab::Cache& in : c.l) {
                        if (in.type != ab::DT_INT32 || in.i.size() < (size_t)in.size) return bad("attribute " + key + " of " + n.name + ": a list of lists must hold int lists");
                        ll.push_back(PTuple<int>());
                        for (size_t i = 0; i < (size_t)in.size; ++i) ll[ll.size() - 1].push_back(in.i[i]);
                    }
                    node_p->set_attr(key, ll);
                } else {
                    return bad("attribute " + key + " of " + n.name + ": list element type " + std::to_string(c.type));
                }
            } break;
            case ab::DT_TENSOR: {
                const ab::Tensor& t = v.tensor;
                if (t.shared) {      // weights shared from a node read earlier
                    auto it = by_name.find(t.share_from);
                    if (it == by_name.end() || !it->second->inspect_attr(key)) return bad("weight " + key + " of " + n.name + " is shared from an unknown node " + t.share_from);
                    node_p->set_attr(key, it->second->template get_attr<PBlock<Ttype> >(key));
                    node_p->set_share_pair(key, t.share_from);
                    break;
                }
                std::string why;
                bool ok;
                if (t.data.type == ab::DT_FLOAT) ok = weight_block<Ttype, float>(t, AK_FLOAT, t.data.f.data(), t.data.f.size(), node_p, key, &why);
                else if (t.data.type == ab::DT_INT8) ok = weight_block<Ttype, char>(t, AK_INT8, t.data.c.data(), t.data.c.size(), node_p, key, &why);
                else { ok = false; why = "weight " + key + " of " + n.name + ": payload type " + std::to_string(t.data.type); }
                if (!ok) return bad(why);
            } break;
            default:
                return bad("attribute " + key + " of " + n.name + ": value type " + std::to_string(v.type));
            }
        }
        node_p->get_op_name() = n.op.name;
        graph->add_vertex(node_p->name(), node_p);
    }

    // an edge whose ends are not nodes of the file would trip a CHECK deep inside GraphBase (an abort): refused here with its name
    for (int dir = 0; dir < 2; ++dir)
        for (const auto& kv : (dir == 0 ? g.edges_in : g.edges_out)) {
            if (!by_name.count(kv.first)) return bad(std::string(dir == 0 ? "edges_in" : "edges_out") + " of an unknown node " + kv.first);
            for (const ab::Target& tg : kv.second.target)
                if (!by_name.count(tg.node)) return bad("edge " + (dir == 0 ? tg.node + "_" + kv.first : kv.first + "_" + tg.node) + ": unknown node " + tg.node);
            for (const std::string& other : kv.second.val)
                if (!by_name.count(other)) return bad("edge " + (dir == 0 ? other + "_" + kv.first : kv.first + "_" + other) + ": unknown node " + other);
        }
    auto finish = [&](graph::Edge<Ttype>& e) {
        auto it = g.edges_info.find(e.name());
        if (it != g.edges_info.end()) {
            e.shared() = it->second.shared;
            e.share_from() = it->second.share_from;
        }
    };
    for (int dir = 0; dir < 2; ++dir) {
        for (const auto& kv : (dir == 0 ? g.edges_in : g.edges_out)) {
            const std::string& key = kv.first;
            const ab::List& l = kv.second;
            if (!l.target.empty()) {
                for (const ab::Target& tg : l.target) {
                    graph::Edge<Ttype> e(dir == 0 ? tg.node : key, dir == 0 ? key : tg.node);
                    e.set_scale(tg.scale);
                    e.set_layout((LayoutType)(tg.layout == 0 ? (int)Layout_NCHW : tg.layout));
                    finish(e);
                    if (dir == 0) graph->add_in_arc(e); else graph->add_out_arc(e);
                }
            } else {
                for (const std::string& other : l.val) {
                    graph::Edge<Ttype> e(dir == 0 ? other : key, dir == 0 ? key : other);
                    finish(e);
                    if (dir == 0) graph->add_in_arc(e); else graph->add_out_arc(e);
                }
            }
        }
    }
    graph->statistics.template set_info<graph::IS_OPTIMIZED>(g.is_optimized);
    graph->statistics.template set_info<graph::TEMP_MEM>(g.temp_mem_used);
    graph->statistics.template set_info<graph::ORI_TEMP_MEM>(g.original_temp_mem_used);
    graph->statistics.template set_info<graph::SYSTEM_MEM>(g.system_mem_used);
    graph->statistics.template set_info<graph::MODEL_MEM>(g.model_mem_used);
    return Status::OK();
}

template <typename Ttype, Precision Ptype>
Status save_anakin_bin(graph::Graph<Ttype, Ptype>* graph, const char* path) {
    ab::Graph g;
    g.name = graph->name();
    for (auto& s : graph->get_ins()) g.ins.push_back(s);
    for (auto& s : graph->get_outs()) g.outs.push_back(s);

    auto put_node = [&](graph::NodePtr& node_p) -> std::string {
        g.nodes.emplace_back();
        ab::Node& n = g.nodes.back();
        n.name = node_p->name();
        n.lane = node_p->lane();
        n.need_wait = node_p->need_wait();
        n.bit_type = node_p->bit_type() == AK_INT8 ? ab::DT_INT8 : ab::DT_FLOAT;
        n.op.name = node_p->get_op_name();
        n.op.present = true;
        for (auto it = node_p->attr().begin(); it != node_p->attr().end(); ++it) {
            const std::string& key = it->first;
            auto& value = it->second;
            const std::string ty = value.type();
            n.attr.emplace_back();
            n.attr.back().first = key;
            ab::Value& v = n.attr.back().second;
            if (ty == "anakin_string") { v.type = ab::DT_STR; v.s = any_cast<std::string>(value); }
            else if (ty == "anakin_int32") { v.type = ab::DT_INT32; v.i = any_cast<int>(value); }
            else if (ty == "anakin_float") { v.type = ab::DT_FLOAT; v.f = any_cast<float>(value); }
            else if (ty == "anakin_bool") { v.type = ab::DT_BOOLEN; v.b = any_cast<bool>(value); }
            else if (ty == "anakin_tuple_string") {
                auto t = any_cast<PTuple<std::string> >(value);
                v.type = ab::DT_CACHE_LIST; v.cache_list.type = ab::DT_STR; v.cache_list.size = t.size();
                for (int i = 0; i < t.size(); ++i) v.cache_list.s.push_back(t[i]);
            } else if (ty == "anakin_tuple_int") {
                auto t = any_cast<PTuple<int> >(value);
                v.type = ab::DT_CACHE_LIST; v.cache_list.type = ab::DT_INT32; v.cache_list.size = t.size();
                for (int i = 0; i < t.size(); ++i) v.cache_list.i.push_back(t[i]);
            } else if (ty == "anakin_tuple_float") {
                auto t = any_cast<PTuple<float> >(value);
                v.type = ab::DT_CACHE_LIST; v.cache_list.type = ab::DT_FLOAT; v.cache_list.size = t.size();
                for (int i = 0; i < t.size(); ++i) v.cache_list.f.push_back(t[i]);
            } else if (ty == "anakin_tuple_bool") {
                auto t = any_cast<PTuple<bool> >(value);
                v.type = ab::DT_CACHE_LIST; v.cache_list.type = ab::DT_BOOLEN; v.cache_list.size = t.size();
                for (int i = 0; i < t.size(); ++i) v.cache_list.b.push_back(t[i] == "true" ? 1 : 0);      // (PTuple<bool> keeps "true" / "false" strings: parameter.h:145-152)
            } else if (ty == "anakin_tuple_tuple_int") {
                auto t = any_cast<PTuple<PTuple<int> > >(value);
                v.type = ab::DT_CACHE_LIST; v.cache_list.type = ab::DT_CACHE_LIST; v.cache_list.size = t.size();
                for (int i = 0; i < t.size(); ++i) {
                    v.cache_list.l.emplace_back();
                    ab::Cache& in = v.cache_list.l.back();
                    in.type = ab::DT_INT32; in.size = t[i].size();
                    for (int j = 0; j < t[i].size(); ++j) in.i.push_back(t[i][j]);
                }
            } else if (ty == "anakin_block") {
                v.type = ab::DT_TENSOR;
                if (node_p->check_shared(key)) {
                    v.tensor.shared = true;
                    v.tensor.share_from = node_p->get_share_target(key);
                } else {
                    auto block = any_cast<PBlock<Ttype> >(value);
                    if (block.h_tensor().get_dtype() != AK_FLOAT) return "weight " + key + " of " + n.name + ": only float blocks are written (as in the reference)";
                    const float* cpu = static_cast<const float*>(block.h_tensor().data());
                    auto valid = block.shape();
                    auto real = block.real_shape();
                    v.tensor.shape.present = true;
                    for (int i = 0; i < real.dims(); ++i) v.tensor.shape.value.push_back(real[i]);
                    v.tensor.shape.size = real.size();
                    if (!(valid == real)) {
                        v.tensor.valid_shape.present = true;
                        for (int i = 0; i < valid.dims(); ++i) v.tensor.valid_shape.value.push_back(valid[i]);
                        v.tensor.valid_shape.size = valid.size();
                    }
                    v.tensor.data.f.assign(cpu, cpu + real.count());
                    v.tensor.data.type = ab::DT_FLOAT;
                    v.tensor.data.size = real.count();
                    auto sc = block.h_tensor().get_scale();
                    v.tensor.scale.f.assign(sc.begin(), sc.end());
                    if (!sc.empty()) { v.tensor.scale.type = ab::DT_FLOAT; v.tensor.scale.size = (int64_t)sc.size(); }
                }
            } else {
                // (a string list arrives here: PTuple<std::string> has no ANAKIN_TO_TYPE_ID in framework/core/data_types.h, its any reports an
                // empty type name - the reference's own save falls into its last branch for it as well, model_io.cpp:447-455)
                return "attribute " + key + " of " + n.name + ": type '" + ty + "' has no model-file form";
            }
        }
        return std::string();
    };
    // execution order when the graph has one (after Optimize), otherwise every vertex; nodes that share weights after their sources
    std::vector<std::string> order = graph->get_nodes_in_order();
    if (order.empty()) {
        auto collect = [&](graph::NodePtr& node_p) { order.push_back(node_p->name()); };
        graph->Scanner->BFS(collect);
    }
    for (int shared = 0; shared < 2; ++shared)
        for (auto& name : order) {
            graph::NodePtr node_p = (*graph)[name];
            if ((int)node_p->is_weight_shared() != shared) continue;
            const std::string err = put_node(node_p);
            if (!err.empty()) return bad(err);
        }

    auto find_list = [](std::vector<std::pair<std::string, ab::List> >& v, const std::string& k) -> ab::List& {
        for (auto& kv : v)
            if (kv.first == k) return kv.second;
        v.emplace_back();
        v.back().first = k;
        return v.back().second;
    };
    auto put_edges = [&](graph::NodePtr& node_p) {
        for (int dir = 0; dir < 2; ++dir) {
            auto& arcs = dir == 0 ? graph->get_in_arc_its(node_p->name()) : graph->get_out_arc_its(node_p->name());
            for (auto& e : arcs) {
                ab::List& l = find_list(dir == 0 ? g.edges_in : g.edges_out, dir == 0 ? e->second() : e->first());
                l.target.emplace_back();
                ab::Target& tg = l.target.back();
                tg.node = dir == 0 ? e->first() : e->second();
                tg.scale = e->scale();
                tg.layout = (int)e->layout();
                ab::Tensor& ts = g.edges_info[e->name()];
                ts.name = e->name();
                ts.shared = e->shared();
                ts.share_from = e->share_from();
            }
        }
    };
    graph->Scanner->BFS(put_edges);

    g.has_summary = true;
    g.is_optimized = graph->statistics.template get_info<graph::IS_OPTIMIZED>();
    g.temp_mem_used = graph->statistics.template get_info<graph::TEMP_MEM>();
    g.original_temp_mem_used = graph->statistics.template get_info<graph::ORI_TEMP_MEM>();
    g.system_mem_used = graph->statistics.template get_info<graph::SYSTEM_MEM>();
    g.model_mem_used = graph->statistics.template get_info<graph::MODEL_MEM>();

    const std::string bytes = ab::encode(g);
    std::ofstream out(path, std::ios::out | std::ios::trunc | std::ios::binary);
    if (!out) return bad(std::string("cannot write ") + path);
    out.write(bytes.data(), (std::streamsize)bytes.size());
    return out.good() ? Status::OK() : bad(std::string("short write to ") + path);
}

#define MI355X_ANAKIN_BIN_INSTANCE(T, P)                                                        \
    template Status load_anakin_bin<T, P>(graph::Graph<T, P>*, const char*, size_t);            \
    template Status save_anakin_bin<T, P>(graph::Graph<T, P>*, const char*);

MI355X_ANAKIN_BIN_INSTANCE(X86, Precision::FP32)
MI355X_ANAKIN_BIN_INSTANCE(X86, Precision::FP16)
MI355X_ANAKIN_BIN_INSTANCE(X86, Precision::INT8)
#ifdef USE_MI355X_PLACE
MI355X_ANAKIN_BIN_INSTANCE(MI355X, Precision::FP32)
MI355X_ANAKIN_BIN_INSTANCE(MI355X, Precision::FP16)
MI355X_ANAKIN_BIN_INSTANCE(MI355X, Precision::INT8)
#endif

}  // namespace parser
}  // namespace anakin
