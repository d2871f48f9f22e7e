// integration/mi355x/framework/mi355x_created_nodes.h - precision and scale for the nodes Graph::Optimize() CREATES.
//
// graph_strategy::apply_stride_up (framework/graph/optimize_strategy.h:213-248, run for every target at graph.cpp:407) moves
// the stride of a stage-entry 1x1 conv up into the layer before it and inserts a 1x1 / stride-s max pooling on the shortcut
// that still needs the full-resolution tensor. Those pooling nodes are born after the user's precisions and scales were
// loaded (Graph::load_calibrator_config / SetOpPrec / SetVarScale act on the model's own nodes): bit_type AK_INVALID, edges
// without a scale. Net::init would make them FP32 operators between 8-bit tensors. A sub-sampling max pooling of an 8-bit
// tensor is the same 8-bit tensor with fewer pixels, so the MI355X target gives such a node its producer's precision and its
// edges the producer's input scale - at the top of Net<MI355X>::init, where both the direct route (Net::init on a user's graph)
// and Worker (worker.cpp:28, which offers no hook between Optimize() and init) pass. Other targets: untouched.
#pragma once
#include <type_traits>
#include <vector>
#include "framework/graph/graph.h"

namespace anakin {

template <typename Ttype, Precision Ptype>
inline void mi355x_created_nodes_inherit(graph::Graph<Ttype, Ptype>& graph) {
    if (!std::is_same<Ttype, saber::MI355X>::value) {
        return;
    }
    auto inherit = [&](graph::NodePtr& node_p) {
        if (node_p->bit_type() != saber::AK_INVALID) {
            return;
        }
        auto& ins = graph.get_in_arc_its(node_p->name());
        if (ins.empty()) {
            return;
        }
        graph::NodePtr src = graph[ins[0]->bottom()];
        if (src->bit_type() != saber::AK_INT8 && src->bit_type() != saber::AK_UINT8) {
            return;
        }
        node_p->set_bit_type(saber::AK_INT8);
        std::vector<float> sc;
        for (auto& e : graph.get_in_arc_its(src->name())) {
            if (e->scale().size()) {
                sc = e->scale();
            }
        }
        if (sc.empty()) {
            return;
        }
        for (auto& e : graph.get_in_arc_its(node_p->name())) {
            e->set_scale(sc);
        }
        for (auto& e : graph.get_out_arc_its(node_p->name())) {
            e->set_scale(sc);
        }
        // (the arc lists of the nodes at the other end hold their own handles to the same edges)
        for (auto& e : graph.get_out_arc_its(src->name())) {
            if (e->top() == node_p->name()) {
                e->set_scale(sc);
            }
        }
        for (auto& e : graph.get_out_arc_its(node_p->name())) {
            for (auto& e2 : graph.get_in_arc_its(e->top())) {
                if (e2->bottom() == node_p->name()) {
                    e2->set_scale(sc);
                }
            }
        }
    };
    graph.Scanner->BFS(inherit);
}

}  // namespace anakin
