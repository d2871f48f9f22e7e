// framework/model_parser/parser/parser.cpp REPLACEMENT for builds without protobuf (this container: no protoc, no
// libprotobuf; the reference's parser.cpp is the only protobuf consumer of framework/ besides the nanopb copy).
// Graph<>::load / save keep their symbols and fail with a Status, exactly what a missing model file does; graphs are
// built programmatically instead (Graph::AddOp / AddOpAttr / Freeze, framework/graph/graph.h:97-139 — the route of
// test/framework/net/net_subgraph_test.cpp). TEST-BUILD INFRASTRUCTURE of integration/, not part of the product library.
#include "framework/model_parser/parser/parser.h"

namespace anakin {
namespace parser {

static Status no_protobuf() { return Status::ANAKINFAIL("model parser not built: protobuf is not available in this build"); }

template <typename Ttype, Precision Ptype>
Status load(graph::Graph<Ttype, Ptype>*, std::string&) { return no_protobuf(); }
template <typename Ttype, Precision Ptype>
Status load(graph::Graph<Ttype, Ptype>*, const char*) { return no_protobuf(); }
template <typename Ttype, Precision Ptype>
Status load(graph::Graph<Ttype, Ptype>*, const char*, size_t) { return no_protobuf(); }
template <typename Ttype, Precision Ptype>
Status save(graph::Graph<Ttype, Ptype>*, std::string&) { return no_protobuf(); }
template <typename Ttype, Precision Ptype>
Status save(graph::Graph<Ttype, Ptype>*, const char*) { return no_protobuf(); }

bool InspectAnakin(const std::string&) { return false; }
bool InspectAnakin(const char*, size_t) { return false; }

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
