// tests/cpp_host/anakin_bin_tool.cpp - integration/mi355x/framework/anakin_bin_model.h on its own (no reference headers, no HIP): decode a
// GraphProto and encode it again, so that tests/test_anakin_bin.py can hold both directions against the official protobuf runtime.
//   anakin_bin_tool reencode <in> <out>     exit 0: decoded and written; 3: not a well-formed GraphProto
//   anakin_bin_tool summary <in>            nodes / attributes / weight floats / edges, one line
#include <cstdio>
#include <fstream>
#include <string>

#include "anakin_bin_model.h"

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    std::ifstream f(argv[2], std::ios::binary | std::ios::ate);
    if (!f) return 2;
    std::string bytes((size_t)f.tellg(), '\0');
    f.seekg(0);
    f.read(&bytes[0], (std::streamsize)bytes.size());
    anakin_bin::Graph g;
    if (!anakin_bin::decode((const unsigned char*)bytes.data(), bytes.size(), g)) return 3;
    const std::string mode = argv[1];
    if (mode == "reencode" && argc > 3) {
        const std::string out = anakin_bin::encode(g);
        std::ofstream o(argv[3], std::ios::binary | std::ios::trunc);
        o.write(out.data(), (std::streamsize)out.size());
        return o.good() ? 0 : 2;
    }
    if (mode == "summary") {
        size_t attrs = 0, floats = 0, targets = 0;
        for (auto& n : g.nodes) {
            attrs += n.attr.size();
            for (auto& kv : n.attr) floats += kv.second.tensor.data.f.size();
        }
        for (auto& kv : g.edges_in) targets += kv.second.target.size() + kv.second.val.size();
        printf("name %s nodes %zu attrs %zu weight_floats %zu in_edges %zu ins %zu outs %zu optimized %d\n", g.name.c_str(), g.nodes.size(), attrs, floats, targets,
               g.ins.size(), g.outs.size(), (int)g.is_optimized);
        return 0;
    }
    return 2;
}
