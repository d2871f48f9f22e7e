#!/usr/bin/env python
"""Builds a PATCHED COPY of the reference's Saber library with the MI355X target added — the saber half of
docs/Manual/addCustomDevice.md:59-280 — in a scratch directory (default integration/_build/anakin, git-ignored).

    python integration/apply_mi355x_target.py [/root/reference] [integration/_build/anakin]

Nothing of the reference is stored in this repository: the script copies `saber/`, `utils/` and the test helper header
from the reference tree where it lies, then makes the one-to-five-line insertions below (each located by a short
anchor string) and drops in this repository's own files:

    saber/saber_types.h                      enum eMI355X + typedef TargetType<eMI355X> MI355X
    saber/core/target_traits.h               struct __mi355x_device + TargetTypeTraits<MI355X> (device target)
    saber/core/target_wrapper.h              #include of impl/mi355x/mi355x_target_wrapper.h (as the MLU target does, :691-693)
    saber/funcs/timer.h                      #include of impl/mi355x/mi355x_timer.h
    saber/funcs/conv.h, conv_eltwise.h       facade ladders: #include of impl/mi355x/saber_conv[_eltwise].h (pattern conv.h:23-50)
    + saber/core/impl/mi355x/{mi355x_target_wrapper.h, mi355x_impl.cpp}, saber/funcs/impl/mi355x/{saber_conv.h,
      mi355x_timer.h}                        (integration/mi355x/*: TargetWrapper / Device / SaberTimer on HIP, SaberConv2D)

The framework half of the manual (:281-455) is applied to a copy of `framework/` as well (`patch_framework`):

    framework/core/parameter.h               #include of framework/core/mi355x_pblock.h (PBlock<MI355X>: device tensor + host mirror)
    framework/core/data_types.h              ANAKIN_PBLOCK_TO_TYPE_ID(MI355X, anakin_block)
    framework/core/type_traits_extend.h      TARGET_NAME_SET(saber::MI355X, ...)   (target_host<MI355X> is the X86 default)
    framework/core/operator/operator_attr.cpp, core/net/{net,operator_func,auto_layout_config}.cpp, graph/graph.cpp,
    utils/parameter_fusion.cpp               the `template class ...<MI355X, ...>` instantiation lines
    framework/core/net/net.h, net.cpp, calibrator_parse.cpp
                                             MI355X takes the x86 branch of the edge dtype / layout rule (8-bit edges NHWC,
                                             conv+relu outputs u8; calibrator_parse.cpp:82-128,194-244) and of the automatic
                                             node-dtype configuration (net.cpp:76)
    framework/graph/graph.cpp                MI355X makes the x86 choices in Optimize(): no ConvReluPool patterns, no horizontal
                                             combination, INT8 conv+eltwise fusion off (graph.cpp:375-436) -> the SAME op list
    framework/operators/<the ResNet/VGG operators>.cpp
                                             the MI355X twin of every X86 INSTANCE_* / ANAKIN_REGISTER_OP_HELPER / __alias__ line.
                                             In this scratch copy the twin REPLACES the X86 line (token substitution): the x86
                                             Saber implementations need xbyak / mkl-dnn, which this container does not have,
                                             so Operator<X86> cannot be linked here; X86 stays the HOST target only.
    framework/model_parser/parser/parser.cpp replaced by integration/mi355x/framework/text_model_parser.cpp (protobuf is absent and this
                                             is its only consumer): Graph::load(path) reads a TEXT model - the network as original
                                             operators + raw weight blobs - and builds the graph with Graph::AddOp / AddOpAttr / Freeze
    + the facade ladders of saber/funcs/{pooling,eltwise,fc,softmax,activation,conv_pooling,gemm}.h"""
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def insert(path, anchor, text, after=True, count=1):
    s = open(path).read()
    assert s.count(anchor) >= 1, (path, anchor)
    i = s.index(anchor) if count == 1 else s.rindex(anchor)
    pos = i + len(anchor) if after else i
    open(path, "w").write(s[:pos] + text + s[pos:])


# the operators of the ResNet / VGG graphs (before and after the reference's fusion pass)
OPERATORS = ["input", "output", "split", "gather", "convolution", "relu", "pooling", "eltwise_op", "dense", "softmax",
             "fusion_ops/conv_batchnorm_scale", "fusion_ops/conv_batchnorm_scale_relu", "fusion_ops/conv_relu",
             "fusion_ops/eltwise_relu", "fusion_ops/conv_eltwise"]


def sub(path, old, new, count=1):
    s = open(path).read()
    assert s.count(old) >= 1, (path, old)
    open(path, "w").write(s.replace(old, new, count if count else -1))


def instantiate(path, template):
    """Adds the MI355X lines next to a file's X86 explicit instantiations (`template class Foo<X86, ...>;`)."""
    import re
    s = open(path).read()
    lines = sorted(set(re.findall(template, s)))
    assert lines, (path, template)
    add = "\n#ifdef USE_MI355X_PLACE\n" + "".join(l.replace("X86", "MI355X") + "\n" for l in lines) + "#endif\n"
    i = s.index("#endif", s.rindex(lines[-1])) + len("#endif")      # right after the X86 block
    open(path, "w").write(s[:i] + add + s[i:])


def patch_framework(ref, dst):
    F = os.path.join(dst, "framework")
    shutil.copytree(os.path.join(ref, "framework"), F,
                    ignore=shutil.ignore_patterns("service", "c_api", "nanopb", "proto", "lite", "*.pb.*", "CMakeLists.txt"))
    shutil.copy(os.path.join(HERE, "mi355x", "framework", "mi355x_pblock.h"), os.path.join(F, "core"))
    shutil.copy(os.path.join(HERE, "mi355x", "framework", "text_model_parser.cpp"),
                os.path.join(F, "model_parser", "parser", "parser.cpp"))
    # model_io.cpp (NodeIO over protoc-generated classes) gives way to the wire-format reader / writer: `.anakin.bin` without libprotobuf
    shutil.copy(os.path.join(HERE, "mi355x", "framework", "anakin_bin_parser.cpp"),
                os.path.join(F, "model_parser", "parser", "model_io.cpp"))
    shutil.copy(os.path.join(HERE, "mi355x", "framework", "anakin_bin_model.h"),
                os.path.join(F, "model_parser", "parser", "anakin_bin_model.h"))
    insert(os.path.join(F, "core", "parameter.h"), "#endif",
           "#ifdef USE_MI355X_PLACE\n#include \"framework/core/mi355x_pblock.h\"\n#endif\n\n", after=False, count=-1)
    insert(os.path.join(F, "core", "data_types.h"), "\tANAKIN_PBLOCK_TO_TYPE_ID(X86, anakin_block)\n#endif\n",
           "#ifdef USE_MI355X_PLACE\n\tANAKIN_PBLOCK_TO_TYPE_ID(MI355X, anakin_block)\n#endif\n")
    insert(os.path.join(F, "core", "type_traits_extend.h"), "TARGET_NAME_SET(saber::X86, saber_X86)\n",
           "TARGET_NAME_SET(saber::MI355X, saber_MI355X)\n")
    insert(os.path.join(F, "core", "operator", "operator_attr.cpp"), "//#ifdef USE_BM_PLACE\n",
           "".join("template\nOpAttrWarpper& OpAttrWarpper::__alias__<MI355X, Precision::%s>(const std::string& op_name);\n" % p
                   for p in ("FP32", "FP16", "INT8")), after=False)
    net = os.path.join(F, "core", "net")
    instantiate(os.path.join(net, "net.cpp"), r"template class Net<X86, [^;]*;")
    instantiate(os.path.join(net, "operator_func.cpp"), r"template class OperatorFunc<X86, [^;]*;")
    instantiate(os.path.join(net, "auto_layout_config.cpp"), r"template class AutoLayoutConfigHelper<X86, [^;]*;")
    # Worker<MI355X, P, R> (the multi-instance serving shape: one Net per thread, framework/core/net/worker.h:38-60) and the
    # calibration-table generator (EntropyCalibrator / BatchStream, framework/core/net/entropy_calibrator.cpp, calibrator.h)
    instantiate(os.path.join(net, "worker.cpp"), r"template class Worker<X86, [^;]*;")
    # a Worker with several pool threads = several Nets in flight on one GPU: its constructor declares that to the MI355X plans BEFORE
    # the pool threads initialise their Nets (a stream per Net, SABER_HIP_NET_SHARED_DEVICE: no placement-dependent kernel variants) -
    # mi355x_net_plan.h: MI355XNetPlanDefaults. Other targets: a no-op.
    insert(os.path.join(net, "worker.cpp"), "namespace anakin {\n",
           "\ntemplate <typename T> static inline void mi355x_worker_defaults(int) {}\n"
           "#ifdef USE_MI355X_PLACE\n"
           "template <> inline void mi355x_worker_defaults<saber::MI355X>(int threads) { MI355XNetPlanDefaults::worker_threads(threads); }\n"
           "#endif\n")
    sub(os.path.join(net, "worker.cpp"), "_model_path(model_path), ThreadPool(num_thread) {}",
        "_model_path(model_path), ThreadPool(num_thread) { mi355x_worker_defaults<Ttype>(num_thread); }")
    instantiate(os.path.join(net, "entropy_calibrator.cpp"), r"template class EntropyCalibrator<X86>;")
    instantiate(os.path.join(net, "batch_stream.cpp"), r"template class BatchStream<X86>;")
    instantiate(os.path.join(F, "graph", "graph.cpp"), r"template class Graph<X86, [^;]*;")
    instantiate(os.path.join(F, "utils", "parameter_fusion.cpp"), r"template class WeightsFusion<[a-z]*, X86>;")
    # the x86 edge rule for MI355X: Net picks the branch by target type, CalibratorParser by name
    sub(os.path.join(net, "net.cpp"), "auto_layout_config && std::is_same<Ttype, X86>::value",
        "auto_layout_config && (std::is_same<Ttype, X86>::value || std::is_same<Ttype, MI355X>::value)")
    sub(os.path.join(net, "net.h"), "        if (std::is_same<X86, Ttype>::value) {\n            edge_it->weight()->set_dtype(",
        "        if (std::is_same<MI355X, Ttype>::value) {\n"
        "            edge_it->weight()->set_dtype(_calibrator_parser.get_dtype(edge_it->bottom(), edge_it->top(),\n"
        "                                         bottom_op_name, top_op_name, \"MI355X\", (*_graph_p)[edge_it->bottom()]));\n"
        "        } else if (std::is_same<X86, Ttype>::value) {\n            edge_it->weight()->set_dtype(")
    sub(os.path.join(net, "net.h"), "        if (std::is_same<X86, Ttype>::value) {\n            //set tensor layout\n",
        "        if (std::is_same<MI355X, Ttype>::value) {\n"
        "            edge_it->weight()->set_layout(_calibrator_parser.get_layout(edge_it->bottom(), edge_it->top(),\n"
        "                                          _calibrator_parser.get_layout(edge_it->name()), \"mi355x\", bottom_op_name,\n"
        "                                          top_op_name, (*_graph_p)[edge_it->bottom()]));\n"
        "        } else if (std::is_same<X86, Ttype>::value) {\n            //set tensor layout\n")
    # Net<MI355X>::prediction() as one executor call (mi355x_net_plan.h): the plan is a member of Net, built at the end of
    # init() from the op loop run under saber_hip_capture_begin / _end, dropped when init() runs again
    for n in ("mi355x_net_plan.h", "mi355x_net_planner.h", "mi355x_created_nodes.h"):
        shutil.copy(os.path.join(HERE, "mi355x", "framework", n), net)
    insert(os.path.join(net, "net.h"), "namespace anakin {", "#include \"framework/core/net/mi355x_net_plan.h\"\n\n", after=False)
    sub(os.path.join(net, "net.h"), "    OperatorFunc<Ttype, Ptype>* _fusion{nullptr};\n};\n",
        "    OperatorFunc<Ttype, Ptype>* _fusion{nullptr};\n\n"
        "    template <typename T_, Precision P_, OpRunType R_> friend struct MI355XPlanner;\n"
        "    MI355XNetPlan _mi355x_plan;\npublic:\n"
        "    ///< the MI355X target's captured op list behind prediction() (inspection / switching it off: plan.enabled = false; plan.drop())\n"
        "    MI355XNetPlan& mi355x_plan() { return _mi355x_plan; }\n};\n\n"
        "}\n#include \"framework/core/net/mi355x_net_planner.h\"\nnamespace anakin {\n")
    sub(os.path.join(net, "net.cpp"), "    init_env(graph);\n    // shallow copy\n",
        "    _mi355x_plan.drop();\n    mi355x_created_nodes_inherit(graph);\n    init_env(graph);\n    // shallow copy\n", count=0)
    insert(os.path.join(net, "net.cpp"), "namespace anakin {", "#include \"framework/core/net/mi355x_created_nodes.h\"\n\n", after=False)
    sub(os.path.join(net, "net.cpp"), "    init_memory();\n\n    graph.statistics = _graph_p->statistics; // copy statistic back\n",
        "    init_memory();\n    MI355XPlanner<Ttype, Ptype, RunType>::prepare(*this);\n\n"
        "    graph.statistics = _graph_p->statistics; // copy statistic back\n")
    sub(os.path.join(net, "net.cpp"), "void Net<Ttype, Ptype, RunType>::prediction() {\n",
        "void Net<Ttype, Ptype, RunType>::prediction() {\n"
        "    if (MI355XPlanner<Ttype, Ptype, RunType>::run(*this)) {\n        return;\n    }\n")
    cp = os.path.join(net, "calibrator_parse.cpp")
    s = open(cp).read()
    # get_dtype: the "X86" block, once more for "MI355X" with PBlock<MI355X>
    a = s.index("    if (dev_name == \"X86\") {")
    b = s.index("    if (bottom_op_type == \"Input\") {")
    twin = s[a:b].replace('"X86"', '"MI355X"').replace("USE_X86_PLACE", "USE_MI355X_PLACE").replace("PBlock<X86>", "PBlock<MI355X>")
    s = s[:b] + twin + s[b:]
    # get_layout: the "x86" branch, once more for "mi355x"
    a = s.index("    if (target_type == \"x86\") {")
    b = s.index("    } else {\n        LOG(FATAL) << \"not support target type \"")
    twin = s[a:b].replace('"x86"', '"mi355x"').replace("USE_X86_PLACE", "USE_MI355X_PLACE").replace("PBlock<X86>", "PBlock<MI355X>")
    s = s[:b] + "    } else " + twin.lstrip() + s[b:]
    open(cp, "w").write(s)
    # Optimize(): the x86 choices
    g = os.path.join(F, "graph", "graph.cpp")
    sub(g, "if (std::is_same<Ttype, X86>::value &&\n                        (fusion_name == \"ConvReluPool\"",
        "if ((std::is_same<Ttype, X86>::value || std::is_same<Ttype, MI355X>::value) &&\n"
        "                        (fusion_name == \"ConvReluPool\"")
    sub(g, "            if (std::is_same<Ttype,X86>::value) {\n", "            if (std::is_same<Ttype,X86>::value || std::is_same<Ttype,MI355X>::value) {\n")
    sub(g, "if ((std::is_same<Ttype, NV>::value||std::is_same<Ttype, X86>::value) && Precision::INT8 == Ptype) {",
        "if ((std::is_same<Ttype, NV>::value||std::is_same<Ttype, X86>::value||std::is_same<Ttype, MI355X>::value) && Precision::INT8 == Ptype) {")
    # operators: the MI355X twin of every X86 line (see the module docstring for why it replaces it here)
    import re
    for op in OPERATORS:
        for ext in (".cpp", ".h"):
            f = os.path.join(F, "operators", op + ext)
            s = open(f).read()
            open(f, "w").write(re.sub(r"\bX86\b", "MI355X", s))


def sync(tmp, dst):
    """Moves the freshly generated tree over `dst`, keeping the old file (and its mtime) wherever the content is unchanged,
    so that an incremental build only recompiles what really changed."""
    import filecmp
    for root, _, files in os.walk(tmp):
        rel = os.path.relpath(root, tmp)
        os.makedirs(os.path.join(dst, rel), exist_ok=True)
        for f in files:
            a, b = os.path.join(root, f), os.path.join(dst, rel, f)
            if not (os.path.exists(b) and filecmp.cmp(a, b, shallow=False)):
                shutil.copy(a, b)
    for root, _, files in os.walk(dst):
        rel = os.path.relpath(root, dst)
        for f in files:
            if not os.path.exists(os.path.join(tmp, rel, f)):
                os.remove(os.path.join(root, f))
    shutil.rmtree(tmp)


def main(ref, final_dst):
    dst = final_dst.rstrip("/") + ".tmp"
    if os.path.exists(dst):
        shutil.rmtree(dst)
    os.makedirs(dst)
    for d in ("saber", "utils"):
        shutil.copytree(os.path.join(ref, d), os.path.join(dst, d),
                        ignore=shutil.ignore_patterns("*.cu", "*.cl", "*.S", "arm", "mlu", "bm", "amd", "cuda"))
    patch_framework(ref, dst)
    S = os.path.join(dst, "saber")
    insert(os.path.join(S, "saber_types.h"), "    eMLUHX86 = 13,\n", "    eMI355X = 14,\n")
    insert(os.path.join(S, "saber_types.h"), "typedef TargetType<eMLUHX86> MLUHX86;\n",
           "typedef TargetType<eMI355X> MI355X;   // AMD Instinct MI355X (gfx950), HIP runtime\n")
    insert(os.path.join(S, "core", "target_traits.h"), "struct __mlu_device {};\n", "struct __mi355x_device {};\n")
    insert(os.path.join(S, "core", "target_traits.h"), "} //namespace saber",
           "template <>\nstruct TargetTypeTraits<MI355X> {\n    typedef __device_target target_category;\n"
           "    typedef __mi355x_device target_type;\n};\n", after=False)
    insert(os.path.join(S, "core", "target_wrapper.h"), "#endif //ANAKIN_SABER_CORE_TARGET_WRAPPER_H",
           "#ifdef USE_MI355X_PLACE\n#include \"saber/core/impl/mi355x/mi355x_target_wrapper.h\"\n#endif\n\n", after=False)
    insert(os.path.join(S, "funcs", "timer.h"), "#endif //SABER_TIMER_H",
           "#ifdef USE_MI355X_PLACE\n#include \"saber/funcs/impl/mi355x/mi355x_timer.h\"\n#endif\n\n", after=False)
    insert(os.path.join(S, "funcs", "conv.h"), "namespace anakin {",
           "#ifdef USE_MI355X_PLACE\n#include \"saber/funcs/impl/mi355x/saber_conv.h\"\n#endif\n\n", after=False)
    insert(os.path.join(S, "funcs", "conv_eltwise.h"), "namespace anakin {",
           "#ifdef USE_MI355X_PLACE\n#include \"saber/funcs/impl/mi355x/saber_conv_eltwise.h\"\n#endif\n\n", after=False)
    for facade, impl in (("pooling", "saber_pooling"), ("eltwise", "saber_eltwise"), ("fc", "saber_fc"),
                         ("softmax", "saber_softmax"), ("activation", "saber_activation"),
                         ("conv_pooling", "saber_conv_pooling")):
        insert(os.path.join(S, "funcs", facade + ".h"), "namespace anakin",
               "#ifdef USE_MI355X_PLACE\n#include \"saber/funcs/impl/mi355x/%s.h\"\n#endif\n\n" % impl, after=False)
    # gemm.h includes its targets' specialisations at the END of the file (gemm.h:95-105)
    insert(os.path.join(S, "funcs", "gemm.h"), "#ifdef USE_X86_PLACE\n#include \"saber/funcs/impl/x86/vender_gemm.h\"",
           "#ifdef USE_MI355X_PLACE\n#include \"saber/funcs/impl/mi355x/saber_gemm.h\"\n#endif\n\n", after=False)
    for subdir, names in (("core", ("mi355x_target_wrapper.h", "mi355x_impl.cpp")),
                          ("funcs", ("saber_conv.h", "saber_conv_eltwise.h", "mi355x_timer.h", "saber_pooling.h",
                                     "saber_eltwise.h", "saber_fc.h", "saber_softmax.h", "saber_activation.h",
                                     "saber_conv_pooling.h", "saber_gemm.h"))):
        d = os.path.join(S, subdir, "impl", "mi355x")
        os.makedirs(d, exist_ok=True)
        for n in names:
            shutil.copy(os.path.join(HERE, "mi355x", subdir, n), d)
    sync(dst, final_dst)
    print("patched Saber + framework trees with the MI355X target:", final_dst)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "/root/reference",
         sys.argv[2] if len(sys.argv) > 2 else os.path.join(HERE, "_build", "anakin"))
