"""Writes a synthetic workload (anakin_amd.workloads) as the input files of integration/test_net_mi355x.cpp: the network
as ORIGINAL Caffe-style operators (conv / BatchNorm / Scale / ReLU / pooling / eltwise / fc / softmax) with the raw,
unfolded BatchNorm blobs — the form a `.anakin.bin` holds before the reference's optimiser runs — plus the per-node
precisions and per-variable scales a calibrator config would carry. Test infrastructure."""
import os

import numpy as np


def int8_plan(spec):
    """Node precisions for the INT8 graph: every operator 8-bit except the input and the softmax (the configuration the
    x86 edge rule can express end to end, calibrator_parse.cpp:82-128: an int8 -> fp32 edge would be f32, which the INT8
    eltwise cannot write, saber_eltwise.cpp:85-95)."""
    prec = {}
    for l in spec:
        prec[l["name"]] = "fp32" if l["kind"] == "softmax" else "int8"
    return prec


def calibrator_files(spec, scales, outdir):
    """The two text files Graph::load_calibrator_config reads (framework/graph/graph.cpp:555-571, parser
    framework/core/net/calibrator_parse.cpp:338-460), authored from the topology alone - what a user's calibration run hands to
    the framework next to the model:
      net_config.txt    one line per node of the FROZEN graph, `<node>(<op type>)    <precision>    <target>` (the format
                        CalibratorParser::auto_config writes, :278-306): every node of an 8-bit layer `int8` (`uint8` for a conv with a fused relu),
                        the softmax `fp32`;
      calibrator.txt    one line per edge carrying a calibrated tensor, `<bottom>_<top> <scale>` (Arc::name(); edges without a
                        line get the parser's default 1.0).
    Node names are the ones integration/test_net_mi355x.cpp gives the original operators (conv `x` -> `x`, `bn_x`, `scale_x`,
    `x_relu`; output variable `x_out`; Graph::Freeze names the Split behind a variable with several readers `<var>split` and the
    Output node after the variable, graph.cpp:237-296)."""
    prec = int8_plan(spec)
    nodes, last, first = [], {"data": "data"}, {}
    for l in spec:
        kd, nm = l["kind"], l["name"]
        p = prec[nm]
        if kd == "conv" and l["relu"] and p == "int8":
            # what Net::init's auto-configuration (AutoLayoutConfigHelper::auto_config_node_dtype, auto_layout_config.cpp:108-133)
            # makes of an INT8 conv with a fused relu; stated in the file so that a Net initialised WITHOUT it - Worker's
            # init(graph), worker.cpp:28 - runs the same plan (calibrator_parse.cpp:73, calibrator_factory.h:35: an INT8 operator)
            p = "uint8"
        if kd == "conv":
            bn = l.get("_bn", False)
            chain = [(nm, "Convolution")] + ([("bn_" + nm, "BatchNorm"), ("scale_" + nm, "Scale")] if bn else []) + \
                ([(nm + "_relu", "ReLU")] if l["relu"] else [])
        elif kd in ("pool", "gpool"):
            chain = [(nm, "Pooling")]
        elif kd == "eltwise":
            chain = [(nm, "Eltwise")] + ([(nm + "_relu", "ReLU")] if l["relu"] else [])
        elif kd == "fc":
            chain = [(nm, "Dense")] + ([(nm + "_relu", "ReLU")] if l.get("relu") else [])
        else:
            chain = [(nm, "Softmax")]
        nodes += [(n_, op, p) for n_, op in chain]
        first[nm], last[nm] = chain[0][0], chain[-1][0]
    readers = {}
    for l in spec:
        for key in ("src", "a", "b"):
            if key in l:
                readers.setdefault(l[key], []).append(first[l["name"]])
    edges = []
    for src in ["data"] + [l["name"] for l in spec]:
        var = src if src == "data" else src + "_out"
        rd = readers.get(src, [])
        if not rd:
            rd = [var]                                   # the graph's output node, named after the variable
        if len(rd) > 1:
            split = var + "split"
            nodes.append((split, "Split", prec.get(src, "int8") if src != "data" else "fp32"))
            edges.append((last[src] + "_" + split, scales[src]))
            edges += [(split + "_" + r, scales[src]) for r in rd]
        else:
            edges.append((last[src] + "_" + rd[0], scales[src]))
    cfg, cal = os.path.join(outdir, "net_config.txt"), os.path.join(outdir, "calibrator.txt")
    with open(cfg, "w") as f:
        for n_, op, p in nodes:
            f.write("%s(%s)    %s    MI355X \n" % (n_, op, p))
    with open(cal, "w") as f:
        for e, s in edges:
            f.write("%s %.9g\n" % (e, s))
    return cfg, cal


def write_model(model, scales, batch, outdir, precision="int8", hw=224, calibrator_config=False, rename=None):
    """Returns (model.txt, weights.bin) paths. `scales`: name -> activation scale (workloads.calibrate).
    calibrator_config: INT8 precisions and scales reach the graph through Graph::load_calibrator_config text files
    (calibrator_files above) instead of SetOpPrec / SetVarScale records."""
    os.makedirs(outdir, exist_ok=True)
    spec, params, raw = model["spec"], model["params"], model.get("raw", {})
    if rename is not None:       # node names as the graph will carry them (short_names below); weights stay keyed by the model's names
        keys = ("name", "src", "a", "b", "eltwise")
        orig = {rename(l["name"]): l["name"] for l in spec}
        spec = [dict(l, **{k: rename(l[k]) for k in keys if k in l}) for l in spec]
        params = {rename(k): v for k, v in params.items()}
        raw = {rename(k): v for k, v in raw.items()}
        scales = {rename(k): v for k, v in scales.items()}
        assert len(orig) == len(spec), "rename must be injective"
    lines, blobs = ["precision " + precision, "weights weights.bin", "input data %d 3 %d %d" % (batch, hw, hw)], []
    consumers = {}
    for l in spec:
        for key in ("src", "a", "b"):
            if key in l:
                consumers[l[key]] = consumers.get(l[key], 0) + 1
    for l in spec:
        kd, nm = l["kind"], l["name"]
        if kd == "conv":
            bn = nm in raw
            lines.append("conv %s %s %d %d %d %d %d %d %d" % (nm, l["src"], l["cin"], l["cout"], l["k"], l["stride"], l["pad"],
                                                             int(l["relu"]), int(bn)))
            if bn:
                r = raw[nm]
                blobs += [r["w"], r["mean"], r["var"], r["gamma"], r["beta"]]
            else:
                blobs += list(params[nm])
        elif kd == "pool":
            lines.append("pool %s %s %s %d %d %d 0" % (nm, l["src"], "MAX" if l["type"] == 0 else "AVG", l["win"], l["stride"], l["pad"]))
        elif kd == "gpool":
            lines.append("pool %s %s AVG 7 7 0 1" % (nm, l["src"]))
        elif kd == "eltwise":
            # the INT8 eltwise ignores its output scale (saber_eltwise.cpp:85): the requantisation rides in the coefficients
            c = 1.0 / scales[nm] if precision == "int8" else 1.0
            lines.append("eltwise %s %s %s %d %.9g %.9g" % (nm, l["a"], l["b"], int(l["relu"]), c, c))
        elif kd == "fc":
            lines.append("fc %s %s %d %d %d" % (nm, l["src"], l["cin"], l["cout"], int(bool(l.get("relu")))))
            blobs += list(params[nm])
        elif kd == "softmax":
            lines.append("softmax %s %s" % (nm, l["src"]))
    if precision == "int8" and calibrator_config:
        cspec = [dict(l, _bn=l["name"] in raw) for l in spec]
        cfg, cal = calibrator_files(cspec, scales, outdir)
        lines.append("calibrator %s %s" % (os.path.basename(cfg), os.path.basename(cal)))      # relative to the model file
    elif precision == "int8":
        prec = int8_plan(spec)
        for l in spec:
            lines.append("prec %s %s" % (l["name"], prec[l["name"]]))
            if consumers.get(l["name"], 0) > 1:      # Graph::Freeze inserts `<var>split` (graph.cpp:262-284)
                lines.append("precsplit %s %s" % (l["name"], prec[l["name"]]))
        lines.append("scale data %.9g" % scales["data"])
        for l in spec:
            lines.append("scale %s %.9g" % (l["name"], scales[l["name"]]))
    mt, wb = os.path.join(outdir, "model.txt"), os.path.join(outdir, "weights.bin")
    open(mt, "w").write("\n".join(lines) + "\n")
    with open(wb, "wb") as f:
        for b in blobs:
            f.write(np.ascontiguousarray(b, np.float32).tobytes())
    return mt, wb


def short_names(name):
    """`res4b22_branch2c` -> `r4b22_2c`: node names of at most 15 characters. The reference's graph_strategy::apply_stride_up
    (framework/graph/llvm/optimizer/optimize_strategy.h:236) calls GraphBase::remove_byio (framework/graph/graph_base.inl:218-240),
    which erases the arc from the arc list and THEN compares `origin()->top()` / `->bottom()` of that erased list node through the
    stale iterators of the per-vertex arc tables (heap-use-after-free, found with AddressSanitizer: INTEGRATION.md). With names up to
    15 characters the strings live inside the freed node (small-string buffer) and the comparison still reads the old bytes; from 16
    characters on (ResNet101's `res4b10_branch2a` ...) their heap storage is gone, the wrong arcs stay, and Optimize() aborts
    later at graph_base.inl:99. Short names keep the reference's code on the path its authors exercised."""
    return name.replace("res", "r", 1).replace("_branch", "_") if name.startswith("res") else name


def parse_oplist(path):
    """oplist.txt -> list of dicts {index, name, type, prec, ins: [...], outs: [...]}, edges as dicts."""
    ops = []
    for line in open(path):
        t = line.split()
        if not t:
            continue
        if t[0] == "op":
            ops.append(dict(index=int(t[1]), name=t[2], type=t[3], prec=t[4], ins=[], outs=[]))
        else:
            shape = [int(v) for v in t[4].strip("[]").split(",")]
            e = dict(edge=t[1], dtype=t[2], layout=t[3], shape=shape, scale=float(t[6]), ptr=t[8],
                     shared_from=t[10] if len(t) > 10 else None)
            ops[-1]["ins" if t[0] == "in" else "outs"].append(e)
    return ops
