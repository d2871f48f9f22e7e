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


def write_model(model, scales, batch, outdir, precision="int8", hw=224):
    """Returns (model.txt, weights.bin) paths. `scales`: name -> activation scale (workloads.calibrate)."""
    os.makedirs(outdir, exist_ok=True)
    spec, params, raw = model["spec"], model["params"], model.get("raw", {})
    lines, blobs = ["precision " + precision, "input data %d 3 %d %d" % (batch, hw, hw)], []
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
    if precision == "int8":
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
