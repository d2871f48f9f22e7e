"""`.anakin.bin` model files without a protobuf library: the reference's model format (protobuf wire format over
framework/model_parser/proto/{graph,node,tensor,operator}.proto) read into / written from plain dicts, and the ResNet / VGG operator
family converted to and from the model dicts of `anakin_amd.workloads`.

Role in the reference: framework/model_parser/parser/parser.cpp:131-240 (GraphProto -> Graph) and model_io.cpp:9-300 (NodeProto -> Node:
attributes by their DateTypeProto tag, weight tensors with shape + packed float payload); the files themselves are written by
tools/external_converter_v2 (parser/graph_io.py: NodeAttrWrapper, TensorProtoIO, GraphProtoIO) - a FROZEN graph: one node per original
operator plus Input / Split / Output nodes, `edges_in` / `edges_out` as per-node target lists. The C++ side of this repository reads the
same files through integration/mi355x/framework/anakin_bin_model.h; both are checked against the official protobuf runtime in
tests/test_anakin_bin.py.

    graph = read_graph("resnet50.anakin.bin")             # every field of the GraphProto, as dicts / lists / numpy arrays
    model = load_model("resnet50.anakin.bin")             # -> workloads model dict (spec / params / raw, + "input_shape", "scales", "precisions")
    write_model(model, "resnet50.anakin.bin", batch=1)    # the file the converter would write for this network
"""
import struct

import numpy as np

# DateTypeProto (tensor.proto)
STR, INT8, INT32, FLOAT16, FLOAT, DOUBLE, BOOLEN, CACHE_LIST, TENSOR = 0, 2, 4, 8, 13, 14, 20, 30, 31
LP_NCHW = 8

# message -> {field number: (name, kind, repeated)}; kind: string | bytes | int | bool | float | a message name | ("map", value message)
SCHEMA = {
    "Dim": {1: ("value", "int", True), 2: ("size", "int", False)},
    "TensorShape": {3: ("dim", "Dim", False)},
    "CacheDate": {1: ("s", "bytes", True), 2: ("i", "int", True), 3: ("f", "float", True), 4: ("b", "bool", True), 5: ("l", "CacheDate", True),
                  6: ("type", "int", False), 7: ("size", "int", False), 8: ("c", "bytes", False)},
    "TensorProto": {1: ("name", "bytes", False), 2: ("shared", "bool", False), 3: ("share_from", "bytes", False), 8: ("shape", "TensorShape", False),
                    9: ("valid_shape", "TensorShape", False), 10: ("data", "CacheDate", False), 11: ("scale", "CacheDate", False)},
    "valueType": {1: ("s", "bytes", False), 2: ("i", "int", False), 3: ("f", "float", False), 4: ("b", "bool", False), 8: ("cache_list", "CacheDate", False),
                  10: ("tensor", "TensorProto", False), 14: ("type", "int", False)},
    "OpProto": {1: ("name", "string", False), 2: ("is_commutative", "bool", False), 3: ("in_num", "int", False), 4: ("out_num", "int", False),
                5: ("description", "string", False)},
    "NodeProto": {1: ("name", "string", False), 2: ("ins", "string", True), 3: ("outs", "string", True), 10: ("attr", ("map", "valueType"), True),
                  11: ("lane", "int", False), 12: ("need_wait", "bool", False), 15: ("Op", "OpProto", False), 16: ("bit_type", "int", False)},
    "TargetProto": {1: ("node", "string", False), 2: ("scale", "float", True), 3: ("layout", "int", False)},
    "List": {1: ("val", "string", True), 2: ("target", "TargetProto", True)},
    "Version": {1: ("major", "int", False), 2: ("minor", "int", False), 3: ("patch", "int", False), 4: ("version", "int", False)},
    "Info": {1: ("temp_mem_used", "int", False), 2: ("original_temp_mem_used", "int", False), 3: ("system_mem_used", "int", False),
             4: ("model_mem_used", "int", False), 10: ("is_optimized", "bool", False)},
    "GraphProto": {1: ("name", "string", False), 2: ("nodes", "NodeProto", True), 3: ("edges_in", ("map", "List"), True), 4: ("edges_out", ("map", "List"), True),
                   5: ("edges_info", ("map", "TensorProto"), True), 6: ("ins", "string", True), 7: ("outs", "string", True), 10: ("version", "Version", False),
                   11: ("summary", "Info", False)},
}
# valueType's `oneof data`: the member a type tag selects is written even at its default
ONEOF = {STR: "s", INT32: "i", FLOAT: "f", DOUBLE: "f", BOOLEN: "b", CACHE_LIST: "cache_list", TENSOR: "tensor"}


class FormatError(ValueError):
    pass


def _varint(buf, pos):
    v, shift = 0, 0
    while True:
        if pos >= len(buf):
            raise FormatError("truncated varint")
        b = buf[pos]
        pos += 1
        v |= (b & 0x7F) << shift
        if not b & 0x80:
            return v, pos
        shift += 7
        if shift > 63:
            raise FormatError("varint longer than 10 bytes")


def _signed(v, bits=64):
    v &= (1 << 64) - 1
    if v >> 63:
        v -= 1 << 64
    if bits == 32:
        v = ((v + (1 << 31)) & 0xFFFFFFFF) - (1 << 31)
    return v


def _decode(buf, msg, depth=0):
    if depth > 8:                                       # CacheDate lists nested deeper than any model holds: refused, not recursed into
        raise FormatError("lists nested more than 8 deep")
    fields, out, pos = SCHEMA[msg], {}, 0
    while pos < len(buf):
        key, pos = _varint(buf, pos)
        num, wt = key >> 3, key & 7
        if num == 0:
            raise FormatError("field number 0")
        if wt == 0:
            raw, pos = _varint(buf, pos)
        elif wt == 1:
            raw, pos = buf[pos:pos + 8], pos + 8
            if len(raw) != 8:
                raise FormatError("truncated 8-byte field")
        elif wt == 2:
            n, pos = _varint(buf, pos)
            raw, pos = buf[pos:pos + n], pos + n
            if len(raw) != n:
                raise FormatError("a length runs past its message")
        elif wt == 5:
            raw, pos = buf[pos:pos + 4], pos + 4
            if len(raw) != 4:
                raise FormatError("truncated 4-byte field")
        else:
            raise FormatError("wire type %d" % wt)
        if num not in fields:
            continue                                    # unknown field: skipped
        name, kind, rep = fields[num]
        if isinstance(kind, tuple):                     # map entry {1: key, 2: value}
            if wt != 2:
                continue
            e = _decode_entry(raw, kind[1])
            out.setdefault(name, {})[e[0]] = e[1]
            continue
        # a known field that arrives with a wire type its kind cannot have is an UNKNOWN field (skipped), as in every protobuf parser;
        # repeated numeric scalars are accepted packed (2) or one by one
        if kind == "float":
            if wt == 2 and rep:
                if len(raw) % 4:
                    raise FormatError("packed floats: %d bytes" % len(raw))
                val = np.frombuffer(bytes(raw), dtype="<f4")
            elif wt == 5:
                val = np.frombuffer(bytes(raw), dtype="<f4")
            else:
                continue
            if rep:
                out[name] = np.concatenate([out[name], val]) if name in out else val
            else:
                out[name] = float(val[0])
        elif kind in ("int", "bool"):
            if wt == 2 and rep:                         # packed
                vals, p = [], 0
                while p < len(raw):
                    v, p = _varint(raw, p)
                    vals.append(_signed(v))
            elif wt == 0:
                vals = [_signed(raw)]
            else:
                continue
            if kind == "bool":
                vals = [bool(v) for v in vals]
            if rep:
                out.setdefault(name, []).extend(vals)
            else:
                out[name] = vals[-1]
        elif kind in ("string", "bytes"):
            if wt != 2:
                continue
            val = bytes(raw).decode("utf-8", "surrogateescape") if kind == "string" else bytes(raw)
            if rep:
                out.setdefault(name, []).append(val)
            else:
                out[name] = val
        else:
            if wt != 2:
                continue
            val = _decode(raw, kind, depth + 1 if (msg == "CacheDate" and kind == "CacheDate") else depth)
            if rep:
                out.setdefault(name, []).append(val)
            else:
                out[name] = val
    return out


def _decode_entry(buf, value_msg):
    key, val, pos = "", {}, 0
    while pos < len(buf):
        k, pos = _varint(buf, pos)
        wt = k & 7
        if k >> 3 == 0:
            raise FormatError("field number 0")
        if wt == 0:
            _, pos = _varint(buf, pos)
            continue
        if wt in (1, 5):
            n = 8 if wt == 1 else 4
            if pos + n > len(buf):
                raise FormatError("truncated fixed field in a map entry")
            pos += n
            continue
        if wt != 2:
            raise FormatError("wire type %d" % wt)
        n, pos = _varint(buf, pos)
        raw, pos = buf[pos:pos + n], pos + n
        if len(raw) != n:
            raise FormatError("a map entry runs past its message")
        if k >> 3 == 1:
            key = bytes(raw).decode("utf-8", "surrogateescape")
        elif k >> 3 == 2:
            val = _decode(raw, value_msg)
    return key, val


def _put_varint(out, v):
    v &= (1 << 64) - 1
    while v >= 0x80:
        out.append((v & 0x7F) | 0x80)
        v >>= 7
    out.append(v)


def _put_len(out, num, payload):
    _put_varint(out, (num << 3) | 2)
    _put_varint(out, len(payload))
    out += payload


def _encode(d, msg):
    out = bytearray()
    always = ONEOF.get(d.get("type", 0)) if msg == "valueType" else None
    for num in sorted(SCHEMA[msg]):
        name, kind, rep = SCHEMA[msg][num]
        if name not in d and name != always:
            continue
        v = d.get(name)
        if isinstance(kind, tuple):
            for k in v:
                e = bytearray()
                _put_len(e, 1, k.encode("utf-8", "surrogateescape"))
                _put_len(e, 2, _encode(v[k], kind[1]))
                _put_len(out, num, e)
        elif kind == "float":
            if rep:
                a = np.ascontiguousarray(v, dtype="<f4")
                if a.size:
                    _put_len(out, num, a.tobytes())
            elif name == always or struct.pack("<f", v or 0.0) != b"\0\0\0\0":
                _put_varint(out, (num << 3) | 5)
                out += struct.pack("<f", v or 0.0)
        elif kind in ("int", "bool"):
            if rep:
                if len(v):
                    p = bytearray()
                    for x in v:
                        _put_varint(p, int(x))
                    _put_len(out, num, p)
            elif name == always or v:
                _put_varint(out, (num << 3) | 0)
                _put_varint(out, int(v or 0))
        elif kind in ("string", "bytes"):
            for x in (v if rep else [v]):
                x = x if x is not None else ""
                b = x.encode("utf-8", "surrogateescape") if isinstance(x, str) else bytes(x)
                if rep or b or name == always:
                    _put_len(out, num, b)
        else:
            for x in (v if rep else [v]):
                _put_len(out, num, _encode(x if x is not None else {}, kind))
    return out


def read_graph(path_or_bytes):
    """The GraphProto of an `.anakin.bin` as nested dicts (absent fields are absent keys; repeated floats are numpy arrays)."""
    buf = path_or_bytes if isinstance(path_or_bytes, (bytes, bytearray, memoryview)) else open(path_or_bytes, "rb").read()
    return _decode(memoryview(buf), "GraphProto")


def write_graph(graph, path=None):
    b = bytes(_encode(graph, "GraphProto"))
    if path is not None:
        with open(path, "wb") as f:
            f.write(b)
    return b


# ------------------------------------------------------------------------------------------------ attributes
def attr_value(v):
    """python value of a valueType (model_io.cpp:30-300): str / int / float / bool / list / numpy weight array (shared weights: the share_from name)"""
    t = v.get("type", 0)
    if t == STR:
        return v.get("s", b"").decode("utf-8", "surrogateescape")
    if t == INT32:
        return int(v.get("i", 0))
    if t in (FLOAT, DOUBLE):
        return float(v.get("f", 0.0))
    if t == BOOLEN:
        return bool(v.get("b", False))
    if t == CACHE_LIST:
        c = v.get("cache_list", {})
        n, ct = int(c.get("size", 0)), c.get("type", 0)
        if ct == FLOAT:
            return [float(x) for x in c.get("f", [])[:n]]
        if ct == INT32:
            return [int(x) for x in c.get("i", [])[:n]]
        if ct == BOOLEN:
            return [bool(x) for x in c.get("b", [])[:n]]
        if ct == STR:
            return [x.decode("utf-8", "surrogateescape") for x in c.get("s", [])[:n]]
        if ct == CACHE_LIST:
            return [[int(x) for x in l.get("i", [])[:int(l.get("size", 0))]] for l in c.get("l", [])]
        raise FormatError("list element type %d" % ct)
    if t == TENSOR:
        ts = v.get("tensor", {})
        if ts.get("shared"):
            return ("shared", ts.get("share_from", b"").decode())
        shape = [int(x) for x in ts.get("shape", {}).get("dim", {}).get("value", [])]
        data = ts.get("data", {})
        if data.get("type", 0) == FLOAT:
            a = np.asarray(data.get("f", np.zeros(0, np.float32)), dtype=np.float32)
        elif data.get("type", 0) == INT8:
            a = np.frombuffer(data.get("c", b""), dtype=np.int8)
        else:
            raise FormatError("weight payload type %d" % data.get("type", 0))
        if len(shape) != 4 or a.size < int(np.prod(shape)) or int(data.get("size", 0)) != int(np.prod(shape)):
            raise FormatError("weight tensor: %d values for shape %s" % (a.size, shape))
        return a[:int(np.prod(shape))].reshape(shape)
    raise FormatError("value type %d" % t)


def _val(x):
    """a valueType dict for a python value, as the converter's NodeAttrWrapper writes it (graph_io.py:19-80)"""
    if isinstance(x, bool):
        return {"b": x, "type": BOOLEN}
    if isinstance(x, int):
        return {"i": x, "type": INT32}
    if isinstance(x, float):
        return {"f": x, "type": FLOAT}
    if isinstance(x, str):
        return {"s": x.encode(), "type": STR}
    if isinstance(x, np.ndarray):
        a = np.ascontiguousarray(x, dtype=np.float32)
        assert a.ndim == 4
        return {"tensor": {"shape": {"dim": {"value": list(a.shape), "size": 4}}, "data": {"f": a.reshape(-1), "type": FLOAT, "size": int(a.size)}}, "type": TENSOR}
    if isinstance(x, (list, tuple)):
        if len(x) and isinstance(x[0], bool):
            return {"cache_list": {"b": list(x), "type": BOOLEN, "size": len(x)}, "type": CACHE_LIST}
        if len(x) and isinstance(x[0], int):
            return {"cache_list": {"i": list(x), "type": INT32, "size": len(x)}, "type": CACHE_LIST}
        if len(x) and isinstance(x[0], str):
            return {"cache_list": {"s": [s.encode() for s in x], "type": STR, "size": len(x)}, "type": CACHE_LIST}
        return {"cache_list": {"f": [float(v) for v in x], "type": FLOAT, "size": len(x)}, "type": CACHE_LIST}
    raise TypeError(type(x))


# ------------------------------------------------------------------------------------------------ workloads model <-> file
def write_model(model, path, batch=1, hw=224, precision="fp32", scales=None, rename=None, calibration_in_file=True):
    """The network of a workloads model dict as the `.anakin.bin` the converter writes: ORIGINAL operators (Convolution / BatchNorm / Scale /
    ReLU / Pooling / Eltwise / Dense / Softmax) with raw, unfolded blobs, frozen (Input / Split / Output nodes named as Graph::Freeze names
    them, graph.cpp:237-296: `<var>split`, the output after its variable) - node for node the graph the text model of
    integration/net_model.py builds. precision "int8" + scales: nodes carry bit_type INT8 and the edges their activation scale (what
    Graph::SetOpPrec / SetVarScale leave in the graph: graph.cpp:108-180); calibration_in_file=False leaves both out - the deployment then
    hands the two calibrator text files to Graph::load_calibrator_config - but the INT8 Eltwise's coefficients still carry 1 / its output
    scale (the INT8 eltwise ignores its output scale, saber_eltwise.cpp:85: the requantisation rides in the coefficients)."""
    R = rename or (lambda n: n)
    spec, params, raw = model["spec"], model["params"], model.get("raw", {})
    V = lambda n: n if n == "data" else R(n) + "_out"                                        # noqa: E731 - a layer's output variable
    nodes, produced = [], {}            # produced: variable -> (node, readers)
    int8 = precision == "int8"
    f9 = lambda x: float(np.float32(float("%.9g" % x)))      # noqa: E731 - the float a 9-digit decimal record becomes (the text model's route)

    def node(name, op, ins, outs, attrs, prec_of=None):
        # Graph::SetOpPrec is called for the node that carries the layer's name (and the Split behind it) - the BatchNorm / Scale / ReLU
        # nodes of a conv keep the default and take the conv's precision when Graph::Optimize fuses them (integration/net_model.py: `prec`)
        bt = INT8 if int8 and calibration_in_file and op in ("Convolution", "Pooling", "Eltwise", "Dense") else FLOAT
        n = {"name": name, "Op": {"name": op}, "attr": {k: _val(v) for k, v in attrs.items()}, "bit_type": bt, "_ins": list(ins), "_outs": list(outs)}
        nodes.append(n)
        return n

    node("data", "Input", [], ["data"], {"input_shape": [batch, 3, hw, hw]})
    one = np.ones((1, 1, 1, 1), np.float32)
    for l in spec:
        kd, nm = l["kind"], R(l["name"])
        if kd == "conv":
            bn, relu, cout = l["name"] in raw, bool(l["relu"]), l["cout"]
            attrs = {"group": 1, "bias_term": not bn, "padding": [l["pad"]] * 2, "strides": [l["stride"]] * 2, "dilation_rate": [1, 1], "filter_num": cout,
                     "kernel_size": [l["k"]] * 2, "axis": 1}
            if bn:
                r = raw[l["name"]]
                attrs["weight_1"] = r["w"]
            else:
                attrs["weight_1"] = params[l["name"]][0]
                attrs["weight_2"] = params[l["name"]][1].reshape(1, cout, 1, 1)
            cur = nm + "_conv" if (bn or relu) else V(l["name"])
            node(nm, "Convolution", [V(l["src"])], [cur], attrs)
            if bn:
                node("bn_" + nm, "BatchNorm", [cur], [nm + "_bn"], {"epsilon": 1e-5, "momentum": 0.999, "weight_1": r["mean"].reshape(1, cout, 1, 1),
                                                                 "weight_2": r["var"].reshape(1, cout, 1, 1), "weight_3": one})
                cur = nm + "_scale" if relu else V(l["name"])
                node("scale_" + nm, "Scale", [nm + "_bn"], [cur], {"num_axes": 1, "bias_term": True, "axis": 1, "weight_1": r["gamma"].reshape(1, cout, 1, 1),
                                                                   "weight_2": r["beta"].reshape(1, cout, 1, 1)})
            if relu:
                node(nm + "_relu", "ReLU", [cur], [V(l["name"])], {"alpha": 0.0})
        elif kd in ("pool", "gpool"):
            g = kd == "gpool"
            node(nm, "Pooling", [V(l["src"])], [V(l["name"])], {"method": "AVG" if g or l.get("type") else "MAX", "pool_size": [7, 7] if g else [l["win"]] * 2,
                                                               "strides": [7, 7] if g else [l["stride"]] * 2, "padding": [0, 0] if g else [l["pad"]] * 2,
                                                               "global_pooling": g, "cmp_out_shape_floor_as_conv": bool(l.get("floor", False))})
        elif kd == "eltwise":
            relu = bool(l["relu"])
            c = 1.0 / scales[l["name"]] if int8 and scales else 1.0
            node(nm, "Eltwise", [V(l["a"]), V(l["b"])], [nm + "_sum" if relu else V(l["name"])], {"type": "Add", "coeff": [f9(c)] * 2})
            if relu:
                node(nm + "_relu", "ReLU", [nm + "_sum"], [V(l["name"])], {"alpha": 0.0})
        elif kd == "fc":
            relu = bool(l.get("relu"))
            w, b = params[l["name"]]
            node(nm, "Dense", [V(l["src"])], [nm + "_fc" if relu else V(l["name"])], {"out_dim": l["cout"], "bias_term": True, "axis": 1,
                                                                                      "weight_1": w.reshape(1, 1, l["cout"], l["cin"]), "weight_2": b.reshape(1, l["cout"], 1, 1)})
            if relu:
                node(nm + "_relu", "ReLU", [nm + "_fc"], [V(l["name"])], {"alpha": 0.0})
        elif kd == "softmax":
            node(nm, "Softmax", [V(l["src"])], [V(l["name"])], {"axis": 1})
        else:
            raise ValueError(kd)
    # ---- freeze: variables -> node-to-node edges, a Split behind a variable with several readers, an Output behind one with none ----
    readers, writer = {}, {}
    for n in nodes:
        for v in n["_ins"]:
            readers.setdefault(v, []).append(n["name"])
        for v in n["_outs"]:
            writer[v] = n["name"]
    layer_of_var = {V(l["name"]): l["name"] for l in spec}
    layer_of_var["data"] = "data"
    var_scale = {v: f9(scales[layer_of_var[v]]) for v in layer_of_var if int8 and scales and calibration_in_file and layer_of_var[v] in scales}
    edges, outs = [], []           # (bottom node, top node, scale or None)
    for v, w in list(writer.items()):
        rd = readers.get(v, [])
        s = var_scale.get(v)
        if not rd:
            nodes.append({"name": v, "Op": {"name": "Output"}, "attr": {}, "bit_type": FLOAT, "_ins": [v], "_outs": []})
            edges.append((w, v, s))
            outs.append(v)
        elif len(rd) == 1:
            edges.append((w, rd[0], s))
        else:
            sp = v + "split"
            prec = INT8 if int8 and calibration_in_file else FLOAT
            nodes.append({"name": sp, "Op": {"name": "Split"}, "attr": {"split_num": _val(len(rd))}, "bit_type": prec, "_ins": [v], "_outs": []})
            edges.append((w, sp, s))
            edges += [(sp, r, s) for r in rd]
    order = {n["name"]: i for i, n in enumerate(nodes)}
    g = {"name": model.get("name", "net"), "nodes": [], "edges_in": {}, "edges_out": {}, "edges_info": {}, "ins": ["data"], "outs": outs,
         "version": {"major": 0, "minor": 1, "patch": 1}, "summary": {"is_optimized": False}}
    for n in nodes:
        g["nodes"].append({k: v for k, v in n.items() if not k.startswith("_")})
    # in-arcs of a node in the order of its inputs (an Eltwise's a before its b)
    def in_rank(e):
        n = nodes[order[e[1]]]
        for i, v in enumerate(n["_ins"]):
            if writer.get(v) == e[0] or (v + "split") == e[0]:
                return i
        return 0
    for e in sorted(edges, key=lambda e: (order[e[1]], in_rank(e))):
        t = {"node": e[0], "layout": LP_NCHW}
        if e[2] is not None:
            t["scale"] = [e[2]]
        g["edges_in"].setdefault(e[1], {"target": []})["target"].append(t)
    for e in sorted(edges, key=lambda e: (order[e[0]], order[e[1]])):
        t = {"node": e[1], "layout": LP_NCHW}
        if e[2] is not None:
            t["scale"] = [e[2]]
        g["edges_out"].setdefault(e[0], {"target": []})["target"].append(t)
        g["edges_info"][e[0] + "_" + e[1]] = {"name": (e[0] + "_" + e[1]).encode()}
    return write_graph(g, path)


def load_model(path_or_bytes):
    """A graph that is a well-formed GraphProto but not a usable network (an operator without its weights, an edge to a node that does not
    exist ...) raises FormatError naming what is missing, like a malformed file."""
    try:
        return _load_model(path_or_bytes)
    except (KeyError, IndexError) as e:
        raise FormatError("the graph lacks %s %r" % ("an attribute / node" if isinstance(e, KeyError) else "an element", e.args[0] if e.args else e)) from e


def _load_model(path_or_bytes):
    """An `.anakin.bin` of the ResNet / VGG operator family as a workloads model dict: `spec` (conv / pool / gpool / eltwise / fc / softmax
    with BatchNorm + Scale + ReLU folded into their conv entry the way Graph::Optimize's fusion does, graph.cpp:375-436), `params` (weights
    with BatchNorm + Scale folded by workloads.fold_bn = WeightsFusion::update_weights), `raw` (the unfolded blobs), plus "input_shape",
    "scales" (edge scales by layer, when the file carries them) and "precisions" (bit_type by layer). Any other operator raises FormatError."""
    from . import workloads as W
    g = read_graph(path_or_bytes)
    nodes = {n["name"]: n for n in g.get("nodes", [])}
    op = {k: n.get("Op", {}).get("name", "") for k, n in nodes.items()}
    attrs = {k: {a: attr_value(v) for a, v in n.get("attr", {}).items()} for k, n in nodes.items()}
    ins = {k: [t["node"] for t in l.get("target", [])] or list(l.get("val", [])) for k, l in g.get("edges_in", {}).items()}
    outs = {k: [t["node"] for t in l.get("target", [])] or list(l.get("val", [])) for k, l in g.get("edges_out", {}).items()}
    in_scale = {(t["node"], k): list(t.get("scale", [])) for k, l in g.get("edges_in", {}).items() for t in l.get("target", [])}

    def src_of(name, i=0):            # the compute node (or Input) behind input i of `name`, looking through Split nodes
        b = ins[name][i]
        while op[b] == "Split":
            b = ins[b][0]
        return b

    def only_reader(name, kind):      # the single reader of `name` when it is a `kind` node (fusable), else None
        rd = outs.get(name, [])
        return rd[0] if len(rd) == 1 and op[rd[0]] == kind else None

    spec, params, raw, scales, precs, alias = [], {}, {}, {}, {}, {}
    input_name = g["ins"][0]
    input_shape = attrs[input_name].get("input_shape", [1, 3, 224, 224])
    # execution order: the file lists nodes in the converter's (topological) order; make sure of it
    done, order = {input_name}, []
    pending = [n["name"] for n in g["nodes"] if n["name"] != input_name]
    while pending:
        progressed = False
        for nme in list(pending):
            if all(b in done for b in ins.get(nme, [])):
                order.append(nme)
                done.add(nme)
                pending.remove(nme)
                progressed = True
        if not progressed:
            raise FormatError("the graph has a cycle or a dangling input: %s" % pending[:3])
    absorbed = set()
    last_of = {}                      # layer name -> the LAST node of its fused chain (whose out-edges carry the layer's scale)

    def layer_src(name, i=0):
        s = src_of(name, i)
        return alias.get(s, s)
    for nme in order:
        if nme in absorbed:
            continue
        o, a = op[nme], attrs[nme]
        if o in ("Split", "Output", "Input"):
            continue
        tail = nme
        if o == "Convolution":
            w = a["weight_1"]
            cout, cin, k = int(w.shape[0]), int(w.shape[1]), int(w.shape[2])
            if a.get("group", 1) != 1 or list(a.get("dilation_rate", [1, 1])) != [1, 1] or w.shape[2] != w.shape[3]:
                raise FormatError("%s: grouped / dilated / non-square convolutions are outside the path" % nme)
            l = dict(kind="conv", name=nme, src=layer_src(nme), cin=cin, cout=cout, k=k, stride=int(a["strides"][0]), pad=int(a["padding"][0]), relu=False)
            bias = a["weight_2"].reshape(-1) if a.get("bias_term") and "weight_2" in a else None
            bn = only_reader(tail, "BatchNorm")
            if bn:
                sc = only_reader(bn, "Scale")
                if not sc:
                    raise FormatError("%s: a BatchNorm without its Scale" % bn)
                ab_, as_ = attrs[bn], attrs[sc]
                mean, var, factor = ab_["weight_1"].reshape(-1), ab_["weight_2"].reshape(-1), float(ab_["weight_3"].reshape(-1)[0])
                gamma = as_["weight_1"].reshape(-1)
                beta = as_["weight_2"].reshape(-1) if as_.get("bias_term", True) and "weight_2" in as_ else None
                raw[nme] = dict(w=w, mean=mean, var=var, gamma=gamma, beta=beta if beta is not None else np.zeros(cout, np.float32))
                params[nme] = W.fold_bn(w, bias, factor, float(ab_.get("epsilon", 1e-5)), mean, var, gamma, beta)
                absorbed |= {bn, sc}
                tail = sc
            else:
                params[nme] = (w, bias if bias is not None else np.zeros(cout, np.float32))
            rl = only_reader(tail, "ReLU")
            if rl and attrs[rl].get("alpha", 0.0) == 0.0:
                l["relu"] = True
                absorbed.add(rl)
                tail = rl
            spec.append(l)
        elif o == "Pooling":
            if a.get("global_pooling"):
                spec.append(dict(kind="gpool", name=nme, src=layer_src(nme)))
            else:
                l = dict(kind="pool", name=nme, src=layer_src(nme), win=int(a["pool_size"][0]), stride=int(a["strides"][0]), pad=int(a["padding"][0]),
                         type=0 if a.get("method", "MAX") == "MAX" else 1)
                if a.get("cmp_out_shape_floor_as_conv"):
                    l["floor"] = True
                spec.append(l)
        elif o == "Eltwise":
            if a.get("type", "Add") != "Add" or len(ins[nme]) != 2:
                raise FormatError("%s: only two-input sums are on the path" % nme)
            l = dict(kind="eltwise", name=nme, a=layer_src(nme, 0), b=layer_src(nme, 1), relu=False)
            rl = only_reader(tail, "ReLU")
            if rl:
                l["relu"] = True
                absorbed.add(rl)
                tail = rl
            # the conv the sum is fused into (Graph::Optimize's ConvEltwise, graph.cpp:423-436; workloads: `eltwise=` on that conv, which is
            # the sum's `a`): a conv without relu read by the sum alone - the LATER of the two when both inputs qualify (res2a: branch1, branch2c)
            cands = [i for i, c in enumerate(spec) if c["name"] in (l["a"], l["b"]) and c["kind"] == "conv" and not c["relu"]
                     and len(outs.get(last_of[c["name"]], [])) == 1]
            if cands:
                fused = spec[cands[-1]]
                fused["eltwise"] = nme
                if l["b"] == fused["name"]:
                    l["a"], l["b"] = l["b"], l["a"]
            spec.append(l)
        elif o == "Dense":
            w = a["weight_1"]
            cout = int(a.get("out_dim", w.shape[2]))
            cin = int(w.size // cout)
            l = dict(kind="fc", name=nme, src=layer_src(nme), cin=cin, cout=cout)
            params[nme] = (w.reshape(cout, cin), a["weight_2"].reshape(-1) if "weight_2" in a else np.zeros(cout, np.float32))
            rl = only_reader(tail, "ReLU")
            if rl:
                l["relu"] = True
                absorbed.add(rl)
                tail = rl
            spec.append(l)
        elif o == "Softmax":
            spec.append(dict(kind="softmax", name=nme, src=layer_src(nme)))
        elif o == "ReLU":
            raise FormatError("%s: a ReLU that does not follow a conv / eltwise / fc" % nme)
        else:
            raise FormatError("%s: operator %s is outside the path (SURVEY section 8)" % (nme, o))
        last_of[nme] = tail
        for t in {nme, tail} | ({bn, sc} if o == "Convolution" and bn else set()):
            alias[t] = nme
        precs[nme] = "int8" if nodes[nme].get("bit_type", 0) == INT8 else "fp32"
        for top in outs.get(tail, []):
            s = in_scale.get((tail, top))
            if s:
                scales[nme] = float(s[0])
    for top in outs.get(input_name, []):
        s = in_scale.get((input_name, top))
        if s:
            scales["data"] = float(s[0])
    # an fc behind a spatial tensor flattens NCHW (VGG16's fc6): the workloads executor is NHWC and needs the (C, H, W) of its input
    shp = {"data": (int(input_shape[1]), int(input_shape[2]))}
    for l in spec:
        if l["kind"] == "conv":
            c, h = shp[l["src"]]
            shp[l["name"]] = (l["cout"], (h + 2 * l["pad"] - l["k"]) // l["stride"] + 1)
        elif l["kind"] == "pool":
            c, h = shp[l["src"]]
            rnd = np.floor if l.get("floor") else np.ceil
            shp[l["name"]] = (c, int(rnd((h + 2 * l["pad"] - l["win"]) / l["stride"])) + 1)
        elif l["kind"] == "gpool":
            shp[l["name"]] = (shp[l["src"]][0], 1)
        elif l["kind"] == "eltwise":
            shp[l["name"]] = shp[l["a"]]
        elif l["kind"] == "fc":
            c, h = shp[l["src"]]
            if h > 1:
                l["flatten_chw"] = (c, h, h)
            shp[l["name"]] = (l["cout"], 1)
        else:
            shp[l["name"]] = shp[l["src"]]
    return dict(name=g.get("name", "net"), spec=spec, params=params, raw=raw, input_shape=[int(x) for x in input_shape], scales=scales, precisions=precs)
