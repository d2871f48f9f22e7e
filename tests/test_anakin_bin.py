"""The `.anakin.bin` model format without a protobuf library (round-5 verdict, missing item 5; SURVEY section 8 row f-4: model I/O).

Checker: the OFFICIAL protobuf runtime (python google.protobuf) with the reference's schema built as descriptors (tests/anakin_proto.py) -
an implementation of the wire format this repository did not write. Held against it, both directions:
  * anakin_amd/anakin_bin.py (the Python reader / writer the ctypes route and bench.py --model-file use),
  * integration/mi355x/framework/anakin_bin_model.h (the C++ reader / writer behind Graph::load / Graph::save on the MI355X target's build),
    compiled alone with g++ (tests/cpp_host/anakin_bin_tool.cpp);
and, with the integration binaries present, the reference's own framework: a Python-written ResNet50 `.anakin.bin` through Graph::load ->
Optimize -> Net<MI355X>::init on the mock HIP runtime gives the op list and plan of the text model, and Graph::save writes a file the
official runtime reads back as the same graph."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from anakin_amd import anakin_bin as AB          # noqa: E402
from anakin_amd import workloads as W            # noqa: E402

pb = pytest.importorskip("google.protobuf")
import anakin_proto as AP                         # noqa: E402

BIN = os.path.join(ROOT, "integration", "_build", "test_net_mi355x.bin")
MOCK = os.path.join(ROOT, "integration", "_build", "libmock_hip.so")
needs_integration = pytest.mark.skipif(not (os.path.exists(BIN) and os.path.exists(MOCK)), reason="integration/_build not built (needs /root/reference at build time)")


@pytest.fixture(scope="module")
def M():
    return AP.build()


@pytest.fixture(scope="module")
def tool(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("abtool") / "anakin_bin_tool")
    subprocess.check_call(["g++", "-std=c++11", "-O1", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "integration", "mi355x", "framework"),
                           os.path.join(ROOT, "tests", "cpp_host", "anakin_bin_tool.cpp"), "-o", out])
    return out


def every_field_graph(M):
    """a GraphProto touching every field and value kind of the four schemas, including the awkward ones: a oneof member at its default, negative
    int32s (10-byte varints), empty strings inside repeated fields, a list of int lists, int8 and shared weight tensors, per-target scales"""
    g = M["GraphProto"]()
    g.name = "every_field"
    g.ins.append("data"); g.outs.extend(["prob_out", ""])
    g.version.major, g.version.minor, g.version.patch, g.version.version = 0, 1, 2, 1 << 40
    g.summary.temp_mem_used, g.summary.original_temp_mem_used, g.summary.system_mem_used, g.summary.model_mem_used = 4, 22, 0, 102
    g.summary.is_optimized = True
    n = g.nodes.add()
    n.name, n.lane, n.need_wait, n.bit_type = "conv1", 3, True, AP.INT8
    n.ins.append("data"); n.outs.extend(["pool1", "side"])
    n.Op.name, n.Op.is_commutative, n.Op.in_num, n.Op.out_num, n.Op.description = "Convolution", True, 1, 2, "conv"
    a = n.attr
    a["group"].i = 0; a["group"].type = AP.INT32                      # oneof member at its default: still on the wire
    a["neg"].i = -7; a["neg"].type = AP.INT32
    a["alpha"].f = 0.0; a["alpha"].type = AP.FLOAT
    a["eps"].f = 1e-5; a["eps"].type = AP.FLOAT
    a["flag"].b = False; a["flag"].type = AP.BOOLEN
    a["method"].s = b"MAX"; a["method"].type = AP.STR
    a["empty"].s = b""; a["empty"].type = AP.STR
    a["pads"].cache_list.i.extend([3, -1, 0]); a["pads"].cache_list.type = AP.INT32; a["pads"].cache_list.size = 3; a["pads"].type = AP.CACHE_LIST
    a["coeff"].cache_list.f.extend([1.0, -0.5]); a["coeff"].cache_list.type = AP.FLOAT; a["coeff"].cache_list.size = 2; a["coeff"].type = AP.CACHE_LIST
    a["bools"].cache_list.b.extend([True, False, True]); a["bools"].cache_list.type = AP.BOOLEN; a["bools"].cache_list.size = 3; a["bools"].type = AP.CACHE_LIST
    a["names"].cache_list.s.extend([b"a", b"", b"c"]); a["names"].cache_list.type = AP.STR; a["names"].cache_list.size = 3; a["names"].type = AP.CACHE_LIST
    a["nolist"].cache_list.type = AP.FLOAT; a["nolist"].type = AP.CACHE_LIST
    ll = a["ll"]; ll.type = AP.CACHE_LIST; ll.cache_list.type = AP.CACHE_LIST; ll.cache_list.size = 2
    for vals in ([1, 2, 3], [-4]):
        c = ll.cache_list.l.add(); c.i.extend(vals); c.type = AP.INT32; c.size = len(vals)
    w = a["weight_1"]; w.type = AP.TENSOR
    w.tensor.shape.dim.value.extend([2, 3, 1, 1]); w.tensor.shape.dim.size = 4
    w.tensor.valid_shape.dim.value.extend([2, 2, 1, 1]); w.tensor.valid_shape.dim.size = 4
    w.tensor.data.f.extend(np.arange(6, dtype=np.float32) - 2.5); w.tensor.data.type = AP.FLOAT; w.tensor.data.size = 6
    w.tensor.scale.f.extend([0.5, 0.25]); w.tensor.scale.type = AP.FLOAT; w.tensor.scale.size = 2
    q = a["weight_q"]; q.type = AP.TENSOR
    q.tensor.shape.dim.value.extend([1, 1, 2, 2]); q.tensor.shape.dim.size = 4
    q.tensor.data.c = bytes([1, 255, 128, 0]); q.tensor.data.type = AP.INT8; q.tensor.data.size = 4
    n2 = g.nodes.add()
    n2.name, n2.bit_type = "conv1_twin", AP.FLOAT
    n2.Op.name = "Convolution"
    s = n2.attr["weight_1"]; s.type = AP.TENSOR; s.tensor.shared = True; s.tensor.share_from = b"conv1"
    t = g.edges_in["conv1"].target.add(); t.node = "data"; t.scale.extend([0.0078125]); t.layout = 9
    g.edges_in["old_style"].val.extend(["x", "y"])
    t = g.edges_out["conv1"].target.add(); t.node = "pool1"
    t = g.edges_out["conv1"].target.add(); t.node = "side"; t.scale.extend([1.0, 2.0])
    e = g.edges_info["data_conv1"]; e.name = b"data_conv1"; e.shared = True; e.share_from = b"other_edge"
    g.edges_info["conv1_pool1"].name = b"conv1_pool1"
    return g


def as_dict(msg):
    """an official-runtime message as the dict anakin_bin.read_graph would return (absent = at default)"""
    out = {}
    for fd, v in msg.ListFields():
        if fd.message_type is not None and fd.message_type.GetOptions().map_entry:
            vt = fd.message_type.fields_by_name["value"]
            out[fd.name] = {k: as_dict(x) if vt.message_type is not None else x for k, x in v.items()}
        elif fd.is_repeated if hasattr(fd, "is_repeated") else fd.label == fd.LABEL_REPEATED:
            if fd.message_type is not None:
                out[fd.name] = [as_dict(x) for x in v]
            elif fd.type == fd.TYPE_FLOAT:
                out[fd.name] = np.asarray(list(v), np.float32)
            else:
                out[fd.name] = list(v)
        elif fd.message_type is not None:
            out[fd.name] = as_dict(v)
        else:
            out[fd.name] = v
    return out


def same(a, b, path=""):
    if isinstance(a, dict) or isinstance(b, dict):
        assert isinstance(a, dict) and isinstance(b, dict), path
        # a oneof member at its default is a key with a default value on one side and (being "unset" for ListFields only when really unset) present on the other
        assert set(a) == set(b), (path, sorted(set(a) ^ set(b)))
        for k in a:
            same(a[k], b[k], path + "/" + str(k))
    elif isinstance(a, np.ndarray) or isinstance(b, np.ndarray):
        assert np.array_equal(np.asarray(a, np.float32), np.asarray(b, np.float32)), path
    elif isinstance(a, list):
        assert len(a) == len(b), path
        for i, (x, y) in enumerate(zip(a, b)):
            same(x, y, "%s[%d]" % (path, i))
    elif isinstance(a, float) or isinstance(b, float):
        assert np.float32(a) == np.float32(b), (path, a, b)
    else:
        assert a == b, (path, a, b)


def test_python_reader_and_writer_against_the_official_runtime(M):
    g = every_field_graph(M)
    wire = g.SerializeToString()
    d = AB.read_graph(wire)
    same(d, as_dict(g))
    assert d["nodes"][0]["attr"]["neg"]["i"] == -7 and d["nodes"][0]["attr"]["pads"]["cache_list"]["i"] == [3, -1, 0]
    # our writer: the official runtime reads back the same message
    g2 = M["GraphProto"]()
    g2.ParseFromString(AB.write_graph(d))
    assert g2 == g
    # attribute values as python objects (model_io.cpp:30-300)
    at = {k: AB.attr_value(v) for k, v in d["nodes"][0]["attr"].items()}
    assert at["group"] == 0 and at["neg"] == -7 and at["alpha"] == 0.0 and at["flag"] is False and at["method"] == "MAX" and at["empty"] == ""
    assert at["pads"] == [3, -1, 0] and at["coeff"] == [1.0, -0.5] and at["bools"] == [True, False, True] and at["names"] == ["a", "", "c"]
    assert at["ll"] == [[1, 2, 3], [-4]] and at["nolist"] == []
    assert at["weight_1"].shape == (2, 3, 1, 1) and at["weight_1"][1, 2, 0, 0] == 2.5 and at["weight_q"].dtype == np.int8 and at["weight_q"].ravel().tolist() == [1, -1, -128, 0]
    assert AB.attr_value(d["nodes"][1]["attr"]["weight_1"]) == ("shared", "conv1")


def _unpacked_variant(M):
    """the same graph with repeated scalars written ONE BY ONE (legal for a proto3 reader to receive) and an unknown field in the middle"""
    out = bytearray()
    AB._put_len(out, 1, b"unpacked")
    node = bytearray()
    AB._put_len(node, 1, b"n")
    val = bytearray()
    cache = bytearray()
    for x in (5, -2):                                       # CacheDate.i = 2, one varint per element
        AB._put_varint(cache, (2 << 3) | 0); AB._put_varint(cache, x)
    for x in (1.5, -0.25):                                  # CacheDate.f = 3, one fixed32 per element
        AB._put_varint(cache, (3 << 3) | 5); cache += np.float32(x).tobytes()
    AB._put_varint(cache, (6 << 3) | 0); AB._put_varint(cache, AP.INT32)
    AB._put_varint(cache, (7 << 3) | 0); AB._put_varint(cache, 2)
    AB._put_len(val, 8, cache)
    AB._put_varint(val, (99 << 3) | 1); val += b"\1\2\3\4\5\6\7\10"      # unknown 8-byte field
    AB._put_varint(val, (14 << 3) | 0); AB._put_varint(val, AP.CACHE_LIST)
    entry = bytearray()
    AB._put_len(entry, 1, b"k"); AB._put_len(entry, 2, val)
    AB._put_len(node, 10, entry)
    AB._put_len(node, 77, b"unknown length-delimited field")
    AB._put_len(out, 2, node)
    return bytes(out)


def test_readers_accept_unpacked_scalars_and_skip_unknown_fields(M, tool, tmp_path):
    wire = _unpacked_variant(M)
    g = M["GraphProto"]()
    g.ParseFromString(wire)
    assert list(g.nodes[0].attr["k"].cache_list.i) == [5, -2] and list(g.nodes[0].attr["k"].cache_list.f) == [1.5, -0.25]
    d = AB.read_graph(wire)
    assert AB.attr_value(d["nodes"][0]["attr"]["k"]) == [5, -2]
    assert np.array_equal(d["nodes"][0]["attr"]["k"]["cache_list"]["f"], np.float32([1.5, -0.25]))
    src, dst = str(tmp_path / "u.bin"), str(tmp_path / "u2.bin")
    open(src, "wb").write(wire)
    assert subprocess.run([tool, "reencode", src, dst]).returncode == 0
    g2 = M["GraphProto"]()
    g2.ParseFromString(open(dst, "rb").read())
    g.DiscardUnknownFields()                                # (the official runtime carries them along; both readers here drop them)
    assert g2 == g


def test_cpp_reader_and_writer_against_the_official_runtime(M, tool, tmp_path):
    g = every_field_graph(M)
    src, dst = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    open(src, "wb").write(g.SerializeToString())
    assert subprocess.run([tool, "reencode", src, dst]).returncode == 0
    g2 = M["GraphProto"]()
    g2.ParseFromString(open(dst, "rb").read())
    assert g2 == g
    # malformed inputs: every truncation of the file either decodes (a prefix that ends at a field boundary is a valid shorter message) or
    # is refused - never a crash; a length running past the end is refused by both readers
    wire = g.SerializeToString()
    refused = 0
    for cut in list(range(1, 60)) + [len(wire) // 2, len(wire) - 1]:
        open(src, "wb").write(wire[:cut])
        rc = subprocess.run([tool, "reencode", src, dst]).returncode
        assert rc in (0, 3), (cut, rc)
        try:
            AB.read_graph(wire[:cut])
            py_ok = True
        except AB.FormatError:
            py_ok = False
        assert py_ok == (rc == 0), (cut, rc, py_ok)
        refused += rc == 3
    assert refused > 30
    bad = bytearray()
    AB._put_varint(bad, (2 << 3) | 2); AB._put_varint(bad, 1000); bad += b"short"
    open(src, "wb").write(bytes(bad))
    assert subprocess.run([tool, "reencode", src, dst]).returncode == 3
    with pytest.raises(AB.FormatError):
        AB.read_graph(bytes(bad))


@pytest.mark.parametrize("name", ["resnet50", "vgg16_head"])
def test_workloads_models_round_trip_through_the_file(M, tool, tmp_path, name):
    """write_model -> (official runtime parses it) -> load_model: the same layer list, bit-identical folded weights, the raw BatchNorm blobs;
    the C++ reader sees the same node / attribute / weight counts. vgg16_head: VGG16 with its 411 MB fc6 dropped to keep the test small -
    conv + bias (no BatchNorm), max pooling, fc + relu, the NCHW flatten in front of the first fc."""
    if name == "resnet50":
        model = W.build_model("resnet50")
    else:
        model = W.build_model("vgg16")
        spec = [dict(l) for l in model["spec"] if l["name"] not in ("fc6", "fc7")]
        fc8 = next(l for l in spec if l["name"] == "fc8")
        fc8.update(src="pool5", cin=512 * 7 * 7, flatten_chw=(512, 7, 7))
        rng = np.random.default_rng(3)
        params = dict(model["params"], fc8=((rng.standard_normal((1000, 512 * 49)) * 0.01).astype(np.float32), model["params"]["fc8"][1]))
        model = dict(model, spec=spec, params={k: v for k, v in params.items() if k not in ("fc6", "fc7")})
    path = str(tmp_path / "m.anakin.bin")
    wire = AB.write_model(model, path, batch=2)
    g = M["GraphProto"]()
    g.ParseFromString(wire)                                 # a well-formed GraphProto by the official runtime's judgement
    ops = [n.Op.name for n in g.nodes]
    assert ops.count("Input") == 1 and ops.count("Output") == 1 and ops.count("Convolution") == sum(l["kind"] == "conv" for l in model["spec"])
    m2 = AB.load_model(path)
    assert m2["input_shape"] == [2, 3, 224, 224]
    assert [dict(l) for l in m2["spec"]] == [dict(l) for l in model["spec"]]
    for k, (w, b) in model["params"].items():
        assert np.array_equal(m2["params"][k][0], w) and np.array_equal(m2["params"][k][1], b), k
    for k, r in model.get("raw", {}).items():
        assert all(np.array_equal(m2["raw"][k][f], r[f]) for f in ("w", "mean", "var", "gamma", "beta")), k
    r = subprocess.run([tool, "summary", path], capture_output=True, text=True)
    assert r.returncode == 0
    f = r.stdout.split()
    val = lambda k: int(f[f.index(k) + 1])      # noqa: E731
    assert val("nodes") == len(g.nodes) and val("attrs") == sum(len(n.attr) for n in g.nodes)
    assert val("weight_floats") == sum(len(v.tensor.data.f) for n in g.nodes for v in n.attr.values())
    assert val("in_edges") == sum(len(l.target) for l in g.edges_in.values())


def test_int8_scales_and_precisions_travel_in_the_file(tmp_path):
    model = W.build_model("resnet50")
    scales = W.calibrate(model, W.make_input(2))
    wire = AB.write_model(model, None, batch=1, precision="int8", scales=scales)
    m2 = AB.load_model(wire)
    f9 = lambda x: float(np.float32(float("%.9g" % x)))      # noqa: E731
    assert set(m2["scales"]) == set(scales)
    assert all(m2["scales"][k] == f9(v) for k, v in scales.items())
    assert m2["precisions"]["prob"] == "fp32" and m2["precisions"]["conv1"] == "int8" and m2["precisions"]["res5c"] == "int8"


def test_operators_outside_the_path_are_refused():
    model = W.build_model("resnet50")
    wire = AB.write_model(model, None)
    g = AB.read_graph(wire)
    g["nodes"][5]["Op"]["name"] = "Deconvolution"
    with pytest.raises(AB.FormatError, match="outside the path"):
        AB.load_model(AB.write_graph(g))
    # well-formed GraphProtos that are not usable networks: a conv without its weights, an edge from a node that does not exist
    g = AB.read_graph(wire)
    conv = next(n for n in g["nodes"] if n["Op"]["name"] == "Convolution")
    del conv["attr"]["weight_1"]
    with pytest.raises(AB.FormatError, match="weight_1"):
        AB.load_model(AB.write_graph(g))
    g = AB.read_graph(wire)
    g["edges_in"]["conv1"]["target"][0]["node"] = "nowhere"
    with pytest.raises(AB.FormatError):
        AB.load_model(AB.write_graph(g))


# ------------------------------------------------------------------------------------------------ the reference's framework on the mock runtime
def _dry(model_args, d, env_extra=None):
    env = dict(os.environ, LD_PRELOAD=MOCK, SABER_MI355X_NET_PLAN_TUNE="0")
    env.update(env_extra or {})
    return subprocess.run([BIN] + model_args, env=env, capture_output=True, text=True, errors="replace", cwd=d, timeout=600)


def _oplist(d):
    """oplist.txt with the buffer addresses blanked and every operator's OUT edges sorted (the order of a Split's readers is the iteration
    order of an unordered_map inside Graph::Freeze on the text route, graph.cpp:203-214, and the file's order on the other: no operator reads it)"""
    import re
    blocks = []
    for ln in re.sub(r"ptr 0x[0-9a-f]+", "ptr", open(os.path.join(d, "oplist.txt")).read()).split("\n"):
        if ln.startswith("op "):
            blocks.append([ln, [], []])
        elif ln.startswith("  in "):
            blocks[-1][1].append(ln)
        elif ln.startswith("  out "):
            blocks[-1][2].append(ln)
    return [(b[0], b[1], sorted(b[2])) for b in blocks]


@needs_integration
@pytest.mark.parametrize("precision,route", [("int8", "scales_in_file"), ("int8", "calibrator_files"), ("fp32", "-")])
def test_reference_graph_load_reads_a_python_written_anakin_bin(M, tmp_path, precision, route):
    """Graph<MI355X>::load(model.anakin.bin) -> Optimize -> Net::init (dry, mock runtime): op for op, edge for edge (dtype, layout, shape,
    scale, sharing) and plan line for plan line what the TEXT model of the same network gives; Graph::save of the loaded text model is read
    back by the official runtime as the graph the Python writer wrote (nodes, operators, attributes with their weights, edges with scales)."""
    from integration import net_model as NM
    model = W.build_model("resnet50")
    x = W.make_input(1)
    scales = W.calibrate(model, x) if precision == "int8" else {}
    dt, db = str(tmp_path / "text"), str(tmp_path / "bin")
    os.makedirs(dt); os.makedirs(db)
    cal = route == "calibrator_files"
    mt, wb = NM.write_model(model, scales, 1, dt, precision, calibrator_config=cal)
    x.tofile(os.path.join(dt, "input.bin"))
    r = _dry([mt, wb, os.path.join(dt, "input.bin"), dt, "dry"], dt)
    assert r.returncode == 0, r.stderr[-2000:]
    path = os.path.join(db, "model.anakin.bin")
    AB.write_model(model, path, batch=1, precision=precision, scales=scales, calibration_in_file=not cal)
    extra = {"SABER_TEST_PRECISION": precision}
    if cal:
        extra["SABER_TEST_CALIBRATOR"] = "%s %s" % (os.path.join(dt, "net_config.txt"), os.path.join(dt, "calibrator.txt"))
    r = _dry([path, "-", os.path.join(dt, "input.bin"), db, "dry"], db, extra)
    assert r.returncode == 0, r.stderr[-2000:]
    assert _oplist(db) == _oplist(dt)
    assert open(os.path.join(db, "plan.txt")).read() == open(os.path.join(dt, "plan.txt")).read()
    if route == "calibrator_files":
        return
    # Graph::save (C++ writer) of the text-loaded graph against the Python-written file, both read by the official runtime
    saved = os.path.join(dt, "saved.anakin.bin")
    r = _dry([mt, wb, os.path.join(dt, "input.bin"), dt, "savebin", saved], dt)
    assert r.returncode == 0, r.stderr[-2000:]
    a, b = M["GraphProto"](), M["GraphProto"]()
    a.ParseFromString(open(saved, "rb").read())
    b.ParseFromString(open(path, "rb").read())
    na, nb = {n.name: n for n in a.nodes}, {n.name: n for n in b.nodes}
    assert set(na) == set(nb) and list(a.ins) == list(b.ins) and sorted(a.outs) == sorted(b.outs)
    for k in na:
        assert na[k].Op.name == nb[k].Op.name and na[k].bit_type == nb[k].bit_type, k
        assert set(na[k].attr) == set(nb[k].attr), (k, set(na[k].attr) ^ set(nb[k].attr))
        for key in na[k].attr:
            va, vb = na[k].attr[key], nb[k].attr[key]
            assert va.type == vb.type, (k, key)
            if va.type == AP.TENSOR:
                assert list(va.tensor.shape.dim.value) == list(vb.tensor.shape.dim.value), (k, key)
                assert np.array_equal(np.asarray(va.tensor.data.f, np.float32), np.asarray(vb.tensor.data.f, np.float32)), (k, key)
            else:
                assert va == vb, (k, key)
    for ea, eb in ((a.edges_in, b.edges_in), (a.edges_out, b.edges_out)):
        assert set(ea) == set(eb)
        for k in ea:
            ta = sorted((t.node, tuple(t.scale), t.layout) for t in ea[k].target)
            tb = sorted((t.node, tuple(t.scale), t.layout) for t in eb[k].target)
            assert ta == tb, (k, ta, tb)
    # ... and the file the reference wrote loads again into the same op list
    dr = str(tmp_path / "reload")
    os.makedirs(dr)
    r = _dry([saved, "-", os.path.join(dt, "input.bin"), dr, "dry"], dr)
    assert r.returncode == 0, r.stderr[-2000:]
    assert _oplist(dr) == _oplist(dt)


@needs_integration
def test_reference_graph_load_and_save_carry_every_attribute_kind(M, tmp_path):
    """Graph<MI355X>::load of a crafted `.anakin.bin` whose nodes carry every attribute kind NodeIO reads (model_io.cpp:30-300: string, int, float,
    bool, the four list types, a list of int lists, a weight tensor with a valid shape smaller than its real shape and a per-channel scale, a
    weight SHARED from another node) -> Graph::save -> the official runtime reads back the same attributes, node flags (lane, need_wait,
    bit_type), edge scales / layouts and the optimisation summary. (load only builds the graph: no operator is instantiated, so the attribute
    names are free.) An int8 weight payload loads, and Graph::save refuses it by name - the reference writes float blocks only."""
    g = M["GraphProto"]()
    g.name = "attrs"
    g.ins.append("in0"); g.outs.append("out0")
    g.summary.is_optimized = True
    g.summary.temp_mem_used, g.summary.original_temp_mem_used, g.summary.system_mem_used, g.summary.model_mem_used = 3, 9, 1, 77

    def node(name, op, bit=AP.FLOAT, lane=0, wait=False):
        n = g.nodes.add()
        n.name, n.bit_type, n.lane, n.need_wait = name, bit, lane, wait
        n.Op.name = op
        return n
    node("in0", "Input").attr["input_shape"].CopyFrom(_pb_list(M, "i", [1, 3, 8, 8], AP.INT32))
    a = node("a", "Convolution", AP.INT8, lane=2, wait=True)
    at = a.attr
    at["s"].s = b"text"; at["s"].type = AP.STR
    at["i"].i = -5; at["i"].type = AP.INT32
    at["zero"].i = 0; at["zero"].type = AP.INT32
    at["f"].f = 0.25; at["f"].type = AP.FLOAT
    at["b"].b = True; at["b"].type = AP.BOOLEN
    at["li"].CopyFrom(_pb_list(M, "i", [4, -4, 0], AP.INT32))
    at["lf"].CopyFrom(_pb_list(M, "f", [1.5, -2.0], AP.FLOAT))
    at["lb"].CopyFrom(_pb_list(M, "b", [True, False], AP.BOOLEN))
    at["ls"].CopyFrom(_pb_list(M, "s", [b"x", b"yz"], AP.STR))
    ll = at["ll"]; ll.type = AP.CACHE_LIST; ll.cache_list.type = AP.CACHE_LIST; ll.cache_list.size = 2
    for vals in ([7, 8], [9]):
        c = ll.cache_list.l.add(); c.i.extend(vals); c.type = AP.INT32; c.size = len(vals)
    w = at["weight_1"]; w.type = AP.TENSOR
    w.tensor.shape.dim.value.extend([2, 3, 2, 2]); w.tensor.shape.dim.size = 4
    w.tensor.valid_shape.dim.value.extend([2, 3, 1, 2]); w.tensor.valid_shape.dim.size = 4
    w.tensor.data.f.extend(np.arange(24, dtype=np.float32) * 0.5 - 3.0); w.tensor.data.type = AP.FLOAT; w.tensor.data.size = 24
    w.tensor.scale.f.extend([0.5, 0.125]); w.tensor.scale.type = AP.FLOAT; w.tensor.scale.size = 2
    b = node("b", "Convolution")
    s = b.attr["weight_1"]; s.type = AP.TENSOR; s.tensor.shared = True; s.tensor.share_from = b"a"
    node("out0", "Output")
    for bot, top, scale, layout in (("in0", "a", [0.0625], 9), ("a", "b", [], 0), ("b", "out0", [1.0, 2.0], 8)):
        for m_, key, other in ((g.edges_in, top, bot), (g.edges_out, bot, top)):
            t = m_[key].target.add(); t.node = other; t.scale.extend(scale); t.layout = layout
    g.edges_info["a_b"].name = b"a_b"; g.edges_info["a_b"].shared = True; g.edges_info["a_b"].share_from = b"in0_a"
    d = str(tmp_path)
    src, dst = os.path.join(d, "attrs.anakin.bin"), os.path.join(d, "saved.anakin.bin")
    open(src, "wb").write(g.SerializeToString())
    np.zeros(4, np.float32).tofile(os.path.join(d, "input.bin"))
    r = _dry([src, "-", os.path.join(d, "input.bin"), d, "savebin", dst], d, {"SABER_TEST_PRECISION": "fp32"})
    # a string LIST loads (model_io.cpp:97-105) but cannot be written back: PTuple<std::string> has no registered type name
    # (framework/core/data_types.h), so the reference's own save does not know it either - refused with the attribute's name
    assert r.returncode == 2 and "attribute ls of a" in r.stderr, r.stderr[-1500:]
    del g.nodes[1].attr["ls"]
    open(src, "wb").write(g.SerializeToString())
    r = _dry([src, "-", os.path.join(d, "input.bin"), d, "savebin", dst], d, {"SABER_TEST_PRECISION": "fp32"})
    assert r.returncode == 0, r.stderr[-2000:]
    h = M["GraphProto"]()
    h.ParseFromString(open(dst, "rb").read())
    assert h.name == "attrs" and list(h.ins) == ["in0"] and list(h.outs) == ["out0"]
    assert h.summary == g.summary
    ha, ga = {n.name: n for n in h.nodes}, {n.name: n for n in g.nodes}
    assert set(ha) == set(ga)
    na = ha["a"]
    assert (na.lane, na.need_wait, na.bit_type, na.Op.name) == (2, True, AP.INT8, "Convolution")
    for key in ("s", "i", "zero", "f", "b", "li", "lf", "lb", "ll"):
        assert na.attr[key] == ga["a"].attr[key], key
    tw = na.attr["weight_1"].tensor
    assert list(tw.shape.dim.value) == [2, 3, 2, 2] and list(tw.valid_shape.dim.value) == [2, 3, 1, 2]
    assert np.array_equal(np.asarray(tw.data.f, np.float32), np.arange(24, dtype=np.float32) * 0.5 - 3.0) and list(tw.scale.f) == [0.5, 0.125]
    tb = ha["b"].attr["weight_1"].tensor
    assert tb.shared and tb.share_from == b"a" and len(tb.data.f) == 0
    tg = {(k, t.node): (list(t.scale), t.layout) for k, l in h.edges_in.items() for t in l.target}
    # layouts: 0 loads as NCHW (parser.cpp:170-172) - and so does every other value: graph::Edge's copy constructor (framework/graph/node.h:190-196)
    # copies the scale, lane and sharing of an edge but not its layout, so the value set on the parser's temporary edge never reaches the graph's
    # arc, in the reference's own loader as here (an edge's layout is decided later, from the calibrator config: net.cpp / calibrator_parse.cpp)
    assert tg == {("a", "in0"): ([0.0625], 8), ("b", "a"): ([], 8), ("out0", "b"): ([1.0, 2.0], 8)}
    assert h.edges_info["a_b"].shared and h.edges_info["a_b"].share_from == b"in0_a"
    # an int8 payload: loads (model_io.cpp:216-259), and the save names it
    q = g.nodes[1].attr["weight_q"]; q.type = AP.TENSOR
    q.tensor.shape.dim.value.extend([1, 1, 2, 2]); q.tensor.shape.dim.size = 4
    q.tensor.data.c = bytes([1, 255, 128, 0]); q.tensor.data.type = AP.INT8; q.tensor.data.size = 4
    open(src, "wb").write(g.SerializeToString())
    r = _dry([src, "-", os.path.join(d, "input.bin"), d, "savebin", dst], d, {"SABER_TEST_PRECISION": "fp32"})
    assert r.returncode == 2 and "weight_q" in r.stderr and "only float blocks" in r.stderr, r.stderr[-1500:]
    # an edge to a node the file does not hold is refused by name (the reference would abort in a CHECK inside GraphBase)
    t = g.edges_in["b"].target.add(); t.node = "ghost"
    open(src, "wb").write(g.SerializeToString())
    r = _dry([src, "-", os.path.join(d, "input.bin"), d, "savebin", dst], d, {"SABER_TEST_PRECISION": "fp32"})
    assert r.returncode == 2 and "unknown node ghost" in r.stderr, r.stderr[-1500:]
    del g.edges_in["b"].target[-1]
    # a weight whose payload is shorter than its shape is refused at load with the node's name
    del g.nodes[1].attr["weight_q"]
    g.nodes[1].attr["weight_1"].tensor.data.size = 20
    open(src, "wb").write(g.SerializeToString())
    r = _dry([src, "-", os.path.join(d, "input.bin"), d, "savebin", dst], d, {"SABER_TEST_PRECISION": "fp32"})
    assert r.returncode == 2 and "weight_1 of a" in r.stderr, r.stderr[-1500:]


def _pb_list(M, field, vals, ltype):
    v = M["valueType"]()
    v.type = AP.CACHE_LIST
    getattr(v.cache_list, field).extend(vals)
    v.cache_list.type = ltype
    v.cache_list.size = len(vals)
    return v


def test_cpp_reader_survives_corrupted_files_under_the_sanitizers(M, tmp_path):
    """The wire-format reader reads an untrusted file: 300 random corruptions of a GraphProto that touches every field (byte flips, inserted
    and deleted bytes, truncations, lengths blown up to 2^31 .. 2^63) through the reader + writer built with AddressSanitizer and
    UndefinedBehaviorSanitizer (CPU build: the sanitizers are not available on the GPU pool) - every run ends with "decoded" (0) or "refused"
    (3), never a sanitizer report, a crash or a hang; the Python reader agrees with it on which files are well-formed."""
    tool = str(tmp_path / "anakin_bin_tool_asan")
    subprocess.check_call(["g++", "-std=c++11", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-I" + os.path.join(ROOT, "integration", "mi355x", "framework"), os.path.join(ROOT, "tests", "cpp_host", "anakin_bin_tool.cpp"), "-o", tool])
    wire = every_field_graph(M).SerializeToString()
    rng = np.random.default_rng(2026)
    src, dst = str(tmp_path / "c.bin"), str(tmp_path / "c2.bin")
    outcomes = {0: 0, 3: 0}
    for it in range(300):
        b = bytearray(wire)
        kind = it % 5
        if kind == 0:
            for _ in range(int(rng.integers(1, 4))):
                b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
        elif kind == 1:
            p = int(rng.integers(0, len(b)))
            b[p:p] = bytes(rng.integers(0, 256, int(rng.integers(1, 9))).astype(np.uint8))
        elif kind == 2:
            p = int(rng.integers(0, len(b) - 8))
            del b[p:p + int(rng.integers(1, 8))]
        elif kind == 3:
            b = b[:int(rng.integers(0, len(b)))]
        else:                                   # a huge varint where a length or a value stood
            p = int(rng.integers(0, len(b)))
            b[p:p + 1] = bytes([0xFF] * int(rng.integers(4, 10)) + [0x7F])
        open(src, "wb").write(bytes(b))
        r = subprocess.run([tool, "reencode", src, dst], capture_output=True, text=True, timeout=60)
        assert r.returncode in (0, 3), (it, kind, r.returncode, r.stderr[-1500:])
        outcomes[r.returncode] += 1
        try:
            AB.read_graph(bytes(b))
            py_ok = True
        except AB.FormatError:
            py_ok = False
        assert py_ok == (r.returncode == 0), (it, kind, r.returncode, py_ok)
    assert outcomes[0] > 20 and outcomes[3] > 100, outcomes
