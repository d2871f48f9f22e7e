"""The op list of `anakin_amd.workloads.framework_spec` IS what the reference's own optimiser emits.

`integration/_build/test_net_mi355x.bin` is the reference's framework — Graph / fusion pass / stride-up / schedulers /
memory planner / Net, compiled unmodified against the MI355X target (integration/apply_mi355x_target.py) — driven with the
network as ORIGINAL operators (Convolution + BatchNorm + Scale + ReLU ...). Here it runs `dry` on the malloc-backed mock HIP
runtime (integration/mock_hip): Graph::Optimize() and Net<MI355X>::init() execute for real (they are host code), nothing
is computed. The dumped op list — operator types after fusion, producers, and every edge's dtype / layout / shape / scale —
must equal the Python list op for op; the GPU tier (tests/test_gpu_net.py) then checks the bytes.

The binaries are built by integration/build_mi355x_test.sh where /root/reference exists and travel with the repo."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from anakin_amd import workloads as W          # noqa: E402
from integration import net_model as NM        # noqa: E402

BIN = os.path.join(ROOT, "integration", "_build", "test_net_mi355x.bin")
MOCK = os.path.join(ROOT, "integration", "_build", "libmock_hip.so")
pytestmark = pytest.mark.skipif(not (os.path.exists(BIN) and os.path.exists(MOCK)),
                                reason="integration/_build not built (needs /root/reference at build time)")

DT = {W.F32: "f32", W.S8: "s8", W.U8: "u8"}


def dry_run(tmp_path, name, precision, batch=1, calibrator_config=False, rename=None, expect_fail=False):
    model = W.build_model(name)
    x = W.make_input(batch)
    scales = W.calibrate(model, x) if precision == "int8" else {}
    d = str(tmp_path)
    mt, wb = NM.write_model(model, scales, batch, d, precision, calibrator_config=calibrator_config, rename=rename)
    x.tofile(os.path.join(d, "input.bin"))
    env = dict(os.environ, LD_PRELOAD=MOCK, SABER_MI355X_NET_PLAN_TUNE="0")      # (no timing on the mock runtime)
    r = subprocess.run([BIN, mt, wb, os.path.join(d, "input.bin"), d, "dry"], env=env, capture_output=True, text=True, errors="replace",
                       cwd=d)          # (the reference's logger writes ./log/ next to the working directory)
    if expect_fail:
        return r
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    dry_run.plan = open(os.path.join(d, "plan.txt")).read().split("\n")
    return model, scales, NM.parse_oplist(os.path.join(d, "oplist.txt"))


def resolve_producers(ops):
    """name of the COMPUTE op (or Input) behind every input of every op, looking through Split / Gather nodes."""
    by = {o["name"]: o for o in ops}

    def real(node):
        o = by[node]
        return real(o["ins"][0]["edge_bottom"]) if o["type"] == "Split" else node
    for o in ops:
        for e in o["ins"] + o["outs"]:
            # edge name = <bottom>_<top>; node names contain '_' themselves, so match against the known nodes
            cands = [n for n in by if e["edge"].startswith(n + "_") and e["edge"][len(n) + 1:] in by]
            assert cands, e["edge"]
            e["edge_bottom"] = max(cands, key=len)
    return {o["name"]: [real(e["edge_bottom"]) for e in o["ins"]] for o in ops}


def expected_type(l, raw):
    if l["kind"] == "conv":
        bn = l["name"] in raw
        return {(True, True): "ConvBatchnormScaleRelu", (True, False): "ConvBatchnormScale", (False, True): "ConvRelu",
                (False, False): "Convolution"}[(bn, bool(l["relu"]))]
    return {"pool": "Pooling", "gpool": "Pooling", "eltwise": "EltwiseRelu", "fc": "Dense", "softmax": "Softmax"}[l["kind"]]


def _check_int8_list(tmp_path, name, route="set_calls", rename=None):
    model, scales, ops = dry_run(tmp_path, name, "int8", calibrator_config=route == "calibrator_files", rename=rename)
    R = rename or (lambda n: n)
    prod = resolve_producers(ops)
    compute = [o for o in ops if o["type"] not in ("Input", "Output", "Split")]
    spec = W.framework_spec(model["spec"], "int8")
    assert len(compute) == len(spec)
    by = {o["name"]: o for o in compute}
    assert sorted(by) == sorted(R(l["name"]) for l in spec)          # name for name
    # both orders are schedules of the same DAG: every op of the reference's order runs after its producers
    pos = {o["name"]: i for i, o in enumerate(ops)}
    shape, dt, sc = {"data": (224, 3)}, {"data": W.F32}, dict(scales)
    for l in spec:
        o = by[R(l["name"])]
        assert o["type"] == expected_type(l, model["raw"]), l["name"]
        srcs = [l[k] for k in ("src", "a", "b") if k in l]
        assert sorted(prod[o["name"]]) == sorted(R(s_) for s_ in srcs), (l["name"], prod[o["name"]], srcs)
        assert all(pos[R(s_)] < pos[o["name"]] for s_ in srcs)
        out = o["outs"][0]
        if l["kind"] == "conv":
            hin, _ = shape[l["src"]]
            ho = (hin + 2 * l["pad"] - l["k"]) // l["stride"] + 1
            shape[l["name"]], dt[l["name"]] = (ho, l["cout"]), l["odt"]
            assert o["prec"] == ("uint8" if l["relu"] else "int8")      # auto_config_node_dtype
        elif l["kind"] == "pool":
            hin, c = shape[l["src"]]
            rnd = np.floor if l.get("floor") else np.ceil
            shape[l["name"]], dt[l["name"]] = (int(rnd((hin + 2 * l["pad"] - l["win"]) / l["stride"])) + 1, c), dt[l["src"]]
            sc[l["name"]] = sc[l["src"]]
        elif l["kind"] == "gpool":
            assert l["int8"]
            shape[l["name"]], dt[l["name"]] = (1, shape[l["src"]][1]), dt[l["src"]]
            sc[l["name"]] = sc[l["src"]]
        elif l["kind"] == "eltwise":
            shape[l["name"]], dt[l["name"]] = shape[l["a"]], W.S8
        elif l["kind"] in ("fc", "softmax"):
            shape[l["name"]], dt[l["name"]] = (1, 1000), W.F32
        hw, c = shape[l["name"]]
        assert out["dtype"] == DT[dt[l["name"]]], (l["name"], out)
        if dt[l["name"]] == W.F32:
            assert out["layout"] == "nchw" and out["shape"] == [1, c, hw, hw], (l["name"], out)
        else:
            assert out["layout"] == "nhwc" and out["shape"] == [1, hw, hw, c], (l["name"], out)
            if l["kind"] not in ("pool", "gpool"):    # a pooling's output scale is overwritten at init (inherits its input's)
                assert abs(out["scale"] - sc[l["name"]]) <= 1e-6 * sc[l["name"]], (l["name"], out["scale"], sc[l["name"]])
        for e, s_ in zip(o["ins"], srcs if l["kind"] != "eltwise" else [l["a"], l["b"]]):
            if l["kind"] != "eltwise":
                assert e["dtype"] == DT[dt[s_]], (l["name"], e)
    return model, spec, compute


def test_resnet101_int8_op_list_is_the_reference_optimisers(tmp_path):
    """BASELINE.json's deep-stack config through the reference's OWN Graph::Optimize + Net<MI355X>::init: the 144-operator list
    workloads.framework_spec predicts (rounds 2-3 could only extrapolate it: the reference aborted at graph_base.inl:99).
    Root cause, found with AddressSanitizer: GraphBase::remove_byio (framework/graph/graph_base.inl:218-240),
    called by graph_strategy::_stride_up_like_concat (optimize_strategy.h:236), erases the arc from the arc list and then
    compares the ERASED node's names through the stale iterators of the per-vertex arc tables. Names of <= 15 characters sit in
    the freed node's small-string buffer and still compare equal; ResNet101's `res4b10_branch2a` ... (16 characters: heap
    storage) do not, wrong arcs survive, the fusion pass later trips over them. The graph is therefore built with short node
    names (net_model.short_names: `r4b22_2c`) - and, to pin the diagnosis, ResNet50 with LONG names must fail the same way."""
    model, spec, compute = _check_int8_list(tmp_path / "r101", "resnet101", rename=NM.short_names)
    assert len(spec) == 144          # 104 conv + 33 eltwise + pool1 + 3 stride-up poolings + pool5 + fc + softmax
    s2 = [l["name"] for l in spec if l["kind"] == "conv" and l["k"] == 3 and l["stride"] == 2]
    assert s2 == ["res2c_branch2b", "res3d_branch2b", "res4b22_branch2b"], s2
    head = dry_run.plan[0].split()
    assert int(head[head.index("plan") + 1]) == 1 and int(head[head.index("captured_ops") + 1]) == 144, dry_run.plan[0]
    assert int(head[head.index("launches") + 1]) <= 72
    r = dry_run(tmp_path / "r101_long", "resnet101", "int8", expect_fail=True)                 # the model's own names: 16 characters
    # (undefined behaviour: depending on the heap layout of the build it ends in the CHECK at graph_base.inl:99 or in a SIGSEGV)
    assert r.returncode != 0, "ResNet101 with its 16-character node names was expected to trip over the reference's remove_byio"
    r = dry_run(tmp_path / "r50_long", "resnet50", "int8", expect_fail=True, rename=lambda n: n if n == "data" else "a_rather_long_prefix_" + n)
    assert r.returncode != 0, "ResNet50 with node names of > 15 characters was expected to trip over the same reference bug"


@pytest.mark.parametrize("route", ["set_calls", "calibrator_files"])
def test_resnet50_int8_op_list_is_the_reference_optimisers(tmp_path, route):
    """route: how precisions and scales reach the graph - Graph::SetOpPrec / SetVarScale calls, or the two text files of a deployed
    model through Graph::load_calibrator_config (framework/graph/graph.cpp:555, parser framework/core/net/calibrator_parse.cpp)"""
    model, spec, compute = _check_int8_list(tmp_path, "resnet50", route)
    assert len(spec) == 76          # 53 conv + 16 eltwise + pool1 + 3 stride-up poolings + pool5 + fc + softmax
    # the three stride-2 3x3 convolutions and their shortcut poolings (graph_strategy::apply_stride_up)
    s2 = [l["name"] for l in spec if l["kind"] == "conv" and l["k"] == 3 and l["stride"] == 2]
    assert s2 == ["res2c_branch2b", "res3d_branch2b", "res4f_branch2b"]
    assert [l["name"] for l in spec if l.get("floor")] == ["res2b_outsplit_pool", "res3c_outsplit_pool", "res4e_outsplit_pool"]
    # the memory planner aliases edge buffers (MemoryScheduler): far fewer distinct buffers than edges
    ptrs = {e["ptr"] for o in compute for e in o["outs"]}
    assert len(ptrs) <= 8, len(ptrs)
    # The plan behind Net<MI355X>::prediction() (integration/mi355x/framework/mi355x_net_plan.h): Net::init ran its operator
    # loop once under saber_hip_capture_begin / _end - recording needs no device - and handed the list to saber_hip_net_optimize:
    # 76 captured operators (the aliasing renamed away: one tensor per written edge + the input), fused to 51 ops with 16 eltwise
    # epilogues, 4 sibling pairs, the stem + pooling, 3 absorbed stride-up poolings, pool5 inside the last conv; <= 36 launches
    head = dry_run.plan[0].split()
    val = lambda k: int(head[head.index(k) + 1])      # noqa: E731
    assert val("plan") == 1 and val("captured_ops") == 76, dry_run.plan[0]
    plan_ops = [ln.split(None, 2)[2] for ln in dry_run.plan[1:] if ln.startswith("op ")]
    assert len(plan_ops) == 51 and val("launches") <= 36, (len(plan_ops), dry_run.plan[0])
    # (the res2a pair runs inside the stem launch - saber_hip_net_optimize flag 512 - and launches nothing itself)
    assert sum(o.startswith("conv:pair_") for o in plan_ops) == 3 and plan_ops[1] == "conv:(in the stem launch)"
    assert plan_ops[0].startswith("conv:stem7x7s2_maxpool3x3s2") and plan_ops[0].endswith("+pair1x1_256+64")
    assert plan_ops[-1] == "softmax_f32" and "gpool" in plan_ops[-3]
    assert not any(o.startswith(("eltwise", "pool2d")) for o in plan_ops)
    assert int(dry_run.plan[1].split()[1]) == 77      # tensors: data + 76 written edges


def test_resnet50_fp32_and_vgg16_pass_the_reference_optimiser(tmp_path):
    _, _, ops = dry_run(tmp_path / "r50", "resnet50", "fp32")
    kinds = [o["type"] for o in ops]
    # FP32: the conv+eltwise fusion scheduler is ON (graph.cpp:423-436): 16 ConvEltwise + their Gather placeholders
    assert kinds.count("ConvEltwise") == 16 and kinds.count("Gather") == 16
    assert kinds.count("ConvBatchnormScaleRelu") == 33 and kinds.count("ConvBatchnormScale") == 4
    assert kinds.count("Pooling") == 5 and kinds.count("Dense") == 1 and kinds.count("Softmax") == 1
    assert all(e["dtype"] == "f32" and e["layout"] == "nchw" for o in ops for e in o["outs"])
    assert "plan 1 captured_ops 60 " in dry_run.plan[0], dry_run.plan[0]      # 53 conv + 5 pooling + fc + softmax behind prediction()
    _, _, ops = dry_run(tmp_path / "vgg", "vgg16", "fp32")
    kinds = [o["type"] for o in ops]
    assert kinds.count("ConvRelu") == 13 and kinds.count("Pooling") == 5 and kinds.count("Dense") == 3
    assert kinds.count("ReLU") == 2 and kinds.count("Softmax") == 1
    assert "plan 1 captured_ops 24 " in dry_run.plan[0], dry_run.plan[0]



@pytest.mark.parametrize("mode", ["worker", "worker_async", "worker_pinned"])
def test_worker_host_side_on_the_mock_runtime(tmp_path, mode):
    """Worker<MI355X, INT8> (framework/core/net/worker.h:38-60), 3 pool threads, on the malloc-backed mock HIP runtime: nothing is computed,
    but everything the HOST does per request runs - the constructor declaring the shared device to the plans (worker.cpp patch ->
    MI355XNetPlanDefaults::worker_threads: a stream per Net, SABER_HIP_NET_SHARED_DEVICE), Graph::load + Optimize + Net::init with the
    captured plan in every pool thread, the per-thread copy lanes of TargetWrapper<MI355X>::sync_memcpy (pinned staging ring, 4.8 MB in
    1 MB chunks; mi355x_impl.cpp), sync_prediction and async_prediction / async_get_result. The GPU tier (tests/test_gpu_net.py) runs the
    same binary on the device and checks the answers."""
    model = W.build_model("resnet50")
    x = W.make_input(8)
    scales = W.calibrate(model, x[:2])
    d = str(tmp_path)
    mt, wb = NM.write_model(model, scales, 8, d, "int8", calibrator_config=True)
    x.tofile(os.path.join(d, "input.bin"))
    env = dict(os.environ, LD_PRELOAD=MOCK, SABER_MI355X_NET_PLAN_TUNE="0")
    r = subprocess.run([BIN, mt, wb, os.path.join(d, "input.bin"), d, mode, "3", "24"], env=env, capture_output=True, text=True,
                       errors="replace", cwd=d, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    wt = open(os.path.join(d, "worker.txt")).read().split()
    f = {wt[i]: wt[i + 1] for i in range(0, len(wt) - 1, 2)}
    assert int(f["requests"]) == 24 and int(f["mismatches"]) == 0 and int(f["coop_fallbacks"]) == 0
    assert int(f["async"]) == (1 if "async" in mode else 0)


def test_eight_devices_on_the_mock_runtime_every_net_on_its_own_device(tmp_path):
    """Round-5 verdict item 8 (multi-GPU readiness of the C++ side; no 8-GPU node has ever been available): the mock runtime reports 8
    devices (MOCK_HIP_DEVICES=8: a current device per thread, every allocation / stream remembered with its device), 8 threads each
    TargetWrapper<MI355X>::set_device(d) and build their own Graph + Net<MI355X, INT8> + captured plan (the reference's Net takes the
    calling thread's current device, net.cpp:340) and call Gemm once. Asserted by the driver (integration/test_net_mi355x.cpp,
    run_devices): the Net's tensors, every tensor of the plan's arena and every operator context are on the thread's device; no launch
    or copy went to another device's stream or memory; devices 1 .. 7 received identical allocation counts, bytes and launch counts (and
    device 0 the same launches) - nothing silently landed on device 0; the C ABI's per-device / per-thread caches (zero pages,
    api_conv.hip; Gemm plans, api_gemm.hip) follow the calling device."""
    model = W.build_model("resnet50")
    x = W.make_input(2)
    scales = W.calibrate(model, x)
    d = str(tmp_path)
    mt, wb = NM.write_model(model, scales, 2, d, "int8", calibrator_config=True)
    x.tofile(os.path.join(d, "input.bin"))
    env = dict(os.environ, LD_PRELOAD=MOCK, SABER_MI355X_NET_PLAN_TUNE="0", MOCK_HIP_DEVICES="8")
    r = subprocess.run([BIN, mt, wb, os.path.join(d, "input.bin"), d, "devices", "8"], env=env, capture_output=True, text=True,
                       errors="replace", cwd=d, timeout=900)
    txt = open(os.path.join(d, "devices.txt")).read() if os.path.exists(os.path.join(d, "devices.txt")) else ""
    assert r.returncode == 0, txt + r.stdout[-2000:] + r.stderr[-2000:]
    lines = txt.strip().split("\n")
    assert lines[-1] == "devices 8 bad 0 mock 1", txt
    assert sum(1 for l in lines if l.startswith("device ") and "plan 1" in l and l.endswith("wrong 0")) == 8, txt
    assert "wrong_device_launch 0 wrong_device_copy 0 wrong_device_event 0 uneven 0" in txt, txt
