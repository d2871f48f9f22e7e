"""Whole-network parity on the GPU: every edge tensor of the ResNet50 INT8 op list must be
bit-identical to the CPU oracle's forward pass, for the reference (unfused) op list and for the
fused-epilogue list, eager and hipGraph-replayed."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from anakin_amd import lib as L  # noqa: E402
from anakin_amd import workloads as W  # noqa: E402
from tests import py_fuser as PF  # noqa: E402  (the Python fuser: test infrastructure)
from oracle import net_oracle as NO  # noqa: E402


@pytest.fixture(scope="module")
def setup():
    L.require_device()
    model = W.build_model("resnet50")
    x = W.make_input(2, hw=224)
    scales = W.calibrate(model, x)
    ref = NO.run_int8(model, scales, x)
    return model, x, scales, ref


def _h(t):
    torch.cuda.synchronize()
    return t.cpu().numpy()


@pytest.mark.parametrize("fuse", [False, "lanes", True, "chain3"])
def test_resnet50_int8_every_edge_bit_exact(setup, fuse):
    """False: the reference op list one to one. True: fused eltwise epilogues, sibling pairs, conv1+pool1 as
    SaberConv2DPooling, pool5 writing the fc's quantised input, conv1x1 chains. "chain3": the default executor options = all of
    that plus the 3x3 convs leading their chain launch (their own output edges stay in LDS and are skipped here; every
    written edge and the logits must still match). "lanes": fused epilogues with two-lane execution instead of the pairs."""
    model, x, scales, ref = setup
    net = PF.build_int8_net(model, dict(scales), 2, fuse_eltwise=bool(fuse), lanes=fuse == "lanes",
                           chain=None if fuse in (False, "lanes", "chain3") else 1)
    net.tensor("data").copy_(torch.from_numpy(x).cuda())
    net.run()
    checked = 0
    unwritten = [n for n in net.tensors if net.unwritten(n)]
    # (chain3: seven 3x3 edges stay in LDS; with the sibling pairs formed, pool1 stays inside the stem launch: flag 512)
    assert len(unwritten) == {False: 0, "lanes": 0, True: 1, "chain3": 8}[fuse], unwritten
    for name in net.tensors:
        if name in unwritten:
            checked += 1
            continue
        if name in ref and name != "data":
            got, want = _h(net.tensor(name)), ref[name]
            if got.dtype == np.float32:
                want = want.reshape(got.shape)
                if name == "prob":
                    assert np.abs(got - want).max() <= 1e-4 * want.max()
                else:
                    assert np.array_equal(got, want), name   # pool5 / fc logits are exact too
            else:
                assert np.array_equal(got, want), name
            checked += 1
    assert checked >= (39 if fuse else 70)
    # hipGraph replay must reproduce the eager result; so must the autotuned (RUNTIME) tiles
    logits = _h(net.tensor("fc1000")).copy()
    net.tensor("fc1000").zero_()
    net.capture()
    net.replay()
    assert np.array_equal(_h(net.tensor("fc1000")), logits)
    net.autotune(iters=2)
    net.tensor("data").copy_(torch.from_numpy(x).cuda())
    net.run()
    assert np.array_equal(_h(net.tensor("fc1000")), logits)


FP32_RTOL = 1e-4   # BASELINE.json north_star: FP32 "within 1e-4 rel"


def _fp32_every_edge(model_name, batch):
    """FP32 op list at BASELINE.json's full input size: EVERY logical edge is compared with the oracle right after the op
    that produces it (later in-place residual sums reuse the buffers), with two criteria:
      max-norm      max|got - ref| / max|ref|                         <= 1e-4   (tensor_cmp_host style)
      element-wise  max over elements |got - ref| / (|ref| + mean|ref|) <= 1e-4   (relative, with the tensor's mean
                    magnitude as the absolute floor: outputs that cancel to ~0 have no meaningful relative error)
    Returns the worst of each for the assertion message."""
    model = W.build_model(model_name)
    x = W.make_input(batch, hw=224)
    ref = NO.run_fp32(model, x)
    net = W.build_fp32_net(model, batch, hw=224)
    net.tensor("data").copy_(torch.from_numpy(x).cuda())
    done, worst_max, worst_el, checked = -1, 0.0, 0.0, 0
    for idx, name in net.produced:
        while done < idx:
            done += 1
            net.run_op(done)
        got = _h(net.tensor(net.alias.get(name, name)))
        want = ref[name]
        got = got.transpose(0, 3, 1, 2) if got.ndim == 4 else got.reshape(want.reshape(got.shape[0], -1).shape)
        want = want.reshape(got.shape)
        d = np.abs(got - want)
        e_max = float(d.max() / np.abs(want).max())
        e_el = float((d / (np.abs(want) + np.abs(want).mean())).max())
        assert e_max <= FP32_RTOL and e_el <= FP32_RTOL, (model_name, name, e_max, e_el)
        worst_max, worst_el, checked = max(worst_max, e_max), max(worst_el, e_el), checked + 1
    assert done == net.num_ops() - 1
    return checked, worst_max, worst_el


@pytest.mark.parametrize("b3", ["0", "1"])
def test_fp32_networks_every_edge_with_and_without_the_bf16_plane_kernels(b3, monkeypatch):
    """The same every-edge FP32 criteria with the STATIC choice forced to the f32-MFMA kernels (SABER_HIP_F32_BF16X3=0) and to
    the bf16-plane kernels for every eligible convolution (=1): ResNet50 batch 1 and VGG16 batch 1 at 224 x 224."""
    monkeypatch.setenv("SABER_HIP_F32_BF16X3", b3)
    for name, floor in (("resnet50", 56), ("vgg16", 17)):
        checked, worst_max, worst_el = _fp32_every_edge(name, 1)
        assert checked >= floor, (name, checked)
        print("%s FP32, bf16x3=%s: %d edges, worst max-norm %.2e, worst element-wise %.2e" % (name, b3, checked, worst_max, worst_el))


def test_resnet50_fp32_batch2_full_size_every_edge():
    """BASELINE.json config 2 at its stated size (224x224), batch 2: 53 convs (16 with the fused in-place residual sum),
    2 pools, fc, softmax - every edge within 1e-4."""
    checked, worst_max, worst_el = _fp32_every_edge("resnet50", 2)
    assert checked >= 56, checked


def test_vgg16_fp32_full_size_every_edge_to_the_logits():
    """BASELINE.json config 4 (VGG16 FP32) at 224x224, batch 1: all 13 convs, 5 max pools, 3 fc (with the NCHW-flatten
    weight reorder) and the softmax against the oracle's full forward."""
    checked, worst_max, worst_el = _fp32_every_edge("vgg16", 1)
    assert checked >= 17, checked   # 8 conv edges + 5 fused conv+relu+pool stages (their conv edges do not exist) + 3 fc + prob
    # the five fused SaberConv2DPooling launches write the same bytes as conv followed by pooling
    model = W.build_model("vgg16")
    x = W.make_input(1, hw=224)
    nets = [W.build_fp32_net(model, 1, hw=224, fuse_pool=f) for f in (True, False)]
    assert nets[1].num_ops() - nets[0].num_ops() == 5
    for net in nets:
        net.tensor("data").copy_(torch.from_numpy(x).cuda())
        net.run()
    for name in ("pool1", "pool3", "pool5", "fc8"):
        assert np.array_equal(_h(nets[0].tensor(name)), _h(nets[1].tensor(name))), name


def test_resnet101_int8_full_size_every_edge_bit_exact():
    """BASELINE.json config 4 (ResNet101 INT8) at 224x224, batch 1: every edge the fused op list materialises (70+: all
    branch2a / 2b outputs, the 33 block outputs, pools, logits) bit-identical to the oracle's unfused op list."""
    model = W.build_model("resnet101")
    x = W.make_input(1, hw=224)
    scales = W.calibrate(model, x)
    ref = NO.run_int8(model, scales, x)
    net = PF.build_int8_net(model, dict(scales), 1)
    net.tensor("data").copy_(torch.from_numpy(x).cuda())
    net.run()
    checked = 0
    for name in net.tensors:
        if net.unwritten(name):            # a 3x3 conv's edge that stays in LDS (conv3x3 + chain launch)
            checked += 1
            continue
        if name in ref and name not in ("data", "prob"):
            got, want = _h(net.tensor(name)), ref[name]
            assert np.array_equal(got, want.reshape(got.shape)), name
            checked += 1
    assert checked >= 70, checked
    assert "fc1000" in net.tensors and "res4b22" in net.tensors and "res5c" in net.tensors


def test_resnet50_fp32_within_tolerance():
    model = W.build_model("resnet50")
    x = W.make_input(1, hw=64)
    ref = NO.run_fp32(model, x)
    net = W.build_fp32_net(model, 1, hw=64)
    net.tensor("data").copy_(torch.from_numpy(x).cuda())
    net.run()
    got = _h(net.tensor("fc1000"))
    want = ref["fc1000"]
    assert np.abs(got - want).max() <= 1e-4 * np.abs(want).max()
    mid = _h(net.tensor("res2a_branch2b")).transpose(0, 3, 1, 2)
    assert np.abs(mid - ref["res2a_branch2b"]).max() <= 1e-4 * np.abs(ref["res2a_branch2b"]).max()


def test_resnet101_int8_logits_bit_exact():
    """BASELINE.json config 4 (ResNet101 INT8): 104 convs, same op kinds; logits must match the oracle exactly."""
    model = W.build_model("resnet101")
    x = W.make_input(1, hw=96)
    scales = W.calibrate(model, x)
    ref = NO.run_int8(model, scales, x)
    net = PF.build_int8_net(model, dict(scales), 1, fuse_eltwise=True, hw=96)
    net.tensor("data").copy_(torch.from_numpy(x).cuda())
    net.run()
    assert np.array_equal(_h(net.tensor("fc1000")), ref["fc1000"])
    assert np.array_equal(_h(net.tensor("res4b22")), ref["res4b22"])


def test_vgg16_fp32_within_tolerance():
    """BASELINE.json config 4 (VGG16 FP32): 13 3x3 convs + 5 max pools + 3 FC, FP32 NHWC on the device."""
    model = W.build_model("vgg16")
    x = W.make_input(1, hw=224)
    net = W.build_fp32_net(model, 1, hw=224)
    net.tensor("data").copy_(torch.from_numpy(x).cuda())
    net.run()
    # oracle for the first two conv layers + pool (the full 15.5 GMAC naive pass is too slow for a unit test)
    from oracle import oracle as O
    w1, b1 = model["params"]["conv1"]
    w2, b2 = model["params"]["conv2"]
    y1 = O.conv_f32_nchw(x, w1, b1, True, (1, 1))
    y2 = O.conv_f32_nchw(y1, w2, b2, True, (1, 1))
    p1 = O.pool_f32_nchw(y2, (2, 2), (2, 2), (0, 0), 0)
    got = _h(net.tensor("pool1")).transpose(0, 3, 1, 2)
    assert np.abs(got - p1).max() <= 1e-4 * np.abs(p1).max()
    prob = _h(net.tensor("prob"))
    assert prob.shape == (1, 1000) and abs(float(prob.sum()) - 1.0) < 1e-4 and np.isfinite(prob).all()


def test_resnet50_int8_batch_invariance_full_size(setup):
    """Size-independent property at BASELINE's full batch-8 size: every op on the path is per-image, so
    image i of a batch-8 run (autotuned tiles, hipGraph) must equal the batch-1 run of the same image bit
    for bit, and both must equal the oracle's logits for the images it was run on."""
    model, x2, scales, ref = setup
    x8 = np.concatenate([x2, W.make_input(6, seed=99)], 0)
    net8 = PF.build_int8_net(model, dict(scales), 8)
    net8.tensor("data").copy_(torch.from_numpy(x8).cuda())
    net8.run()
    net8.autotune(iters=3)
    net8.tensor("data").copy_(torch.from_numpy(x8).cuda())
    net8.capture()
    net8.replay()
    l8 = _h(net8.tensor("fc1000")).copy()
    assert np.array_equal(l8[:2], ref["fc1000"])          # the two images the oracle ran
    net1 = PF.build_int8_net(model, dict(scales), 1)
    for i in (0, 5, 7):
        net1.tensor("data").copy_(torch.from_numpy(x8[i:i + 1]).cuda())
        net1.run()
        assert np.array_equal(_h(net1.tensor("fc1000"))[0], l8[i]), i


def test_cxx_net_optimize_equals_python_fused_list(setup):
    """saber_hip_net_optimize (C++ host side) on the reference op list handed over UNFUSED finds the same fusions the
    Python list builder applies (16 conv+eltwise, 4 sibling pairs, conv1+pool1, pool5 -> fc quantisation): 73 ops -> 52
    launches, and every surviving edge + the logits are bit-identical to the Python-fused list's and to the oracle's."""
    model, x, scales, ref = setup
    a = PF.build_int8_net(model, dict(scales), 2)                      # fused by the Python fuser (tests/py_fuser.py)
    b = W.build_int8_net(model, dict(scales), 2)                       # the product: unfused list + saber_hip_net_optimize
    assert b.unfused_ops == 73 and b.removed == 22 and b.num_ops() == a.num_ops() == 52, (b.unfused_ops, b.removed, b.num_ops())
    # ... and both lists then get the conv1x1 chains (branch2c + sum -> next branch2a): 12 candidates, the 10 with C <= 256 on
    # and where the chain head is the only reader of the block's 3x3 conv that conv leads the launch (5 with C <= 128 on)
    # (+ conv3x3 + conv1x1 in the last blocks of res2 / res3, whose 1x1 conv heads no chain)
    # (+ the stem launch running the res2a sibling pair: flag 512)
    assert a.stem_paired == b.stem_paired == 1
    # (+ the fc running the Softmax that reads it: flag 4096)
    assert a.fc_softmaxed == b.fc_softmaxed == 1
    assert a.chained == b.chained == 17 and a.num_launches() == b.num_launches() == 33, (a.chained, a.num_launches())
    assert [a.op_name(i) for i in range(52)] == [b.op_name(i) for i in range(52)]
    for net in (a, b):
        net.tensor("data").copy_(torch.from_numpy(x).cuda())
        net.run()
    checked = 0
    for name in a.tensors:
        if name in b.tensors and name != "data":
            if a.unwritten(name) or b.unwritten(name):
                assert a.unwritten(name) and b.unwritten(name), name
                checked += 1
                continue
            ga, gb = _h(a.tensor(name)), _h(b.tensor(name))
            assert np.array_equal(ga, gb), name
            if name in ref and name != "prob":
                assert np.array_equal(ga, ref[name].reshape(ga.shape)), name
            checked += 1
    assert checked >= 39, checked


def test_strided_head_chain_with_the_next_pair_in_the_net():
    """saber_hip_net_optimize flag 1024 (opt-in: measured no faster): res2c's strided-head chain launch also runs the res3a
    sibling pair. One launch fewer, every written edge and the logits bit-identical to the default list's and to the oracle's - eager,
    after autotuning (which may switch the site back to separate launches) and through the saved selection."""
    model = W.framework_model(W.build_model("resnet50"), "int8")      # the reference optimiser's list: the stride moved up into res2c's 3x3
    x = W.make_input(2, hw=224)
    scales = W.calibrate(model, x)
    ref = NO.run_int8(model, dict(scales), x)
    a = PF.build_int8_net(model, dict(scales), 2)
    b = PF.build_int8_net(model, dict(scales), 2, head_pair=True)
    assert b.chained == a.chained + 1 and b.num_launches() == a.num_launches() - 1, (a.chained, b.chained, a.num_launches(), b.num_launches())
    assert sum("conv3x3+conv1x1+pair1x1_c64" in b.op_name(i) for i in range(b.num_ops())) == 1
    for net in (a, b):
        net.tensor("data").copy_(torch.from_numpy(x).cuda())
        net.run()

    def same():
        for name in a.tensors:
            if name == "data" or a.unwritten(name) or b.unwritten(name):
                continue
            assert np.array_equal(_h(a.tensor(name)), _h(b.tensor(name))), name
        assert np.array_equal(_h(b.tensor("fc1000")), ref["fc1000"].reshape(2, -1))
    same()
    b.autotune(iters=3)
    for name in b.tensors:
        if name != "data" and not b.unwritten(name):
            b.tensor(name).zero_()
    b.tensor("data").copy_(torch.from_numpy(x).cuda())
    b.run()
    same()
    c = PF.build_int8_net(model, dict(scales), 2, head_pair=True)
    c.set_choices(b.choices())
    assert [b.op_name(i) for i in range(b.num_ops())] == [c.op_name(i) for i in range(c.num_ops())]
    c.tensor("data").copy_(torch.from_numpy(x).cuda())
    c.run()
    assert np.array_equal(_h(c.tensor("fc1000")), ref["fc1000"].reshape(2, -1))


def test_autotuned_selection_round_trips_through_choices(setup):
    """Net.choices() / set_choices() (the bench's --tune-cache) carry the whole autotuned selection - kernel variant per op
    AND the chain decisions (separate launches / conv1x1 chain / chain led by the 3x3 conv, with their tile sizes) - into a
    freshly built net: same op names, same launch count, bit-identical logits."""
    model, x, scales, ref = setup
    a = PF.build_int8_net(model, dict(scales), 2)
    a.tensor("data").copy_(torch.from_numpy(x).cuda())
    a.run()
    a.autotune(iters=3)
    a.run()
    torch.cuda.synchronize()
    b = PF.build_int8_net(model, dict(scales), 2)
    b.set_choices(a.choices())
    assert [a.op_name(i) for i in range(a.num_ops())] == [b.op_name(i) for i in range(b.num_ops())]
    assert a.num_launches() == b.num_launches()
    b.tensor("data").copy_(torch.from_numpy(x).cuda())
    b.run()
    torch.cuda.synchronize()
    assert np.array_equal(_h(a.tensor("fc1000")), _h(b.tensor("fc1000")))
    assert np.array_equal(_h(b.tensor("fc1000")), ref["fc1000"].reshape(2, -1))


def test_measurement_entry_points_agree_with_each_other(setup):
    """What bench.py's roofline rests on: saber_hip_net_time_pass (an event per launch: shares of a pass), _time_op_in_pass (two
    events around ONE launch inside otherwise untimed passes - the figure a rocprofv3 trace shows) and _op_work (algorithmic bytes /
    ops per launch): every launching op has a positive time and work, ops absorbed into a chain / stage launch report 0,
    and the bracketed time of the longest launch is of the order of its share (the markers stretch a pass, so the two differ -
    by tens of per cent, not by factors)."""
    model, x, scales, ref = setup
    net = PF.build_int8_net(model, dict(scales), 2)
    net.tensor("data").copy_(torch.from_numpy(x).cuda())
    net.run()
    shares = net.time_pass(iters=5)
    names = [net.op_name(i) for i in range(net.num_ops())]
    launching = [i for i, nm in enumerate(names) if "(in the " not in nm]
    assert len(launching) == net.num_launches()
    for i, nm in enumerate(names):
        by, fl = net.op_work(i)
        if i in launching:
            assert shares[i] > 0 and by > 0, (i, nm)
        else:
            assert shares[i] == 0 and by == 0 and fl == 0, (i, nm)
    top = max(launching, key=lambda i: shares[i])
    t = net.time_op_in_pass(top, iters=5)
    assert 0.5 * shares[top] < t < 2.0 * shares[top] + 5.0, (names[top], t, shares[top])
    with pytest.raises(L.SaberHipError):
        net.time_op_in_pass(net.num_ops(), iters=5)


# ---------------------------------------------------------------------------------------------------------------------
# The list the reference's OWN optimiser emits (workloads.framework_spec; tests/test_net_oplist.py proves the equality,
# tests/test_gpu_net.py runs it through the reference's Net): stride-up (three stride-2 3x3 convs + 1x1/2 shortcut
# poolings), conv1 -> s8, INT8 average pooling + s8-input fc. Same executor, same fusions.
@pytest.fixture(scope="module")
def setup_fw():
    L.require_device()
    model = W.framework_model(W.build_model("resnet50"), "int8")
    x = W.make_input(2, hw=224)
    scales = W.calibrate(model, x)
    ref = NO.run_int8(model, scales, x)
    return model, x, scales, ref


@pytest.mark.parametrize("fuse", [False, "lanes", True, "chain3", "cxx"])
def test_resnet50_int8_framework_list_every_edge_bit_exact(setup_fw, fuse):
    model, x, scales, ref = setup_fw
    if fuse == "cxx":      # the PRODUCT builder: the list one op per reference operator + saber_hip_net_optimize (the only product fuser)
        net = W.build_int8_net(model, dict(scales), 2)
    else:
        net = PF.build_int8_net(model, dict(scales), 2, fuse_eltwise=bool(fuse), lanes=fuse == "lanes",
                                chain=None if fuse in (False, "lanes", "chain3") else 1)
    if not fuse:
        assert net.num_ops() == 76           # one op per reference operator
    net.tensor("data").copy_(torch.from_numpy(x).cuda())
    net.run()
    checked = 0
    # "cxx": the list is handed over unfused and saber_hip_net_optimize removes ops; the edges of removed ops (conv1 inside
    # SaberConv2DPooling, the separate eltwise inputs) stay declared but are never written: compare what the Python-fused
    # list materialises
    live = set(PF.build_int8_net(model, dict(scales), 2).tensors) if fuse == "cxx" else None
    for name in net.tensors:
        if net.unwritten(name) or (live is not None and name not in live):
            checked += 1
            continue
        if name in ref and name != "data":
            got, want = _h(net.tensor(name)), ref[name]
            if name == "prob":
                assert np.abs(got - want.reshape(got.shape)).max() <= 1e-4 * want.max()
            else:
                assert got.dtype == want.dtype, (name, got.dtype, want.dtype)
                assert np.array_equal(got, want.reshape(got.shape)), name
            checked += 1
    assert checked >= (40 if fuse else 76), checked
    logits = _h(net.tensor("fc1000")).copy()
    net.tensor("fc1000").zero_()
    net.capture()
    net.replay()
    assert np.array_equal(_h(net.tensor("fc1000")), logits)
    net.autotune(iters=2)
    net.tensor("data").copy_(torch.from_numpy(x).cuda())
    net.run()
    assert np.array_equal(_h(net.tensor("fc1000")), logits)


def test_resnet50_int8_framework_list_batch8_invariance(setup_fw):
    model, x2, scales, ref = setup_fw
    x8 = np.concatenate([x2, W.make_input(6, seed=99)], 0)
    net8 = PF.build_int8_net(model, dict(scales), 8)
    net8.tensor("data").copy_(torch.from_numpy(x8).cuda())
    net8.run()
    net8.autotune(iters=3)
    net8.tensor("data").copy_(torch.from_numpy(x8).cuda())
    net8.capture()
    net8.replay()
    l8 = _h(net8.tensor("fc1000")).copy()
    assert np.array_equal(l8[:2], ref["fc1000"].reshape(2, -1))
    net1 = PF.build_int8_net(model, dict(scales), 1)
    for i in (0, 5, 7):
        net1.tensor("data").copy_(torch.from_numpy(x8[i:i + 1]).cuda())
        net1.run()
        assert np.array_equal(_h(net1.tensor("fc1000"))[0], l8[i]), i


def test_resnet101_int8_framework_list_every_edge_bit_exact():
    """BASELINE.json config 4 on the list the reference's optimiser emits for ResNet101 (tests/test_net_oplist.py): 144 operators,
    stride-up poolings, 8-bit tail; every edge the executor materialises == the oracle running the same list, batch 1 at 224x224;
    the same logits after autotuning."""
    model = W.framework_model(W.build_model("resnet101"), "int8")
    x = W.make_input(1, hw=224)
    scales = W.calibrate(model, x)
    ref = NO.run_int8(model, dict(scales), x)
    net = PF.build_int8_net(model, dict(scales), 1)
    net.tensor("data").copy_(torch.from_numpy(x).cuda())
    net.run()
    checked = 0
    for name in net.tensors:
        if net.unwritten(name):
            checked += 1
            continue
        if name in ref and name not in ("data", "prob"):
            got, want = _h(net.tensor(name)), ref[name]
            assert got.dtype == want.dtype, (name, got.dtype, want.dtype)
            assert np.array_equal(got, want.reshape(got.shape)), name
            checked += 1
    assert checked >= 70, checked
    logits = _h(net.tensor("fc1000")).copy()
    net.autotune(iters=2)
    net.tensor("data").copy_(torch.from_numpy(x).cuda())
    net.run()
    assert np.array_equal(_h(net.tensor("fc1000")), logits)


def test_resnet50_int8_framework_list_batch16_invariance(setup_fw):
    """Batch 16 (two images per XCD-sized group for the image-resident last conv + pooling, M doubled everywhere else): image i of the
    batch == the same image run alone, after autotuning, eager and as a hipGraph."""
    model, x2, scales, ref = setup_fw
    x16 = np.concatenate([x2, W.make_input(14, seed=123)], 0)
    net = PF.build_int8_net(model, dict(scales), 16)
    net.tensor("data").copy_(torch.from_numpy(x16).cuda())
    net.run()
    net.autotune(iters=2)
    net.tensor("data").copy_(torch.from_numpy(x16).cuda())
    net.run()
    l16 = _h(net.tensor("fc1000")).copy()
    assert np.array_equal(l16[:2], ref["fc1000"].reshape(2, -1))
    net.capture()
    net.tensor("fc1000").zero_()
    net.replay()
    assert np.array_equal(_h(net.tensor("fc1000")), l16)
    net1 = PF.build_int8_net(model, dict(scales), 1)
    for i in (3, 9, 15):
        net1.tensor("data").copy_(torch.from_numpy(x16[i:i + 1]).cuda())
        net1.run()
        assert np.array_equal(_h(net1.tensor("fc1000"))[0], l16[i]), i


# ---------------------------------------------------------------------------------------------------------------------------------
# Round 6: the executor's arena with lifetime aliasing (saber_hip_net_compact_arena; the reference's MemoryScheduler role,
# framework/graph/llvm/optimizer/memory_scheduler.cpp). Materialise-everything stays the mode of every parity test above.
def _all_modes(net, fn):
    """runs fn() under the static selection, then with every stage / chain decision flipped off and on again (restored selections
    change WHICH op launches which tensors: the compacted layout must be valid for all of them)"""
    base = net.choices()
    fn("static")
    # (saber_hip_net_get_choice: bits 28 / 29 = a chain / 3x3-led chain decision is recorded, bits 24..27 its pixel fragments with
    # 0 = "run as separate launches", bit 30 = this op launches its whole stage)
    off = [(c & ~(0xf << 24)) & ~(1 << 30) for c in base]
    try:
        net.set_choices(off)
        fn("chains_off")
    finally:
        net.set_choices(base)
    fn("restored")


@pytest.mark.parametrize("batch", [2, 8])
def test_compacted_arena_int8_same_bits_smaller_footprint(setup_fw, batch):
    """ResNet50 INT8, the driver's configuration (C++-fused framework list, stage + stem pair): outputs before compaction ==
    outputs after, bit for bit - eager, hipGraph, and with the chain / stage decisions switched off and on AFTER the compaction -
    and the footprint drops several-fold. Inputs and outputs stay readable; `keep` pins an intermediate edge."""
    model, _, scales, _ = setup_fw
    x = W.make_input(batch)
    net = W.build_int8_net(model, dict(scales), batch)
    net.tensor("data").copy_(torch.from_numpy(x).cuda())
    net.run()
    want = {n: _h(net.tensor(n)).copy() for n in ("fc1000", "prob", "res3a")}
    before = net.arena_bytes()
    after = net.compact(keep=("res3a",))
    assert net.compacted() and after * 2 < before, (before, after)
    assert np.array_equal(_h(net.tensor("data")), x)                 # the pass's input survived the move

    def check(tag):
        for mode in ("eager", "graph"):
            net.tensor("fc1000").fill_(0)
            net.tensor("prob").fill_(0)
            if mode == "graph":
                net.capture()
                net.replay()
            else:
                net.run()
            for n in want:
                assert np.array_equal(_h(net.tensor(n)), want[n]), (tag, mode, n)
    _all_modes(net, check)
    # several passes back to back (a slot reused by a later edge must not leak into the next pass)
    for _ in range(3):
        net.run()
    assert np.array_equal(_h(net.tensor("fc1000")), want["fc1000"])


def test_compacted_arena_fp32_and_two_nets_side_by_side():
    """FP32 ResNet50 (in-place residual sums: the ConvEltwise output aliases the shortcut's buffer already) through the compacted
    arena within bit identity of the materialised run; two compacted nets on two streams do not disturb each other."""
    model = W.build_model("resnet50")
    x = W.make_input(2)
    nets = []
    for i in range(2):
        net = W.build_fp32_net(model, 2, shared_device=True)
        net.tensor("data").copy_(torch.from_numpy(x if i == 0 else x[::-1].copy()).cuda())
        net.run()
        torch.cuda.synchronize()
        want = _h(net.tensor("fc1000")).copy()
        before = net.arena_bytes()
        after = net.compact()
        assert after * 2 < before, (before, after)
        nets.append((net, want))
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    for _ in range(5):
        for (net, _), st in zip(nets, streams):
            with torch.cuda.stream(st):
                net.run()
    torch.cuda.synchronize()
    for net, want in nets:
        assert np.array_equal(_h(net.tensor("fc1000")), want)


def test_compact_arena_leaves_a_two_lane_net_alone(setup_fw):
    model, _, scales, _ = setup_fw
    net = PF.build_int8_net(model, dict(scales), 2, fuse_eltwise=True, lanes=True)
    before = net.arena_bytes()
    assert net.compact() == before and not net.compacted()


def test_reproducible_fp32_flag_two_autotuned_nets_same_bits():
    """saber_hip_net_optimize flag 8192 (SABER_HIP_NET_REPRODUCIBLE_FP32): the autotuner and a restored selection leave FP32 ops on
    their static kernels - two nets of one model, each "autotuned" on its own, answer bit-identically on every edge they share; without
    the flag the autotuner does move FP32 ops (that is its job) and only the 1e-4 contract holds."""
    model = W.build_model("resnet50")
    x = W.make_input(2)
    outs, names = [], []
    for i in range(2):
        net = W.build_fp32_net(model, 2, reproducible=True)
        static_names = [net.op_name(k) for k in range(net.num_ops())]
        net.tensor("data").copy_(torch.from_numpy(x).cuda())
        net.run()
        net.autotune(iters=7)
        assert [net.op_name(k) for k in range(net.num_ops())] == static_names      # nothing moved
        net.tensor("data").copy_(torch.from_numpy(x).cuda())
        net.run()
        outs.append({n: _h(net.tensor(n)).copy() for n in ("fc1000", "prob", "pool1")})
        names.append(static_names)
    for n in outs[0]:
        assert np.array_equal(outs[0][n], outs[1][n]), n
    ref = NO.run_fp32(model, x)
    got = outs[0]["fc1000"]
    want = ref["fc1000"].reshape(got.shape)
    assert np.abs(got - want).max() <= 1e-4 * np.abs(want).max()


def _random_resnet_like(rng):
    """A ResNet-shaped network with widths / depths / input size nobody tuned for: stem conv (3x3 or 7x7, stride 2) [+ max pooling], 2 - 3 stages of
    1 - 3 bottleneck blocks (mid channels 16 .. 256, x4 expansion, Caffe's stride placement), global pooling, fc, softmax - the operator patterns
    saber_hip_net_optimize looks for (sibling pairs, 3x3-led chains, strided heads, stage runs, eltwise epilogues, pooling absorption), at sizes
    where some of its kernels apply and some do not."""
    stem_c = int(rng.choice([16, 32, 64]))
    stem_k = int(rng.choice([3, 7]))
    options = [[16, 32], [32, 64], [64, 128], [64, 128, 256], [32, 64, 128], [64, 256], [128, 256]]
    mids = options[int(rng.integers(0, len(options)))]
    nblk = [int(rng.integers(1, 4)) for _ in mids]
    hw = int(rng.choice([32, 40, 56, 64, 72, 96]))
    pool = bool(rng.integers(0, 2))
    classes = int(rng.choice([10, 100, 1000]))
    L = [dict(kind="conv", name="conv1", src="data", cin=3, cout=stem_c, k=stem_k, stride=2, pad=stem_k // 2, relu=True)]
    prev, cin = "conv1", stem_c
    if pool:
        L.append(dict(kind="pool", name="pool1", src="conv1", win=3, stride=2, pad=0, type=0))
        prev = "pool1"
    for si, (mid, nb) in enumerate(zip(mids, nblk)):
        cout = mid * 4
        for bi in range(nb):
            stride = 2 if (bi == 0 and si > 0) else 1
            tag = "res%d%s" % (si + 2, chr(ord("a") + bi))
            if bi == 0:
                L.append(dict(kind="conv", name=tag + "_branch1", src=prev, cin=cin, cout=cout, k=1, stride=stride, pad=0, relu=False))
                shortcut = tag + "_branch1"
            else:
                shortcut = prev
            L.append(dict(kind="conv", name=tag + "_branch2a", src=prev, cin=cin, cout=mid, k=1, stride=stride, pad=0, relu=True))
            L.append(dict(kind="conv", name=tag + "_branch2b", src=tag + "_branch2a", cin=mid, cout=mid, k=3, stride=1, pad=1, relu=True))
            L.append(dict(kind="conv", name=tag + "_branch2c", src=tag + "_branch2b", cin=mid, cout=cout, k=1, stride=1, pad=0, relu=False, eltwise=tag))
            L.append(dict(kind="eltwise", name=tag, a=tag + "_branch2c", b=shortcut, relu=True))
            prev, cin = tag, cout
    L.append(dict(kind="gpool", name="pool5", src=prev))
    L.append(dict(kind="fc", name="fc", src="pool5", cin=cin, cout=classes))
    L.append(dict(kind="softmax", name="prob", src="fc"))
    params, raw = {}, {}
    for idx, l in enumerate(L):
        if l["kind"] == "conv":
            c, k, ks = l["cin"], l["cout"], l["k"]
            w = (rng.standard_normal((k, c, ks, ks)) * np.sqrt(2.0 / (c * ks * ks))).astype(np.float32)
            gamma, beta = rng.uniform(0.5, 1.5, k).astype(np.float32), rng.uniform(-0.1, 0.1, k).astype(np.float32)
            mean, var = rng.uniform(-0.1, 0.1, k).astype(np.float32), rng.uniform(0.5, 1.5, k).astype(np.float32)
            params[l["name"]] = W.fold_bn(w, None, 1.0, 1e-5, mean, var, gamma, beta)
            raw[l["name"]] = dict(w=w, mean=mean, var=var, gamma=gamma, beta=beta)
        elif l["kind"] == "fc":
            params[l["name"]] = ((rng.standard_normal((l["cout"], l["cin"])) * np.sqrt(1.0 / l["cin"])).astype(np.float32),
                                 rng.uniform(-0.1, 0.1, l["cout"]).astype(np.float32))
    return dict(name="random_resnet", spec=L, params=params, raw=raw), hw


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(12))
def test_random_resnet_like_int8_networks_every_edge_bit_exact(seed):
    """Property test of the executor-level fuser: a random ResNet-shaped INT8 network (widths, depths, stem, input size, batch drawn at random)
    through workloads.framework_spec (the reference's stride-up rule) and saber_hip_net_optimize with every fusion on - whatever it decides to
    fuse, pair, chain or run as a stage at these sizes, every written edge is the oracle's bytes on the unfused list; eagerly, after an
    autotune, and replayed as a hipGraph; and the same network with no executor-level fusion at all."""
    L.require_device()
    rng = np.random.default_rng(4200 + seed)
    model, hw = _random_resnet_like(rng)
    batch = int(rng.choice([1, 2, 3, 8]))
    fm = W.framework_model(model, "int8")
    x = rng.uniform(-1.0, 1.0, (batch, 3, hw, hw)).astype(np.float32)
    scales = W.calibrate(fm, x)
    ref = NO.run_int8(fm, dict(scales), x)

    def check(net, what):
        n = 0
        for nm in net.tensors:
            if nm == "data" or nm not in ref or net.unwritten(nm):
                continue
            got = _h(net.tensor(nm))
            if nm == "prob":
                assert np.abs(got.reshape(batch, -1) - ref[nm].reshape(batch, -1)).max() <= 1e-4 * ref[nm].max(), (what, nm)
            else:
                assert np.array_equal(got, ref[nm].reshape(got.shape)), (what, nm, seed, hw, batch)
            n += 1
        assert n >= 3
        return n

    for fuse in (True, False):
        net = W.build_int8_net(fm, dict(scales), batch, hw=hw, fuse=fuse)
        net.tensor("data").copy_(torch.from_numpy(x).cuda())
        net.run()
        n_edges = check(net, "eager fuse=%s" % fuse)
        if fuse:
            net.autotune(iters=2)
            for nm in net.tensors:
                if nm != "data" and not net.unwritten(nm):
                    net.tensor(nm).zero_()
            net.tensor("data").copy_(torch.from_numpy(x).cuda())
            net.run()
            check(net, "autotuned")
            net.capture()
            for nm in net.tensors:
                if nm != "data" and not net.unwritten(nm):
                    net.tensor(nm).zero_()
            net.tensor("data").copy_(torch.from_numpy(x).cuda())
            net.replay()
            check(net, "hipGraph")
            print("seed %d: %dx%d batch %d, %d layers -> %d ops in %d launches, %d edges checked, stages %s" % (
                seed, hw, hw, batch, len(fm["spec"]), net.num_ops(), net.num_launches(), n_edges, net.stages()))
        assert net.coop_fallbacks() == 0


def _fp32_edges_of(model, x, hw, autotune=False):
    """every logical edge of an FP32 op list against the oracle right after the op that produces it (the criteria of _fp32_every_edge)"""
    ref = NO.run_fp32(model, x)
    net = W.build_fp32_net(model, x.shape[0], hw=hw)
    net.tensor("data").copy_(torch.from_numpy(x).cuda())
    if autotune:
        net.run()
        net.autotune(iters=2)
        net.tensor("data").copy_(torch.from_numpy(x).cuda())
    done, checked = -1, 0
    for idx, name in net.produced:
        while done < idx:
            done += 1
            net.run_op(done)
        got = _h(net.tensor(net.alias.get(name, name)))
        want = ref[name]
        got = got.transpose(0, 3, 1, 2) if got.ndim == 4 else got.reshape(want.reshape(got.shape[0], -1).shape)
        want = want.reshape(got.shape)
        d = np.abs(got - want)
        e_max = float(d.max() / max(np.abs(want).max(), 1e-12))
        e_el = float((d / (np.abs(want) + np.abs(want).mean() + 1e-12)).max())
        assert e_max <= FP32_RTOL and e_el <= FP32_RTOL, (name, net.op_name(idx), e_max, e_el)
        checked += 1
    assert done == net.num_ops() - 1
    return checked, net


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(8))
def test_random_resnet_like_fp32_networks_every_edge_within_tolerance(seed):
    """The FP32 side of the property test: the same random ResNet-shaped networks through workloads.build_fp32_net (fused stem where it applies,
    sibling pairs, in-place residual sums, whatever FP32 kernel the static choice - and then the autotuner - picks at these sizes): every
    edge within 1e-4 of the oracle on both criteria."""
    L.require_device()
    rng = np.random.default_rng(4200 + seed)
    model, hw = _random_resnet_like(rng)
    batch = int(rng.choice([1, 2, 3, 8]))
    x = rng.uniform(-1.0, 1.0, (batch, 3, hw, hw)).astype(np.float32)
    n0, net = _fp32_edges_of(model, x, hw)
    n1, net = _fp32_edges_of(model, x, hw, autotune=True)
    assert n0 == n1 and n0 >= 5
    print("seed %d: %dx%d batch %d, %d layers -> %d ops, %d edges checked twice" % (seed, hw, hw, batch, len(model["spec"]), net.num_ops(), n0))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(6))
def test_random_vgg_like_fp32_networks_every_edge_within_tolerance(seed):
    """A VGG-shaped FP32 network at random widths / depths / input sizes (odd sizes too: the fused conv + relu + 2x2 pooling launch needs even
    conv outputs, elsewhere the two ops stay apart), conv + bias + relu, max pooling in ceil mode, fc + relu over the NCHW flatten, softmax."""
    L.require_device()
    rng = np.random.default_rng(5100 + seed)
    hw = int(rng.choice([24, 30, 33, 40, 56]))
    nstage = int(rng.integers(2, 4))
    Lspec, prev, cin, i, size = [], "data", 3, 0, hw
    for s in range(nstage):
        width = int(rng.choice([16, 32, 64, 96]))
        for _ in range(int(rng.integers(1, 3))):
            i += 1
            Lspec.append(dict(kind="conv", name="conv%d" % i, src=prev, cin=cin, cout=width, k=3, stride=1, pad=1, relu=True))
            prev, cin = "conv%d" % i, width
        Lspec.append(dict(kind="pool", name="pool%d" % (s + 1), src=prev, win=2, stride=2, pad=0, type=0))
        prev, size = "pool%d" % (s + 1), -(-size // 2)
    hidden, classes = int(rng.choice([64, 128])), int(rng.choice([10, 100]))
    Lspec.append(dict(kind="fc", name="fc6", src=prev, cin=cin * size * size, cout=hidden, relu=True, flatten_chw=(cin, size, size)))
    Lspec.append(dict(kind="fc", name="fc7", src="fc6", cin=hidden, cout=classes))
    Lspec.append(dict(kind="softmax", name="prob", src="fc7"))
    params = {}
    for l in Lspec:
        if l["kind"] == "conv":
            params[l["name"]] = ((rng.standard_normal((l["cout"], l["cin"], 3, 3)) * np.sqrt(2.0 / (l["cin"] * 9))).astype(np.float32),
                                 rng.uniform(-0.1, 0.1, l["cout"]).astype(np.float32))
        elif l["kind"] == "fc":
            params[l["name"]] = ((rng.standard_normal((l["cout"], l["cin"])) * np.sqrt(1.0 / l["cin"])).astype(np.float32),
                                 rng.uniform(-0.1, 0.1, l["cout"]).astype(np.float32))
    model = dict(name="random_vgg", spec=Lspec, params=params, raw={})
    batch = int(rng.choice([1, 2, 5]))
    x = rng.uniform(-1.0, 1.0, (batch, 3, hw, hw)).astype(np.float32)
    n0, net = _fp32_edges_of(model, x, hw)
    n1, net = _fp32_edges_of(model, x, hw, autotune=True)
    assert n0 == n1 and n0 >= 4
    print("seed %d: %dx%d batch %d, %d layers -> %d ops, %d edges checked twice" % (seed, hw, hw, batch, len(Lspec), net.num_ops(), n0))


@pytest.mark.gpu
def test_resnet50_fp32_soak_split_k_hand_off_every_pass_gives_the_same_bits():
    """ResNet50 FP32 batch 8 with an autotuned selection (the few-pixel layers then run split-K: partial sums handed between workgroups through one
    XCD's L2, summed in split order by the last arrival; the in-place residual sums; the pointwise kernels' reduction split over waves): 4 000
    passes, eager and hipGraph replays alternating, two images - the logits and res4f / res5c after every pass are the bits of the first pass
    (the hand-off's result does not depend on who arrives last), never a NaN (the placement guard's poison), and within 1e-4 of the oracle."""
    L.require_device()
    model = W.build_model("resnet50")
    batch = 8
    net = W.build_fp32_net(model, batch)
    x0 = W.make_input(batch)
    net.tensor("data").copy_(torch.from_numpy(x0).cuda())
    net.run()
    net.autotune(iters=3)
    names = [net.op_name(i) for i in range(net.num_ops())]
    assert any("_split" in n for n in names), names
    net.tensor("data").copy_(torch.from_numpy(x0).cuda())
    net.run()
    net.capture()
    ref = NO.run_fp32(model, x0[:2])
    got = _h(net.tensor(net.alias.get("fc1000", "fc1000")))[:2]
    assert np.abs(got - ref["fc1000"].reshape(got.shape)).max() <= FP32_RTOL * np.abs(ref["fc1000"]).max()
    watch = [net.alias.get(nm, nm) for nm in ("res4f", "res5c", "fc1000")]
    bad = torch.zeros((), dtype=torch.int64, device="cuda")
    nan = torch.zeros((), dtype=torch.int64, device="cuda")
    for img in range(2):
        net.tensor("data").copy_(torch.from_numpy(W.make_input(batch, seed=900 + img)).cuda())
        net.run()
        first = {nm: net.tensor(nm).clone() for nm in watch}
        for it in range(2000):
            if it % 2:
                net.replay()
            else:
                net.run()
            for nm in watch:
                bad += (net.tensor(nm) != first[nm]).any().to(torch.int64)
            nan += torch.isnan(net.tensor(watch[-1])).any().to(torch.int64)
    torch.cuda.synchronize()
    assert int(bad.item()) == 0 and int(nan.item()) == 0, (int(bad.item()), int(nan.item()))
