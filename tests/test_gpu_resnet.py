"""Whole-network parity on the GPU: every edge tensor of the ResNet50 INT8 op list must be
bit-identical to the CPU oracle's forward pass, for the reference (unfused) op list and for the
fused-epilogue list, eager and hipGraph-replayed."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from anakin_amd import lib as L  # noqa: E402
from anakin_amd import workloads as W  # noqa: E402
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


@pytest.mark.parametrize("fuse", [False, True])
def test_resnet50_int8_every_edge_bit_exact(setup, fuse):
    model, x, scales, ref = setup
    net = W.build_int8_net(model, dict(scales), 2, fuse_eltwise=fuse)
    net.tensor("data").copy_(torch.from_numpy(x).cuda())
    net.run()
    checked = 0
    for name in net.tensors:
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
    assert checked >= (40 if fuse else 70)
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
