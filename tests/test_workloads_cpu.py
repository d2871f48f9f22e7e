"""Host-side logic of the synthetic workloads: topology counts, BN folding vs the
oracle's restatement of parameter_fusion.cpp, and the oracle forward pass at reduced resolution."""
import numpy as np

from anakin_amd import workloads as W
from oracle import net_oracle as NO
from oracle import oracle as O


def test_resnet50_topology_counts():
    spec = W.resnet_spec(50)
    convs = [l for l in spec if l["kind"] == "conv"]
    assert len(convs) == 53 and sum(1 for l in spec if l["kind"] == "eltwise") == 16
    macs = W.conv_macs(spec)
    assert abs(macs - 3.858e9) / 3.858e9 < 2e-3          # 3.856 GMAC conv + 2.05 MMAC fc
    assert len([l for l in W.resnet_spec(101) if l["kind"] == "conv"]) == 104
    assert len([l for l in W.vgg16_spec() if l["kind"] == "conv"]) == 13
    m = W.build_model("resnet50")
    nbytes = W.algorithmic_bytes_int8(m, 1)
    assert abs(nbytes - (25.5e6 + 26.2e6)) / 51.7e6 < 0.02   # 25.5 MB + B * 26.2 MB


def test_bn_fold_matches_oracle():
    rng = np.random.default_rng(0)
    w = rng.standard_normal((8, 4, 3, 3)).astype(np.float32)
    mean, var = rng.uniform(-.1, .1, 8).astype(np.float32), rng.uniform(.5, 1.5, 8).astype(np.float32)
    g, b = rng.uniform(.5, 1.5, 8).astype(np.float32), rng.uniform(-.1, .1, 8).astype(np.float32)
    w1, b1 = W.fold_bn(w, None, 1.0, 1e-5, mean, var, g, b)
    w2, b2 = O.bn_fold(w, None, 1.0, 1e-5, mean, var, g, b)
    assert np.array_equal(w1, w2) and np.array_equal(b1, b2)


def test_oracle_forward_small_resolution():
    """The whole INT8 op list through the oracle at 64x64 input: shapes, dtypes, and non-degenerate
    activations (scales calibrated so that 8-bit tensors actually use their range)."""
    m = W.build_model("resnet50")
    x = W.make_input(1, hw=64)
    scales = W.calibrate(m, x)
    t = NO.run_int8(m, scales, x)
    assert t["conv1"].dtype == np.uint8 and t["conv1"].shape == (1, 32, 32, 64)
    assert t["res2a"].dtype == np.int8 and t["res5c"].shape == (1, 2, 2, 2048)
    assert t["prob"].shape == (1, 1000) and abs(t["prob"].sum() - 1) < 1e-4
    for name in ("conv1", "res2a_branch2b", "res3d", "res5c"):
        assert int(np.abs(t[name].astype(np.int32)).max()) >= 64, name
    # INT8 logits track the FP32 logits (the reference's own INT8 acceptance is rel-L2 < 0.15,
    # test_saber_conv_int8.cpp:202)
    f = NO.run_fp32(m, x)
    a, b = t["fc1000"].ravel(), f["fc1000"].ravel()
    assert np.linalg.norm(a - b) / np.linalg.norm(b) < 0.35  # drift accumulates over 53 random-weight layers
