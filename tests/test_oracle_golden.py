"""The CPU oracle against the committed golden vectors (which were produced by the compiled
reference, tests/golden/make_golden.py). Runs anywhere: no /root/reference, no GPU."""
import numpy as np
import pytest

from oracle import oracle as O
from tests.golden_util import conv_f32_fixtures, conv_i8_fixtures, load


@pytest.mark.parametrize("name", conv_i8_fixtures())
def test_conv_i8_golden(name):
    g = load(name)
    N, H, W, C, K, k, pad, stride, dil, group, idt, odt, relu = [int(v) for v in g["spec"]]
    ws = g["w_scale"]
    if "w" in g.files:  # weight quantisation is part of the contract
        assert np.array_equal(O.weight_scales(g["w"]), ws)
        assert np.array_equal(O.quant_weights(g["w"], ws), g["wq"])
    bp, sc = O.conv_i8_prepare(ws, g["bias"], float(g["in_scale"]), float(g["out_scale"]), idt, odt)
    y = O.conv_i8(g["x"], g["wq"], bp, sc, odt, relu, (pad, pad), (stride, stride), (dil, dil), group)
    assert y.dtype == g["y"].dtype and np.array_equal(y, g["y"])


def test_quant_dequant_golden():
    g = load("quant_dequant")
    s = float(g["scale"])
    assert np.array_equal(O.quant_nchw_to_nhwc(g["x"], s, O.S8), g["q_s8"])
    assert np.array_equal(O.quant_nchw_to_nhwc(g["x"], s, O.U8), g["q_u8"])
    assert np.array_equal(O.dequant_nhwc_to_nchw(g["q_s8"], s), g["deq_s8"])
    assert np.array_equal(O.dequant_nhwc_to_nchw(g["q_u8"], s), g["deq_u8"])


def test_eltwise_golden():
    g = load("eltwise")
    sa, sb, c = float(g["sa"]), float(g["sb"]), float(g["c"])
    assert np.array_equal(O.eltwise_i8(g["a"], g["b"], sa, sb, c, c, True), g["y_relu"])
    assert np.array_equal(O.eltwise_i8(g["a"], g["b"], sa, sb, 1.0, 1.0, False), g["y_lin"])
    assert np.array_equal(O.eltwise_f32(g["fa"], g["fb"], 1.0, 1.0, True), g["yf"])


@pytest.mark.parametrize("name", conv_f32_fixtures())
def test_conv_f32_golden(name):
    g = load(name)
    N, C, H, W, K, k, pad, stride = [int(v) for v in g["spec"]]
    y = O.conv_f32_nchw(g["x"], g["w"], g["bias"], True, (pad, pad), (stride, stride))
    assert np.array_equal(y, g["y"])  # same summation order as conv_basic_check -> bit-exact


def test_conv_f32_residual_mkl_golden():
    g = load("conv_f32_1x1_residual_mkl")
    y = O.conv_f32_nchw(g["x"], g["w"], g["bias"], True, beta=1.0, out_init=g["res"])
    assert np.abs(y - g["y"]).max() <= 1e-4 * np.abs(g["y"]).max()  # MKL sgemm order: tolerance


def test_fc_i8_golden():
    """INT8 fc, f32 input: restatement == the reference's PackedMKLInt8Gemm output stored in the fixture."""
    g = load("fc_i8_f32in")
    w = g["w"].astype(np.float32)
    N, K = w.shape
    ws = O.weight_scales(w.reshape(N, K, 1, 1))
    wq = O.quant_weights(w.reshape(N, K, 1, 1), ws).reshape(N, K)
    got = O.fc_i8(O.quant_flat_s8(g["x"], float(g["in_scale"])), wq, ws, float(g["in_scale"]), g["bias"])
    assert np.array_equal(got, g["y"])


def test_conv_i8_properties():
    """Size-independent properties: linearity of the int32 accumulator in x and in w; fused
    eltwise epilogue == conv followed by the eltwise op."""
    rng = np.random.default_rng(0)
    x1 = rng.integers(-60, 60, (1, 9, 9, 32)).astype(np.int8)
    x2 = rng.integers(-60, 60, (1, 9, 9, 32)).astype(np.int8)
    wq = rng.integers(-127, 128, (16, 32, 3, 3)).astype(np.int8)
    a1 = O.conv_i8_acc(x1, wq, (1, 1))
    a2 = O.conv_i8_acc(x2, wq, (1, 1))
    a12 = O.conv_i8_acc((x1 + x2).astype(np.int8), wq, (1, 1))
    assert np.array_equal(a1 + a2, a12)
    scale = np.full(16, 3e-4, np.float32)
    res = rng.integers(-128, 128, a1.shape).astype(np.int8)
    y = O.conv_i8(x1, wq, None, scale, O.S8, 0, (1, 1))
    two_op = O.eltwise_i8(y, res, 0.05, 0.07, 20.0, 20.0, True)
    rp = O.Residual(O.RES_ELTWISE, 1, 0.0, O.S8, 20.0, 20.0, 0.05, 0.07)
    fused = O.conv_i8(x1, wq, None, scale, O.S8, 0, (1, 1), residual=rp, res=res)
    assert np.array_equal(two_op, fused)


def test_round_identity_exhaustive():
    """The device epilogues compute roundf as trunc(x + copysign(0.49999997, x)); exact for all |x| < 2^23."""
    import ctypes
    lib = O.lib()
    lib.orc_check_round_identity.restype = ctypes.c_long
    assert lib.orc_check_round_identity() == 0
