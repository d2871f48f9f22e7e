"""Oracle restatements whose reference sources cannot be compiled here (saber_softmax.cpp, the FP32 / INT8 branches of
saber_pooling.cpp include the xbyak JIT headers) checked against a SECOND, independent source: torch's CPU operators and
plain numpy integer arithmetic. These are checkers of the checker; nothing here touches the product path."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.nn.functional as F  # noqa: E402

from oracle import oracle as O  # noqa: E402


def test_softmax_f32_vs_torch():
    rng = np.random.default_rng(7)
    for shape in ((8, 1000), (3, 17), (1, 1)):
        x = (rng.standard_normal(shape) * 5).astype(np.float32)
        want = torch.softmax(torch.from_numpy(x).double(), 1).numpy()
        got = O.softmax_f32(x)
        assert (np.abs(got - want) <= 4e-6 * want + 1e-30).all() and np.allclose(got.sum(1), 1.0, atol=1e-6)   # f32 expf
    x = np.array([[1000.0, 1000.0, -1000.0]], np.float32)   # max-subtraction keeps it finite
    assert np.allclose(O.softmax_f32(x), [[0.5, 0.5, 0.0]])


@pytest.mark.parametrize("case", [((3, 3), (2, 2), (0, 0)), ((3, 3), (2, 2), (1, 1)), ((2, 2), (2, 2), (0, 0)),
                                  ((3, 3), (1, 1), (1, 1)), ((7, 7), (7, 7), (0, 0))])
def test_pool_f32_vs_torch(case):
    """Max / average (padding included / excluded) with the reference's ceil-mode output shape (pooling.h:92-121)."""
    win, st, pad = case
    rng = np.random.default_rng(8)
    x = rng.standard_normal((2, 5, 14, 14)).astype(np.float32)
    xt = torch.from_numpy(x)
    got = O.pool_f32_nchw(x, win, st, pad, 0)
    want = F.max_pool2d(xt, win, st, pad, ceil_mode=True).numpy()
    assert got.shape == want.shape and np.array_equal(got, want)
    for ptype, incl in ((1, True), (2, False)):
        got = O.pool_f32_nchw(x, win, st, pad, ptype)
        want = F.avg_pool2d(xt, win, st, pad, ceil_mode=True, count_include_pad=incl).numpy()
        assert got.shape == want.shape
        if ptype == 2 or pad == (0, 0):
            # identical semantics: the sum over the in-image window / the number of in-image elements
            assert np.abs(got - want).max() <= 1e-6 * max(1.0, np.abs(want).max()), (case, ptype)
        else:
            # include-padding: torch divides windows that overhang the padded image (ceil mode) by the clipped-to-padded
            # count, as the reference does (saber_pooling.cpp:460-470); interior windows must agree exactly
            inner = (slice(None), slice(None), slice(0, want.shape[2] - 1), slice(0, want.shape[3] - 1))
            assert np.abs(got[inner] - want[inner]).max() <= 1e-6 * max(1.0, np.abs(want).max()), (case, ptype)
    g = O.pool_f32_nchw(x, None, None, None, 1, global_pool=True)
    assert np.abs(g[:, :, 0, 0] - x.mean((2, 3))).max() <= 1e-6


@pytest.mark.parametrize("dt", [np.int8, np.uint8])
def test_pool_i8_vs_integer_arithmetic(dt):
    """INT8 pooling: max is exact; average = round-to-nearest-even of (int32 window sum * (1 / count)) in float, saturated
    (the JIT kernel's vcvtdq2ps / vmulps / vcvtps2dq sequence): recomputed here with numpy integers + float32."""
    rng = np.random.default_rng(9)
    lo, hi = (-128, 128) if dt == np.int8 else (0, 256)
    x = rng.integers(lo, hi, (2, 9, 9, 8)).astype(dt)
    for win, st, pad in (((3, 3), (2, 2), (0, 0)), ((2, 2), (2, 2), (0, 0)), ((3, 3), (2, 2), (1, 1))):
        got = O.pool_i8_nhwc(x, win, st, pad, 0)
        xt = torch.from_numpy(x.astype(np.float32)).permute(0, 3, 1, 2)
        want = F.max_pool2d(xt, win, st, pad, ceil_mode=True).permute(0, 2, 3, 1).numpy().astype(dt)
        assert np.array_equal(got, want), (win, st, pad)
    got = O.pool_i8_nhwc(x, (3, 3), (2, 2), (0, 0), 2)
    oh = got.shape[1]
    want = np.empty_like(got)
    for oy in range(oh):
        for ox in range(oh):
            ys, xs = oy * 2, ox * 2
            w_ = x[:, ys:min(ys + 3, 9), xs:min(xs + 3, 9), :].astype(np.int32)
            cnt = w_.shape[1] * w_.shape[2]
            v = w_.sum((1, 2)).astype(np.float32) * np.float32(1.0 / cnt)
            want[:, oy, ox, :] = np.clip(np.rint(v), lo, hi - 1).astype(dt)
    assert np.array_equal(got, want)
