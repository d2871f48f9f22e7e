"""The XCD-resident stage (saber_hip_stage_*, stage_xcd.hip): ResNet res5 - nine convolutions on 7x7 images - as ONE
persistent launch, image i on XCD i % 8, phases separated by an XCD-local barrier. Every edge must hold the bits of
dispatching the same ops one after the other (each of which the parity tests pin to the oracle), for batches that leave
XCDs idle (1, 2), fill them once (8) and wrap around (9, 16), repeatedly (the barrier counters run on across launches)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from anakin_amd import lib as L  # noqa: E402
from anakin_amd import saber as S  # noqa: E402
from oracle import oracle as O  # noqa: E402


@pytest.fixture(scope="module", autouse=True)
def _device():
    L.require_device()


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    torch.cuda.synchronize()
    return t.cpu().numpy()


def conv(rng, N, H, W, cin, cout, k, relu, idt, odt, s_in, s_out, elt=None):
    w = (rng.standard_normal((cout, cin, k, k)) * np.sqrt(2.0 / (k * k * cin))).astype(np.float32)
    b = (rng.standard_normal(cout) * 0.5).astype(np.float32)
    p = S.ConvParam(w, b, 1, (k // 2, k // 2), (1, 1), (1, 1), bool(relu))
    if elt is not None:
        s_res, s_sum = elt
        c = 1.0 / s_sum
        p.res_mode, p.res_relu, p.sum_scale, p.coeff, p.scale_res = L.RES_ELTWISE, True, 1.0, (c, c), s_res
    return S.SaberConv2D(int8=True).init((N, cin, H, W), p, idt, odt, s_in, s_out)


def res5(rng, N, H=7, W=7, blocks=3):
    """[(conv, in, out, res)] of res5a (projection shortcut) + identity blocks, tensor slot 0 = the stage's input"""
    ph, t = [], 1
    s = {0: 0.05}

    def add(cin, cout, k, relu, src, odt, s_out, res=None):
        nonlocal t
        idt = O.U8 if src in u8 else O.S8
        c = conv(rng, N, H, W, cin, cout, k, relu, idt, odt, s[src], s_out, None if res is None else (s[res], s_out))
        ph.append((c, src, t, -1 if res is None else res))
        s[t] = s_out
        if odt == O.U8:
            u8.add(t)
        t += 1
        return t - 1
    u8 = set()
    short = add(1024, 2048, 1, False, 0, O.S8, 0.07)
    x = 0
    for b in range(blocks):
        a = add(1024 if b == 0 else 2048, 512, 1, True, x, O.U8, 0.03)
        m = add(512, 512, 3, True, a, O.U8, 0.025)
        x = add(512, 2048, 1, False, m, O.S8, 0.06 + 0.01 * b, res=short)
        short = x
    return ph, t


@pytest.mark.parametrize("N", [1, 2, 8, 9, 16])
def test_stage_res5_equals_the_ops_one_by_one(N):
    rng = np.random.default_rng(900 + N)
    ph, nt = res5(rng, N)
    x0 = rng.integers(-128, 128, (N, 7, 7, 1024)).astype(np.int8)
    want = [dev(x0)] + [None] * (nt - 1)
    for c, i, o, r in ph:
        want[o] = c.new_output()
        c.dispatch(want[i], want[o], None if r < 0 else want[r])
    ref = [host(t) for t in want]
    stage = S.SaberStage(ph)
    for rep in range(3):                      # the counters run on from launch to launch
        got = [dev(x0)] + [torch.full_like(t, 77) for t in want[1:]]
        stage.dispatch(got)
        stage.status()
        for k in range(1, nt):
            assert np.array_equal(host(got[k]), ref[k]), ("slot", k, "launch", rep)


def test_stage_rejects_what_it_has_no_kernel_for():
    rng = np.random.default_rng(5)
    c = conv(rng, 1, 7, 7, 256, 256, 3, True, O.U8, O.U8, 0.03, 0.03)
    with pytest.raises(RuntimeError):
        S.SaberStage([(c, 0, 1, -1)])                       # 3x3 256 -> 256: no variant
    big = conv(rng, 1, 14, 14, 1024, 512, 1, True, O.S8, O.U8, 0.03, 0.03)
    with pytest.raises(RuntimeError):
        S.SaberStage([(big, 0, 1, -1)])                     # 196 pixels per image
    ok = conv(rng, 1, 7, 7, 1024, 512, 1, True, O.S8, O.U8, 0.03, 0.03)
    with pytest.raises(RuntimeError):
        S.SaberStage([(ok, 0, 0, -1)])                      # in place


IMG_CASES = [
    # cin, cout, k, relu, in dtype, out dtype, eltwise
    (1024, 2048, 1, False, O.S8, O.S8, False),
    (1024, 512, 1, True, O.S8, O.U8, False),
    (512, 512, 3, True, O.U8, O.U8, False),
    (512, 2048, 1, False, O.U8, O.S8, True),
    (2048, 512, 1, True, O.S8, O.U8, False),
]


@pytest.mark.parametrize("case", IMG_CASES)
@pytest.mark.parametrize("N,H,W", [(8, 7, 7), (3, 8, 8), (1, 5, 6)])
def test_image_resident_conv_kernel_equals_oracle_and_default_kernel(case, N, H, W):
    """Kernel variant 12 of saber_hip_conv2d_run (one workgroup = one image x a channel group, the phase code of the stage kernel
    as an ordinary launch): the oracle's bytes and the default kernel's, every res5 shape, ragged images, fewer images than XCDs."""
    cin, cout, k, relu, idt, odt, elt = case
    rng = np.random.default_rng(cin + cout + N)
    s_in, s_out, s_res = 0.03, 0.05, 0.04
    c = conv(rng, N, H, W, cin, cout, k, relu, idt, odt, s_in, s_out, (s_res, s_out) if elt else None)
    x = (rng.integers(0, 256, (N, H, W, cin)).astype(np.uint8) if idt == O.U8 else rng.integers(-128, 128, (N, H, W, cin)).astype(np.int8))
    res = rng.integers(-128, 128, (N, H, W, cout)).astype(np.int8)
    y0 = c.new_output()
    c.dispatch(dev(x), y0, dev(res) if elt else None)
    want = host(y0)
    c.set_tile(12 << 16)
    assert c.algo().startswith("imgres"), c.algo()
    y1 = torch.full_like(y0, 77)
    c.dispatch(dev(x), y1, dev(res) if elt else None)
    assert np.array_equal(host(y1), want), c.algo()
    # fused global average pooling == the INT8 pooling op on the conv's output
    pool = S.pooling_i8(y0, (H, W), (1, 1), (0, 0), 1, global_pooling=True)
    c.set_global_pooling()
    assert c.algo().endswith("+gpool")
    y2, yp = torch.full_like(y0, 55), torch.zeros((N, 1, 1, cout), dtype=y0.dtype, device="cuda")
    c.dispatch_gpool(dev(x), y2, yp, dev(res) if elt else None)
    assert np.array_equal(host(y2), want)
    want_pool = O.pool_i8_nhwc(want, None, None, None, 1, out_dtype=None, global_pool=True)
    assert np.array_equal(host(yp).reshape(want_pool.shape), want_pool), c.algo()
    assert np.array_equal(host(pool).reshape(want_pool.shape), want_pool)
