"""Op-list capture (saber_hip_capture_begin / _end, include/saber_hip.h): the mechanism that puts the executor's fused launches
behind the reference's Net<MI355X>::prediction() (integration/mi355x/framework/mi355x_net_plan.h; tests/test_gpu_net.py runs it
inside the reference's framework). Here the same thing at the C ABI: a ResNet bottleneck block dispatched operator by
operator on THREE buffers the way the reference's memory planner aliases edges (the block input's buffer receives the
3x3 conv's output, then the block's result), once for real and once under capture. The captured list - tensors renamed out of
the aliasing, fused by saber_hip_net_optimize - must reproduce the bytes of the operator-by-operator pass and of the oracle."""
import ctypes as C

import numpy as np
import pytest
import torch

from anakin_amd import lib as L
from anakin_amd import saber as S
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _block(n, hw, c, seed):
    """entry block of a stage: shortcut projection d (1x1, c -> 4c, s8) | a (1x1, c -> c, relu, u8) -> b (3x3, relu, u8) -> cc (1x1, c -> 4c, s8)
    -> eltwise(cc, d) + relu -> s8; then the next block's 1x1 (4c -> c, relu, u8)."""
    rng = np.random.default_rng(seed)
    sc = dict(x=0.02, d=0.05, a=0.03, b=0.04, cc=0.06, e=0.07, nxt=0.05)

    def conv(cin, cout, k, relu, si, so, in_dt, out_dt):
        w = (rng.standard_normal((cout, cin, k, k)) * (1.5 / np.sqrt(cin * k * k))).astype(np.float32)
        b = (rng.standard_normal(cout) * 0.2).astype(np.float32)
        p = S.ConvParam(w, b, 1, (k // 2,) * 2, (1, 1), (1, 1), relu)
        op = S.SaberConv2D(True).init((n, cin, hw, hw), p, in_dt, out_dt, si, so)
        return op, (w, b)
    ops = {}
    ops["d"] = conv(c, 4 * c, 1, False, sc["x"], sc["d"], L.U8, L.S8)
    ops["a"] = conv(c, c, 1, True, sc["x"], sc["a"], L.U8, L.U8)
    ops["b"] = conv(c, c, 3, True, sc["a"], sc["b"], L.U8, L.U8)
    ops["cc"] = conv(c, 4 * c, 1, False, sc["b"], sc["cc"], L.U8, L.S8)
    ops["nxt"] = conv(4 * c, c, 1, True, sc["e"], sc["nxt"], L.S8, L.U8)
    x = rng.integers(0, 256, (n, hw, hw, c)).astype(np.uint8)
    return ops, sc, x


def _oracle(ops, sc, x):
    def run(name, xin, si, so, in_dt, out_dt, relu):
        w, b = ops[name][1]
        ws = O.weight_scales(w)
        bp, s = O.conv_i8_prepare(ws, b, si, so, in_dt, out_dt)
        return O.conv_i8(xin, O.quant_weights(w, ws), bp, s, out_dt, 1 if relu else 0, (w.shape[2] // 2,) * 2)
    d = run("d", x, sc["x"], sc["d"], O.U8, O.S8, False)
    a = run("a", x, sc["x"], sc["a"], O.U8, O.U8, True)
    b = run("b", a, sc["a"], sc["b"], O.U8, O.U8, True)
    cc = run("cc", b, sc["b"], sc["cc"], O.U8, O.S8, False)
    e = O.eltwise_i8(cc, d, sc["cc"], sc["d"], 1.0 / sc["e"], 1.0 / sc["e"], True)
    nxt = run("nxt", e, sc["e"], sc["nxt"], O.S8, O.U8, True)
    return e, nxt


def _dispatch_aliased(ops, sc, bufs, n, hw, c):
    """the op loop on three aliased buffers: x lives in buf0 and is overwritten twice"""
    b0, b1, b2 = bufs
    px, pc = n * hw * hw * c, n * hw * hw * 4 * c

    def v(buf, count, dt):
        return buf[:count].view(dt)
    ops["d"][0].dispatch(v(b0, px, torch.uint8), v(b1, pc, torch.int8))
    ops["a"][0].dispatch(v(b0, px, torch.uint8), v(b2, px, torch.uint8))
    ops["b"][0].dispatch(v(b2, px, torch.uint8), v(b0, px, torch.uint8))            # x is dead: its buffer is reused
    ops["cc"][0].dispatch(v(b0, px, torch.uint8), v(b2, pc, torch.int8))
    lib = L.load()
    L.check(lib.saber_hip_eltwise_sum_i8(pc, b2.data_ptr(), b1.data_ptr(), sc["cc"], sc["d"], 1.0 / sc["e"], 1.0 / sc["e"], 1,
                                         b0.data_ptr(), torch.cuda.current_stream().cuda_stream))
    ops["nxt"][0].dispatch(v(b0, pc, torch.int8), v(b1, px, torch.uint8))


@pytest.mark.parametrize("n,hw,c", [(2, 14, 64), (3, 28, 128), (2, 7, 256)])
def test_captured_block_equals_op_loop_and_oracle(n, hw, c):
    L.require_device()
    ops, sc, x = _block(n, hw, c, seed=n * 100 + hw)
    want_e, want_nxt = _oracle(ops, sc, x)
    px, pc = n * hw * hw * c, n * hw * hw * 4 * c
    bufs = [torch.zeros(pc + 4096, dtype=torch.uint8, device="cuda") for _ in range(3)]
    xt = torch.from_numpy(x.ravel()).cuda()

    bufs[0][:px].copy_(xt)
    _dispatch_aliased(ops, sc, bufs, n, hw, c)                       # operator by operator
    torch.cuda.synchronize()
    loop_e = bufs[0][:pc].view(torch.int8).cpu().numpy().reshape(want_e.shape)
    loop_nxt = bufs[1][:px].cpu().numpy().reshape(want_nxt.shape)
    assert np.array_equal(loop_e, want_e) and np.array_equal(loop_nxt, want_nxt)

    for b in bufs:
        b.zero_()
    with S.Capture(keep=[o[0] for o in ops.values()]) as cap:        # the same calls, recorded
        assert L.load().saber_hip_capture_active() == 1
        _dispatch_aliased(ops, sc, bufs, n, hw, c)
    assert L.load().saber_hip_capture_active() == 0
    torch.cuda.synchronize()
    assert int(bufs[1].count_nonzero()) == 0 and int(bufs[2].count_nonzero()) == 0      # nothing was launched
    net = cap.net
    assert net.num_ops() == 6 and net.num_tensors() == 7             # x + one tensor per write: the aliasing is gone
    e_id, nxt_id = net.tensor_of_ptr(bufs[0]), net.tensor_of_ptr(bufs[1])
    assert e_id >= 0 and nxt_id >= 0 and e_id != nxt_id
    net.bind(e_id, bufs[0])                                          # the outputs stay where the caller's loop leaves them
    net.bind(nxt_id, bufs[1])
    removed = net.optimize(255)
    assert removed >= 2                                              # the eltwise became an epilogue, d + a one sibling pair ...
    net.finalize()
    for it in range(2):                                              # the input's buffer is also the output's: feed, run, check
        bufs[0][:px].copy_(xt)
        net.run()
        torch.cuda.synchronize()
        got_e = bufs[0][:pc].view(torch.int8).cpu().numpy().reshape(want_e.shape)
        got_nxt = bufs[1][:px].cpu().numpy().reshape(want_nxt.shape)
        assert np.array_equal(got_e, want_e), (it, [net.op_name(i) for i in range(net.num_ops())])
        assert np.array_equal(got_nxt, want_nxt), it
    assert net.num_launches() <= 4
    net.autotune(3)
    bufs[0][:px].copy_(xt)
    net.capture()
    net.replay()
    torch.cuda.synchronize()
    assert np.array_equal(bufs[1][:px].cpu().numpy().reshape(want_nxt.shape), want_nxt)


def test_capture_refuses_what_it_cannot_express():
    """an entry point without an op-list form fails the CAPTURE (end returns SABER_HIP_UNIMPL) while the call itself
    returns success - the reference's SABER_CHECK around every dispatch is fatal; overlapping views likewise"""
    L.require_device()
    lib = L.load()
    a = torch.zeros(64 * 64, dtype=torch.float32, device="cuda")
    L.check(lib.saber_hip_capture_begin())
    assert lib.saber_hip_capture_begin() == -2                       # one capture per thread
    assert lib.saber_hip_gemm_f32(0, 0, 64, 64, 64, 1.0, a.data_ptr(), a.data_ptr(), 0.0, a.data_ptr(), None) == 0
    h = C.c_void_p()
    assert lib.saber_hip_capture_end(C.byref(h)) == L.UNIMPL and not h
    assert b"gemm" in lib.saber_hip_last_error()
    assert lib.saber_hip_capture_end(C.byref(h)) == -2               # closed
    # a read that straddles a tensor written at another base address
    buf = torch.zeros(4096, dtype=torch.float32, device="cuda")
    L.check(lib.saber_hip_capture_begin())
    s = torch.cuda.current_stream().cuda_stream
    assert lib.saber_hip_relu_f32(1024, buf.data_ptr(), buf.data_ptr() + 8192, s) == 0
    assert lib.saber_hip_relu_f32(1024, buf.data_ptr() + 8192 + 1024, buf.data_ptr(), s) == 0
    assert lib.saber_hip_capture_end(C.byref(h)) == L.UNIMPL
    assert b"overlap" in lib.saber_hip_last_error()


def test_captured_fp32_inplace_sum_keeps_one_tensor():
    """the FP32 ConvEltwise post-op (RES_SUM_INPLACE) reads and writes the shortcut's buffer: one tensor, not a new version"""
    L.require_device()
    rng = np.random.default_rng(7)
    n, hw, c, k = 2, 14, 32, 64
    w = (rng.standard_normal((k, c, 1, 1)) * 0.1).astype(np.float32)
    b = (rng.standard_normal(k) * 0.1).astype(np.float32)
    p = S.ConvParam(w, b, 1, (0, 0), (1, 1), (1, 1), False)
    p.res_mode, p.res_relu = L.RES_SUM_INPLACE, True
    conv = S.SaberConv2D(False).init((n, c, hw, hw), p, L.F32, L.F32, in_layout=L.NHWC, out_layout=L.NHWC)
    x = torch.from_numpy(rng.standard_normal((n, hw, hw, c)).astype(np.float32)).cuda()
    y0 = torch.from_numpy(rng.standard_normal((n, hw, hw, k)).astype(np.float32)).cuda()
    y = y0.clone()
    conv.dispatch(x, y)
    torch.cuda.synchronize()
    want = y.cpu().numpy()
    y.copy_(y0)
    with S.Capture(keep=[conv]) as cap:
        conv.dispatch(x, y)
    net = cap.net
    assert net.num_ops() == 1 and net.num_tensors() == 2             # x and y, both the caller's
    net.finalize()
    net.run()
    torch.cuda.synchronize()
    assert np.array_equal(y.cpu().numpy(), want)
    # the in-place target is the caller's tensor, never written by the list itself: every run accumulates into it, so the autotuner
    # (which runs passes on its own) refuses the net instead of letting its result drift (round-4 advisor finding)
    with pytest.raises(L.SaberHipError):
        net.autotune(iters=3)
