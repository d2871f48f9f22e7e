"""Times the 1x1 chain kernel against the two launches it replaces (autotuned), per ResNet stage shape.
Run on the GPU box: python scripts/probe/chain_time.py [batch]"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from anakin_amd import saber as S, lib as L

U8, S8 = L.U8, L.S8
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rng = np.random.default_rng(0)


def timed(fn, n=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000.0 / n


for Cc, HW in ((64, 56), (128, 28), (256, 14), (512, 7)):
    K1 = 4 * Cc
    x = torch.from_numpy(rng.integers(0, 256, (batch, HW, HW, Cc)).astype(np.uint8)).cuda()
    res = torch.from_numpy(rng.integers(-128, 128, (batch, HW, HW, K1)).astype(np.int8)).cuda()
    w1 = (rng.standard_normal((K1, Cc, 1, 1)) * np.sqrt(2.0 / Cc)).astype(np.float32)
    w2 = (rng.standard_normal((Cc, K1, 1, 1)) * np.sqrt(2.0 / K1)).astype(np.float32)
    pa = S.ConvParam(w1, None, 1, (0, 0), (1, 1), (1, 1), False)
    pa.res_mode, pa.res_relu, pa.sum_scale, pa.coeff, pa.scale_res = L.RES_ELTWISE, True, 1.0, (16.0, 16.0), 0.043
    ca = S.SaberConv2D(int8=True).init((batch, Cc, HW, HW), pa, U8, S8, 0.02, 0.05)
    cb = S.SaberConv2D(int8=True).init((batch, K1, HW, HW), S.ConvParam(w2, None, 1, (0, 0), (1, 1), (1, 1), True), S8, U8,
                                       0.0625, 0.03)
    y1, y2 = ca.new_output(), cb.new_output()
    ca.autotune(x, y1, res)
    cb.autotune(y1, y2)
    chain = S.SaberConvChain(ca, cb)
    z1, z2 = ca.new_output(), cb.new_output()

    def two():
        ca.dispatch(x, y1, res)
        cb.dispatch(y1, y2)

    t2 = timed(two)
    line = "C=%3d %2dx%-2d b%d  two launches %6.2f us (%s | %s)" % (Cc, HW, HW, batch, t2, ca.algo(), cb.algo())
    tiles = {64: (4, 2), 128: (2, 1)}.get(Cc, (1,))
    for tn in tiles:
        chain.set_tile(tn)
        chain.dispatch(x, res, z1, z2)
        ok = torch.equal(z1, y1) and torch.equal(z2, y2)
        line += "   chain tn=%d %6.2f us %s" % (tn, timed(lambda: chain.dispatch(x, res, z1, z2)), "ok" if ok else "MISMATCH")
    print(line, flush=True)
