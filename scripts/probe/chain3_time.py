"""Times the 3x3-led chain at C = 256 (res4: 14 x 14, batch 8 / 1) in its forms - one workgroup per tile with 4 / 8 waves, two
cooperating workgroups per tile - back to back and with a 64 MB L2 flush between launches (the autotuner's cold protocol is in
the library; here: a torch copy of 64 MB between launches and per-launch events). python scripts/probe/chain3_time.py [batch]"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from anakin_amd import saber as S, lib as L

U8, S8 = L.U8, L.S8
rng = np.random.default_rng(0)


def build(batch, Cc=256, HW=14):
    K1 = 4 * Cc
    x = torch.from_numpy(rng.integers(0, 256, (batch, HW, HW, Cc)).astype(np.uint8)).cuda()
    res = torch.from_numpy(rng.integers(-128, 128, (batch, HW, HW, K1)).astype(np.int8)).cuda()
    w0 = (rng.standard_normal((Cc, Cc, 3, 3)) * np.sqrt(2.0 / (9 * Cc))).astype(np.float32)
    w1 = (rng.standard_normal((K1, Cc, 1, 1)) * np.sqrt(2.0 / Cc)).astype(np.float32)
    w2 = (rng.standard_normal((Cc, K1, 1, 1)) * np.sqrt(2.0 / K1)).astype(np.float32)
    c0 = S.SaberConv2D(True).init((batch, Cc, HW, HW), S.ConvParam(w0, None, 1, (1, 1), (1, 1), (1, 1), True), U8, U8, 0.02, 0.03)
    pa = S.ConvParam(w1, None, 1, (0, 0), (1, 1), (1, 1), False)
    pa.res_mode, pa.res_relu, pa.sum_scale, pa.coeff, pa.scale_res = L.RES_ELTWISE, True, 1.0, (16.0, 16.0), 0.043
    ca = S.SaberConv2D(True).init((batch, Cc, HW, HW), pa, U8, S8, 0.03, 0.05)
    cb = S.SaberConv2D(True).init((batch, K1, HW, HW), S.ConvParam(w2, None, 1, (0, 0), (1, 1), (1, 1), True), S8, U8, 0.0625, 0.03)
    return x, res, c0, ca, cb


def timed(fn, flush=None, n=60):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    if flush is None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1000.0 / n
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        flush()
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) * 1000.0 for a, b in ev)
    return t[len(t) // 2]


for batch in ([int(sys.argv[1])] if len(sys.argv) > 1 else [8, 1]):
    x, res, c0, ca, cb = build(batch)
    chain = S.SaberConvChain(ca, cb, conv3x3=c0)
    z1, z2 = ca.new_output(), cb.new_output()
    big_a = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
    big_b = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
    ref = None
    for tn in (1, 3, 7, 15):
        chain.set_tile(tn)
        chain.dispatch(x, res, z1, z2)
        torch.cuda.synchronize()
        if ref is None:
            ref = (z1.clone(), z2.clone())
        ok = torch.equal(z1, ref[0]) and torch.equal(z2, ref[1])
        warm = timed(lambda: chain.dispatch(x, res, z1, z2))
        cold = timed(lambda: chain.dispatch(x, res, z1, z2), flush=lambda: big_b.copy_(big_a))
        print("C=256 14x14 b%d  tile code %d: back to back %6.2f us, cold (event pair, 64 MB flushed) %6.2f us  %s" % (
            batch, tn, warm, cold, "ok" if ok else "MISMATCH"), flush=True)
