"""Times a run of 3x3-led chains (res3: C = 128, 28 x 28, 3 blocks; res4: C = 256, 14 x 14, 5 blocks) block by block - every chain in
its tile codes - against the one persistent stage launch (saber_hip_conv2d_stage_create), back to back and with a 64 MB flush
between repetitions. python scripts/probe/chain_stage_time.py [C] [batch] [blocks]"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from anakin_amd import saber as S, lib as L

U8, S8 = L.U8, L.S8
Cc = int(sys.argv[1]) if len(sys.argv) > 1 else 128
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 8
nblk = int(sys.argv[3]) if len(sys.argv) > 3 else (3 if Cc == 128 else 5)
HW = 28 if Cc == 128 else 14
K1 = 4 * Cc
rng = np.random.default_rng(0)
x = torch.from_numpy(rng.integers(0, 256, (batch, HW, HW, Cc)).astype(np.uint8)).cuda()
res = torch.from_numpy(rng.integers(-128, 128, (batch, HW, HW, K1)).astype(np.int8)).cuda()
ops = []
for k in range(nblk):
    w0 = (rng.standard_normal((Cc, Cc, 3, 3)) * np.sqrt(2.0 / (9 * Cc))).astype(np.float32)
    w1 = (rng.standard_normal((K1, Cc, 1, 1)) * np.sqrt(2.0 / Cc)).astype(np.float32)
    w2 = (rng.standard_normal((Cc, K1, 1, 1)) * np.sqrt(2.0 / K1)).astype(np.float32)
    c0 = S.SaberConv2D(True).init((batch, Cc, HW, HW), S.ConvParam(w0, None, 1, (1, 1), (1, 1), (1, 1), True), U8, U8, 0.02, 0.03)
    pa = S.ConvParam(w1, None, 1, (0, 0), (1, 1), (1, 1), False)
    pa.res_mode, pa.res_relu, pa.sum_scale, pa.coeff, pa.scale_res = L.RES_ELTWISE, True, 1.0, (16.0, 16.0), 0.043
    ca = S.SaberConv2D(True).init((batch, Cc, HW, HW), pa, U8, S8, 0.03, 0.05)
    cb = S.SaberConv2D(True).init((batch, K1, HW, HW), S.ConvParam(w2, None, 1, (0, 0), (1, 1), (1, 1), True), S8, U8, 0.0625, 0.03)
    ops.append((c0, ca, cb))
chains = [S.SaberConvChain(ca, cb, conv3x3=c0) for c0, ca, cb in ops]
y1 = [ca.new_output() for _, ca, _ in ops]
y2 = [cb.new_output() for _, _, cb in ops]
big_a = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
big_b = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")


def run_chains():
    cx, cr = x, res
    for k, ch in enumerate(chains):
        ch.dispatch(cx, cr, y1[k], y2[k])
        cx, cr = y2[k], y1[k]


def timed(fn, cold, n=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    if not cold:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1000.0 / n
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        big_b.copy_(big_a)
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) * 1000.0 for a, b in ev)
    return t[len(t) // 2]


codes = (1, 2, 5, 6) if Cc == 128 else (1, 3, 7, 15)
for tn in codes:
    try:
        for ch in chains:
            ch.set_tile(tn)
    except L.SaberHipError:
        continue
    print("C=%d %dx%d b%d  %d chain launches, tile code %2d: back to back %7.2f us, cold %7.2f us"
          % (Cc, HW, HW, batch, nblk, tn, timed(run_chains, False), timed(run_chains, True)))
ref1, ref2 = [t.clone() for t in y1], [t.clone() for t in y2]
stage = S.SaberChainStage(chains)
fn = lambda: stage.dispatch(x, res, y1, y2)      # noqa: E731
fn()
torch.cuda.synchronize()
same = all(torch.equal(a, b) for a, b in zip(ref1 + ref2, y1 + y2))
print("C=%d %dx%d b%d  ONE stage launch of %d blocks:        back to back %7.2f us, cold %7.2f us   %s"
      % (Cc, HW, HW, batch, nblk, timed(fn, False), timed(fn, True), "bit-identical to the chains" if same else "MISMATCH"))
