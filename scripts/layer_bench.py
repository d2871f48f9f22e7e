"""Per-layer micro-benchmark: every (tile, stage-depth) variant of the implicit-GEMM kernel on
representative ResNet50 INT8 layers (batch 8), timed as 20 graph-captured back-to-back launches."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from anakin_amd import lib as L
from anakin_amd import saber as S

B = int(os.environ.get("B", "8"))
REP = 20
LAYERS = [
    # name, cin, hin, cout, k, stride, pad, in_dt, relu, eltwise
    ("conv1_7x7", 3, 224, 64, 7, 2, 3, L.F32, True, False),
    ("res2_1x1_64_64", 64, 56, 64, 1, 1, 0, L.U8, True, False),
    ("res2_3x3_64_64", 64, 56, 64, 3, 1, 1, L.U8, True, False),
    ("res2_1x1_64_256_elt", 64, 56, 256, 1, 1, 0, L.U8, False, True),
    ("res2_1x1_64_256", 64, 56, 256, 1, 1, 0, L.U8, False, False),
    ("res2_1x1_256_64", 256, 56, 64, 1, 1, 0, L.S8, True, False),
    ("res3_3x3_128", 128, 28, 128, 3, 1, 1, L.U8, True, False),
    ("res3_1x1_128_512_elt", 128, 28, 512, 1, 1, 0, L.U8, False, True),
    ("res4_1x1_1024_256", 1024, 14, 256, 1, 1, 0, L.S8, True, False),
    ("res4_3x3_256", 256, 14, 256, 3, 1, 1, L.U8, True, False),
    ("res4_1x1_256_1024_elt", 256, 14, 1024, 1, 1, 0, L.U8, False, True),
    ("res5_3x3_512", 512, 7, 512, 3, 1, 1, L.U8, True, False),
    ("res5_1x1_512_2048_elt", 512, 7, 2048, 1, 1, 0, L.U8, False, True),
    ("res5_1x1_2048_512", 2048, 7, 512, 1, 1, 0, L.S8, True, False),
]
only = sys.argv[1:] if len(sys.argv) > 1 else None
torch.cuda.set_stream(torch.cuda.Stream())
rng = np.random.default_rng(0)
for (name, cin, hin, cout, k, stride, pad, in_dt, relu, elt) in LAYERS:
    if only and not any(o in name for o in only):
        continue
    w = (rng.standard_normal((cout, cin, k, k)) * 0.05).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    p = S.ConvParam(w, b, 1, (pad, pad), (stride, stride), (1, 1), relu)
    odt = L.U8 if relu else L.S8
    if elt:
        p.res_mode, p.res_relu, p.coeff, p.scale_res = L.RES_ELTWISE, True, (20.0, 20.0), 0.04
    conv = S.SaberConv2D(True).init((B, cin, hin, hin), p, in_dt, odt, 0.02, 0.05,
                                    in_layout=L.NCHW if in_dt == L.F32 else L.NHWC)
    ho = conv.out_hw[0]
    net = S.Net()
    net.add_tensor("x", (B, cin, hin, hin), in_dt)
    net.add_tensor("y", (B, ho, ho, cout), odt)
    net.add_tensor("r", (B, ho, ho, cout), L.S8)
    for _ in range(REP):
        net.add_conv(conv, "x", "y", "r" if elt else None)
    net.finalize()
    if in_dt == L.F32:
        net.tensor("x").copy_(torch.rand((B, cin, hin, hin), device="cuda") * 2 - 1)
    else:
        net.tensor("x").copy_(torch.randint(0, 127, (B, cin, hin, hin), device="cuda").to(net.tensor("x").dtype))
    net.tensor("r").copy_(torch.randint(-100, 100, (B, ho, ho, cout), device="cuda").to(torch.int8))
    M = B * ho * ho
    gmac = M * cout * cin * k * k / 1e9
    mbytes = (B * hin * hin * cin + M * cout * (2 if elt else 1) + cin * cout * k * k) / 1e6
    res = {}
    for var in ((1,) if cin < 16 else (1, 2, 3, 4)):
      for tile in range(6):
        for ks in (1, 2, 4):
            if var >= 3 and (ks != 4 or tile > 2 or (var == 4 and tile != 0)):
                continue
            conv.set_tile(tile | (ks << 8) | (var << 16))
            net.capture()
            net.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                net.replay()
            e1.record()
            torch.cuda.synchronize()
            res[(L.TILES[tile] + ("r", "d", "d2", "d4")[var - 1], ks)] = e0.elapsed_time(e1) * 1000 / (3 * REP)
    if k == 7 and stride == 2 and cin <= 4:
        conv.set_tile(7 << 16)
        net.capture()
        net.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            net.replay()
        e1.record()
        torch.cuda.synchronize()
        res[("stem", 0)] = e0.elapsed_time(e1) * 1000 / (3 * REP)
    if k == 3 and stride == 1 and cin % 64 == 0:
        for var, nm in ((5, "halo4"), (6, "halo8")):
            conv.set_tile(var << 16)
            net.capture()
            net.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                net.replay()
            e1.record()
            torch.cuda.synchronize()
            res[(nm, 0)] = e0.elapsed_time(e1) * 1000 / (3 * REP)
    best = min(res, key=res.get)
    print("%-24s M=%6d K=%4d Kg=%5d  %.3f GMAC %.2f MB | best %s k%d %.2f us (%.0f GB/s, %.1f TOPS)" % (
        name, M, cout, cin * k * k, gmac, mbytes, best[0], best[1], res[best], mbytes / res[best] * 1e3,
        2 * gmac / res[best] * 1e3))
    print("    " + "  ".join("%s/k%d=%.1f" % (t, ks, v) for (t, ks), v in sorted(res.items(), key=lambda kv: kv[1])[:8]))
