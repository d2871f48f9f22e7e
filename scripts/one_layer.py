"""Run ONE conv layer config N times (for rocprofv3 --pmc passes). env: LAYER, TILE, KS, B, N"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from anakin_amd import lib as L
from anakin_amd import saber as S

B = int(os.environ.get("B", "8"))
cin, hin, cout, k, stride, pad, in_dt, relu, elt = eval(os.environ.get("LAYER", "(64,56,256,1,1,0,2,False,False)"))
tile, ks, n = int(os.environ.get("TILE", "2")), int(os.environ.get("KS", "1")), int(os.environ.get("N", "5"))
rng = np.random.default_rng(0)
w = (rng.standard_normal((cout, cin, k, k)) * 0.05).astype(np.float32)
b = rng.standard_normal(cout).astype(np.float32)
p = S.ConvParam(w, b, 1, (pad, pad), (stride, stride), (1, 1), relu)
odt = L.U8 if relu else L.S8
if elt:
    p.res_mode, p.res_relu, p.coeff, p.scale_res = L.RES_ELTWISE, True, (20.0, 20.0), 0.04
conv = S.SaberConv2D(True).init((B, cin, hin, hin), p, in_dt, odt, 0.02, 0.05)  # in_dt s8/u8: NHWC input
conv.set_tile(tile | (ks << 8))
x = torch.randint(0, 127, (B, hin, hin, cin), device="cuda").to(torch.uint8 if in_dt == L.U8 else torch.int8)
y = conv.new_output()
r = torch.randint(-100, 100, tuple(y.shape), device="cuda").to(torch.int8)
for _ in range(n):
    conv.dispatch(x, y, r if elt else None)
torch.cuda.synchronize()
print(conv.algo())
