import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from anakin_amd import lib as L, saber as S
from oracle import oracle as O
rng = np.random.default_rng(0)
for (N,C,H,W,K,k,pad) in [(1,4,4,4,16,1,0),(1,16,4,4,16,1,0),(1,32,6,6,16,3,1)]:
    x = rng.integers(-3,4,(N,C,H,W)).astype(np.float32)
    w = rng.integers(-2,3,(K,C,k,k)).astype(np.float32)
    want = O.conv_f32_nchw(x,w,None,False,(pad,pad))
    p = S.ConvParam(w,None,1,(pad,pad),(1,1),(1,1),False)
    conv = S.SaberConv2D(False).init(x.shape,p,L.F32,L.F32)
    y = conv.new_output(); conv.dispatch(torch.from_numpy(x).cuda(), y); torch.cuda.synchronize()
    got = y.cpu().numpy()
    print(conv.algo(), "maxerr", np.abs(got-want).max())
    if np.abs(got-want).max()>0:
        print("want[0,:4,0,:4]\n", want[0,:4,0,:4], "\ngot\n", got[0,:4,0,:4])
        # try to explain: is got == conv with permuted channels?
        for perm_name, perm in [("id", np.arange(C))]:
            pass
