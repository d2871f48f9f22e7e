"""Why does the fc take ~7.7 us inside the ResNet50 net when the same kernel costs 3.8 us per launch in
timeline_probe? Times the net's fc op (a) via saber_hip_net_time_ops, (b) as a chain of run_op calls between torch
events, (c) a standalone SaberFc of the same shape, each with the small-batch kernel and the implicit-GEMM kernel."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from anakin_amd import lib as L, saber as S, workloads as W

L.require_device()
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
model = W.build_model("resnet50")
scales = W.calibrate(model, W.make_input(2))
net = W.build_int8_net(model, dict(scales), 8)
net.tensor("data").copy_(torch.from_numpy(W.make_input(8)).cuda())
net.run(); torch.cuda.synchronize()
names = [net.op_name(i) for i in range(net.num_ops())]
ifc = [i for i, n in enumerate(names) if n.startswith("fc:")][0]
print("ops", len(names), "fc index", ifc, names[ifc])
us = net.time_ops(iters=40)
print("time_ops: fc %.2f us, pool %.2f, softmax %.2f" % (us[ifc], us[ifc - 1], us[ifc + 1]))

def chain(f, n=200):
    f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1000 / n

print("run_op chain: fc %.2f us" % chain(lambda: net.run_op(ifc)))
print("run_op chain: softmax %.2f us" % chain(lambda: net.run_op(ifc + 1)))
print("run_op chain: pool %.2f us" % chain(lambda: net.run_op(ifc - 1)))
t0 = time.perf_counter()
for _ in range(2000): net.run_op(ifc + 1)
print("host cost of run_op(softmax): %.2f us" % ((time.perf_counter() - t0) * 1e6 / 2000)); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(2000): net.run_op(ifc)
print("host cost of run_op(fc): %.2f us" % ((time.perf_counter() - t0) * 1e6 / 2000)); torch.cuda.synchronize()

rng = np.random.default_rng(0)
M, N, K = 8, 1000, 2048
wq = rng.integers(-127, 128, (N, K)).astype(np.int8); ws = (rng.random(N).astype(np.float32) * 0.01 + 0.001)
b = rng.standard_normal(N).astype(np.float32)
x = torch.from_numpy(rng.integers(-128, 128, (M, K)).astype(np.int8)).cuda()
y = torch.empty((M, N), dtype=torch.float32, device="cuda")
fc = S.SaberFc(True).init(M, N, K, wq, b, L.S8, 0.031, w_scale=ws)
print("standalone", fc.algo(), "%.2f us" % chain(lambda: fc.dispatch(x, y)))
for t in (0 | (4 << 8) | (1 << 16), 0 | (4 << 8) | (4 << 16), 1 | (4 << 8) | (3 << 16)):
    fc.set_tile(t)
    print("standalone", fc.algo(), "%.2f us" % chain(lambda: fc.dispatch(x, y)))
