import sys, time
sys.path.insert(0, '/root/repo')
import torch
from anakin_amd import lib as L, workloads as W
L.require_device()
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
model = W.build_model("resnet50"); scales = W.calibrate(model, W.make_input(2))
for B in (8, 1):
    net = W.build_int8_net(model, dict(scales), B)
    net.tensor("data").copy_(torch.from_numpy(W.make_input(B)).cuda()); net.run(); net.autotune(iters=5); net.capture()
    for mode in ("eager", "graph"):
        f = net.run if mode == "eager" else net.replay
        for _ in range(20): f()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(300): f()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("batch %d %s: enqueue %.1f us/step, total %.1f us/step" % (B, mode, (t1 - t0) / 300 * 1e6, (t2 - t0) / 300 * 1e6))
