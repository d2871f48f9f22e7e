"""Probe: batch 8 as k concurrent micro-batch op lists on k streams (each its own hipGraph); with `python
scripts/microbatch_probe.py streams` instead: k independent batch-8 op lists in flight on k streams (serving throughput)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from anakin_amd import workloads as W

model = W.build_model("resnet50")
scales = W.calibrate(model, W.make_input(2))
STREAMS = len(sys.argv) > 1 and sys.argv[1] == "streams"
for k in ((1, 2, 3) if STREAMS else (1, 2, 4)):
    b = 8 if STREAMS else 8 // k
    nets, streams = [], []
    for i in range(k):
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            net = W.build_int8_net(model, dict(scales), b)
            net.tensor("data").copy_(torch.from_numpy(W.make_input(b, seed=i)).cuda())
            net.run(); net.autotune(iters=10); net.capture()
        nets.append(net); streams.append(s)
    torch.cuda.synchronize()
    def step():
        for net, s in zip(nets, streams):
            with torch.cuda.stream(s):
                net.replay()
    for _ in range(20): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 300
    for _ in range(n): step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    imgs = k * b
    print("%s=%d x batch %d: %.4f ms per %d images -> %.0f images/s" % ("streams" if STREAMS else "micro-batches", k, b, dt * 1e3, imgs, imgs / dt))
