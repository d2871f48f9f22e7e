"""Same-process A/B of executor-level options on one box: builds the ResNet50 INT8 net under several
build_int8_net flag sets, captures each as a hipGraph and alternates timed replays (medians over rounds)."""
import os, sys, statistics, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from anakin_amd import lib as L, workloads as W  # noqa: F401
from tests import py_fuser as PF      # the step-by-step fusion presets below are the PYTHON fuser (test infrastructure)

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
VARIANTS = {
    "reference op list (no fusion)": dict(fuse_eltwise=False),
    "fused eltwise only": dict(fuse_eltwise=True, pair_siblings=False, fuse_tail=False, fuse_pool=False),
    "+ sibling pairs": dict(fuse_eltwise=True, pair_siblings=True, fuse_tail=False, fuse_pool=False),
    "+ fused tail quantise": dict(fuse_eltwise=True, fuse_pool=False),
    "+ conv1+pool1 (default)": dict(fuse_eltwise=True),
}
L.require_device()
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
model = W.build_model("resnet50")
scales = W.calibrate(model, W.make_input(2))
x = torch.from_numpy(W.make_input(B)).cuda()
nets = {}
for name, kw in VARIANTS.items():
    net = PF.build_int8_net(model, dict(scales), B, **kw)
    net.tensor("data").copy_(x); net.run(); net.autotune(iters=10); net.tensor("data").copy_(x); net.capture()
    for _ in range(20): net.replay()
    nets[name] = net
torch.cuda.synchronize()
res = {k: [] for k in nets}
for r in range(7):
    for name, net in nets.items():
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(200): net.replay()
        torch.cuda.synchronize(); res[name].append((time.perf_counter() - t0) / 200 * 1e3)
for name in nets:
    print("%-34s ops %2d  median %.4f ms  (min %.4f)  %.0f img/s" % (name, nets[name].num_ops(), statistics.median(res[name]), min(res[name]), B / statistics.median(res[name]) * 1e3))
