"""Where do the ~5 us per pass go that hipGraph replay loses to eager launches (open since round 2; round-4 verdict item 6)?

Hypothesis tested here: the loss is PER GRAPH LAUNCH, not per node - a replay brackets its kernels with the runtime's own packets
(the launch's start / completion markers: a marker packet costs ~2.5 us on this queue, saber_hip_net_time_pass), which an eager stream
of the same 24 kernels does not carry. Test: capture K forward passes into ONE graph (K = 1, 2, 4, 8) and replay it; if the time per
pass is a + b / K, b is the per-launch cost and a the per-pass cost of the nodes themselves, to be compared with eager.
Also: the same under the HIP runtime's graph / kernarg switches (the shell wrapper scripts/probe/graph_vs_eager.sh sets them per
process; this script prints one JSON line).

usage: python scripts/probe/graph_vs_eager.py [--quick]      (GPU box)"""
import json
import os
import pickle
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from anakin_amd import lib as L  # noqa: E402
from anakin_amd import workloads as W  # noqa: E402
import bench  # noqa: E402
import types  # noqa: E402

quick = "--quick" in sys.argv
L.require_device()
stream = torch.cuda.Stream()
torch.cuda.set_stream(stream)
B = 8
model = W.framework_model(W.build_model("resnet50"), "int8")
sc_path = "/tmp/graph_vs_eager_scales.pkl"
if os.path.exists(sc_path):
    scales = pickle.load(open(sc_path, "rb"))
else:
    scales = W.calibrate(model, W.make_input(2))
    pickle.dump(scales, open(sc_path, "wb"))
net = W.build_int8_net(model, dict(scales), B, cxx_optimize=True, stage=True, stem_pair=True)
net.tensor("data").copy_(torch.from_numpy(W.make_input(B)).cuda())
net.run()
torch.cuda.synchronize()
args = types.SimpleNamespace(model="resnet50", precision="int8", graph="framework", no_fuse=False, lanes=False, chain=None, py_fuse=False,
                             no_stage=False, no_stem_pair=False, head_pair=False, tune_cache=os.path.join(ROOT, "profiles", "tune.json"),
                             retune=False, no_autotune=False, write_tune_cache=False)
sel = bench.tune(net, args, B, L, 0, iters=7)


def timed(fn, passes_per_call, calls, reps=5):
    best = []
    for _ in range(reps):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(calls):
            fn()
        torch.cuda.synchronize()
        best.append((time.perf_counter() - t0) * 1e6 / (calls * passes_per_call))
    return min(best), statistics.median(best)


out = {"selection": sel, "launches": net.num_launches(), "env": {k: v for k, v in os.environ.items() if k.startswith(("DEBUG_", "HIP_FORCE", "ROC_", "AMD_"))}}
N = 120 if quick else 400
out["eager_us_per_pass"] = timed(net.run, 1, N)
net.capture()
out["lib_graph_us_per_pass"] = timed(net.replay, 1, N)
if not quick:
    per_k = {}
    for K in (1, 2, 4, 8):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream):
            for _ in range(K):
                net.run()
        per_k[K] = timed(g.replay, K, N // K)
        del g
    out["graph_of_K_passes_us_per_pass"] = per_k
    # least squares for t(K) = a + b / K on the minima
    xs = [1.0 / k for k in per_k]
    ys = [per_k[k][0] for k in per_k]
    n = len(xs)
    mx, my = sum(xs) / n, sum(ys) / n
    b = sum((x - mx) * (y - my) for x, y in zip(xs, ys)) / sum((x - mx) ** 2 for x in xs)
    out["fit"] = {"per_pass_us_a": round(my - b * mx, 2), "per_graph_launch_us_b": round(b, 2)}
    # a graph launch with NO nodes of ours: one empty capture replayed back to back = the launch's own packets
    g0 = torch.cuda.CUDAGraph()
    one = torch.zeros(1, device="cuda")
    with torch.cuda.graph(g0, stream=stream):
        one.add_(1)
    out["one_tiny_kernel_graph_us_per_launch"] = timed(g0.replay, 1, 2000)
    out["one_tiny_kernel_eager_us_per_launch"] = timed(lambda: one.add_(1), 1, 2000)
print(json.dumps(out))
