"""Do a Worker request's pieces overlap ACROSS requests on this GPU? (a) a 4.8 MB host-to-device copy loop and a forward-pass loop, each
alone and both at once from two host threads on two streams; (b) 1 / 2 / 3 host threads each running whole requests (copy -> pass ->
32 KB copy back, synchronising after each like Worker::sync_prediction) with shared-device nets, eager launches and hipGraph replays.
usage: python scripts/probe/overlap_probe.py   (GPU box)"""
import json
import os
import pickle
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from anakin_amd import lib as L  # noqa: E402
from anakin_amd import workloads as W  # noqa: E402

L.require_device()
B = 8
model = W.framework_model(W.build_model("resnet50"), "int8")
sc_path = "/tmp/graph_vs_eager_scales.pkl"
scales = pickle.load(open(sc_path, "rb")) if os.path.exists(sc_path) else W.calibrate(model, W.make_input(2))
NT = 3
nets, streams, cstreams, hin, hout, din = [], [], [], [], [], []
choices = None
for i in range(NT):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        n = W.build_int8_net(model, dict(scales), B, cxx_optimize=True, shared_device=True)
        n.tensor("data").copy_(torch.from_numpy(W.make_input(B)).cuda())
        n.run()
        if choices is None:
            n.autotune(iters=5)
            choices = n.choices()
        else:
            n.set_choices(choices)
        n.run()
        n.capture()
    nets.append(n); streams.append(st); cstreams.append(torch.cuda.Stream())
    hin.append(torch.from_numpy(W.make_input(B)).pin_memory())
    hout.append(torch.empty(B, 1000).pin_memory())
torch.cuda.synchronize()
out = {}


def loop_copy(i, secs, res):
    n = 0
    t0 = time.perf_counter()
    with torch.cuda.stream(cstreams[i]):
        while time.perf_counter() - t0 < secs:
            nets[i].tensor("data").copy_(hin[i], non_blocking=True)
            cstreams[i].synchronize()
            n += 1
    res[i] = n / (time.perf_counter() - t0)


def loop_pass(i, secs, res, graph):
    n = 0
    t0 = time.perf_counter()
    with torch.cuda.stream(streams[i]):
        while time.perf_counter() - t0 < secs:
            nets[i].replay() if graph else nets[i].run()
            streams[i].synchronize()
            n += 1
    res[i] = n / (time.perf_counter() - t0)


def loop_request(i, secs, res, graph):
    n = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < secs:
        with torch.cuda.stream(cstreams[i]):
            nets[i].tensor("data").copy_(hin[i], non_blocking=True)
            cstreams[i].synchronize()
        with torch.cuda.stream(streams[i]):
            nets[i].replay() if graph else nets[i].run()
            streams[i].synchronize()
        with torch.cuda.stream(cstreams[i]):
            hout[i].copy_(nets[i].tensor("prob"), non_blocking=True)
            cstreams[i].synchronize()
        n += 1
    res[i] = n / (time.perf_counter() - t0)


def run(fns):
    res = {}
    th = [threading.Thread(target=f, args=a + (res,) + k) for f, a, k in fns]
    for t in th:
        t.start()
    for t in th:
        t.join()
    return res


S = 1.0
out["copy_alone_per_s"] = run([(loop_copy, (0, S), ())])[0]
for g in (False, True):
    tag = "graph" if g else "eager"
    out["pass_alone_per_s_" + tag] = run([(loop_pass, (0, S), (g,))])[0]
    r = run([(loop_copy, (1, S), ()), (loop_pass, (0, S), (g,))])
    out["copy_and_pass_together_" + tag] = {"copy_per_s": r[1], "pass_per_s": r[0]}
    r = run([(loop_pass, (0, S), (g,)), (loop_pass, (1, S), (g,))])
    out["two_pass_loops_" + tag] = {"sum_per_s": r[0] + r[1]}
    r = run([(loop_pass, (i, S), (g,)) for i in range(3)])
    out["three_pass_loops_" + tag] = {"sum_per_s": sum(r.values())}
    for k in (1, 2, 3):
        r = run([(loop_request, (i, S), (g,)) for i in range(k)])
        out["requests_%d_threads_%s" % (k, tag)] = {"requests_per_s": round(sum(r.values()), 1), "images_per_s": round(8 * sum(r.values()), 1)}
print(json.dumps(out, indent=1))
