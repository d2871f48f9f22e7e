"""Does a kernel selection tuned UNDER LOAD serve more images/s with four passes in flight than the selection tuned on an idle GPU (round 6)?
Four shared-device ResNet50 INT8 batch-8 nets on the serving streams (saber_hip_serving_streams); selection A: net 0 autotuned alone (what bench.py's
multi_stream leg does); selection B: net 0 autotuned while nets 1..3 replay continuously on their streams. Each selection is applied to all four nets
(compacted arenas, hipGraph replay) and the four-way and single-net rates are measured."""
import sys
import threading
import time

import torch

sys.path.insert(0, ".")
from anakin_amd import workloads as W  # noqa: E402
from anakin_amd.streams import serving_streams  # noqa: E402

B = 8
model = W.framework_model(W.build_model("resnet50"), "int8")
scales = W.calibrate(model, W.make_input(2))
streams, distinct = serving_streams(4)
nets = []
for i, st in enumerate(streams):
    with torch.cuda.stream(st):
        n = W.build_int8_net(model, dict(scales), B, shared_device=True)
        n.tensor("data").copy_(torch.from_numpy(W.make_input(B, seed=11 + i)).cuda())
        n.run()
        n.compact()
        n.run()
        n.capture()
    nets.append(n)
torch.cuda.synchronize()


def rate(k, rounds=200):
    grp = list(zip(nets[:k], streams[:k]))
    for _ in range(20):
        for n, st in grp:
            with torch.cuda.stream(st):
                n.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(rounds):
        for n, st in grp:
            with torch.cuda.stream(st):
                n.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / rounds
    return k * B / dt, dt * 1e3


def apply(ch):
    for n, st in zip(nets, streams):
        with torch.cuda.stream(st):
            n.set_choices(ch)
            n.run()
            n.capture()
    torch.cuda.synchronize()


def tune(loaded):
    stop = threading.Event()

    def background():
        while not stop.is_set():
            for n, st in zip(nets[1:], streams[1:]):
                with torch.cuda.stream(st):
                    n.replay()
            streams[1].synchronize()
    th = None
    if loaded:
        th = threading.Thread(target=background)
        th.start()
        time.sleep(0.05)
    with torch.cuda.stream(streams[0]):
        nets[0].autotune(iters=9)
        ch = nets[0].choices()
    if th:
        stop.set()
        th.join()
    torch.cuda.synchronize()
    return ch


for rep in range(2):
    for loaded in (False, True):
        ch = tune(loaded)
        apply(ch)
        r = [rate(k) for k in (1, 2, 3, 4)]
        print("rep %d selection tuned %-10s | %s | launches %d" % (rep, "under load" if loaded else "idle", " | ".join("%d: %6.0f img/s (%.3f ms)" % (k + 1, a, b) for k, (a, b) in enumerate(r)), nets[0].num_launches()), flush=True)
