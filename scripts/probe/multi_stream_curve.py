"""How many independent batch-B ResNet50 INT8 passes in flight give the most images/s on one MI355X (round 6)?
k shared-device nets (SABER_HIP_NET_SHARED_DEVICE: no placement-dependent kernels), one selection tuned once, each a hipGraph on its own
stream; k = 1 .. 6 at batch 8 and batch 4; every edge in its own slot and with compacted arenas; also EAGER launches from one host thread."""
import sys
import time

import torch

sys.path.insert(0, ".")
from anakin_amd import workloads as W  # noqa: E402
from anakin_amd.streams import serving_streams  # noqa: E402

PICK = "--pick" in sys.argv      # the first four streams from saber_hip_serving_streams: no two share a hardware queue

model = W.framework_model(W.build_model("resnet50"), "int8")
scales = W.calibrate(model, W.make_input(2))
BATCHES = tuple(int(a) for a in sys.argv[1:] if a.isdigit()) or (8, 4)
for B in BATCHES:
    nets, streams = [], []
    picked = []
    if PICK:
        picked, distinct = serving_streams(4)
        print("batch %d: %d streams on distinct hardware queues" % (B, distinct), flush=True)
    for i in range(6):
        st = picked[i] if i < len(picked) else torch.cuda.Stream()
        with torch.cuda.stream(st):
            n = W.build_int8_net(model, dict(scales), B, shared_device=True)
            n.tensor("data").copy_(torch.from_numpy(W.make_input(B, seed=11 + i)).cuda())
            n.run()
            if i == 0:
                n.autotune(iters=7)
                ch = n.choices()
            else:
                n.set_choices(ch)
            n.run()
            n.capture()
        nets.append(n)
        streams.append(st)
    torch.cuda.synchronize()
    for form in ("every_edge", "compact"):
        if form == "compact":
            for n, st in zip(nets, streams):
                with torch.cuda.stream(st):
                    n.compact()
                    n.run()
                    n.capture()
            torch.cuda.synchronize()
        for mode in ("graph", "eager"):
            line = []
            for k in range(1, 7):
                grp = list(zip(nets[:k], streams[:k]))

                def rnd():
                    for n, st in grp:
                        with torch.cuda.stream(st):
                            n.replay() if mode == "graph" else n.run()
                for _ in range(20):
                    rnd()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(150):
                    rnd()
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / 150
                line.append("%d: %6.0f img/s (%.3f ms/round)" % (k, k * B / dt, dt * 1e3))
            print("batch %d %-10s %-5s | %s" % (B, form, mode, " | ".join(line)), flush=True)
    del nets, streams
