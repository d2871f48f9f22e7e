"""rocprofv3 target: k shared-device ResNet50 INT8 batch-8 nets (compacted arenas) replayed together on the serving streams for 100 rounds
(argv[1] = k, default 4) - `rocprofv3 --kernel-trace --stats` of it shows what every kernel of the pass costs with k - 1 other passes in flight."""
import sys

import torch

sys.path.insert(0, ".")
from anakin_amd import workloads as W  # noqa: E402
from anakin_amd.streams import serving_streams  # noqa: E402

k = int(sys.argv[1]) if len(sys.argv) > 1 else 4
model = W.framework_model(W.build_model("resnet50"), "int8")
scales = W.calibrate(model, W.make_input(2))
streams, distinct = serving_streams(k)
nets = []
for i, st in enumerate(streams):
    with torch.cuda.stream(st):
        n = W.build_int8_net(model, dict(scales), 8, shared_device=True)
        n.tensor("data").copy_(torch.from_numpy(W.make_input(8, seed=11 + i)).cuda())
        n.run()
        if i == 0:
            n.autotune(iters=5)
            ch = n.choices()
        else:
            n.set_choices(ch)
        n.compact()
        n.run()
        n.capture()
    nets.append(n)
torch.cuda.synchronize()
for _ in range(120):
    for n, st in zip(nets, streams):
        with torch.cuda.stream(st):
            n.replay()
torch.cuda.synchronize()
print("launches per net", nets[0].num_launches())
