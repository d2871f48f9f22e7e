"""Time of ResNet res5 (nine INT8 convs on 7x7) at batch N: the ops one by one (best kernels after per-op autotune) vs the
XCD-resident stage launch (saber_hip_stage_run). Back-to-back loops on one stream, torch events."""
import sys
import numpy as np
import torch

sys.path.insert(0, ".")
from anakin_amd import saber as S          # noqa: E402
from tests.test_gpu_stage import res5, dev  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rng = np.random.default_rng(1)
ph, nt = res5(rng, N)
x0 = rng.integers(-128, 128, (N, 7, 7, 1024)).astype(np.int8)
t = [dev(x0)] + [None] * (nt - 1)
for c, i, o, r in ph:
    t[o] = c.new_output()
    c.dispatch(t[i], t[o], None if r < 0 else t[r])
for c, i, o, r in ph:
    c.autotune(t[i], t[o], None if r < 0 else t[r])


def timed(fn, iters=300):
    for _ in range(20):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000.0 / iters


def ops():
    for c, i, o, r in ph:
        c.dispatch(t[i], t[o], None if r < 0 else t[r])


stage = S.SaberStage(ph)
print("batch %d: %d ops one by one %.2f us" % (N, len(ph), timed(ops)))
print("batch %d: stage launch        %.2f us" % (N, timed(lambda: stage.dispatch(t))))
stage.status()
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    stage.dispatch(t)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        for _ in range(20):
            stage.dispatch(t)
print("batch %d: stage, 20 per graph  %.2f us" % (N, timed(g.replay, 50) / 20))
stage.status()
stage.trace(arm=True)
for _ in range(3):
    stage.dispatch(t)
tr = stage.trace().astype(np.int64)
names = ["start", "arrived", "w issued", "barrier", "dma issued", "in LDS", "mma done", "end"]
t0 = tr[:, 0, 0][tr[:, 0, 0] > 0].min()      # (workgroups of XCDs without an image leave no stamps)
print("stage trace (us; median over the workgroups of the active XCDs), one row per phase: stamp - phase start | phase start - launch start")
for p in range(tr.shape[1]):
    rows = tr[:, p, :]
    act = rows[rows[:, 0] > 0]
    d = (act - act[:, :1]) / 100.0
    print("phase %d (%s): " % (p, ph[p][0].algo()) + "  ".join("%s %.2f" % (names[k], np.median(d[:, k])) for k in range(1, 8)) +
          "  | starts %.2f .. %.2f" % ((act[:, 0].min() - t0) / 100.0, (act[:, 0].max() - t0) / 100.0))
print("whole launch (first start .. last end): %.2f us" % ((tr[:, :, 7].max() - t0) / 100.0))
