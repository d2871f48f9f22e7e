"""What does plain streaming reach on this box at the sizes of ResNet50 FP32's res2 / res3 tensors? (torch elementwise kernels as a
neutral yardstick for the conv + in-place-sum launches: DESIGN 4.8)  usage: python scripts/probe/stream_rate.py"""
import torch

def t(fn, reps=30, flush=None):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    torch.cuda.synchronize()
    if flush is None:
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1000 / reps
    tot = 0.0
    for _ in range(reps):
        flush.add_(1.0)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        tot += e0.elapsed_time(e1) * 1000
    return tot / reps

big = torch.zeros(96 << 20, dtype=torch.float32, device="cuda")       # 384 MB: beyond the 256 MB Infinity Cache
for name, px, c in (("res2 56x56x256 b8", 25088, 256), ("res2 56x56x64 b8", 25088, 64), ("res3 28x28x512 b8", 6272, 512), ("res4 14x14x1024 b8", 1568, 1024)):
    x = torch.randn(px, c, device="cuda")
    r = torch.randn(px, c, device="cuda")
    o = torch.empty_like(x)
    mb = x.numel() * 4 / 1e6
    for label, fn, traffic in (("copy          ", lambda: o.copy_(x), 2 * mb), ("add out-place ", lambda: torch.add(x, r, out=o), 3 * mb),
                               ("add in-place  ", lambda: r.add_(x), 3 * mb), ("relu in-place ", lambda: r.relu_(), 2 * mb)):
        warm = t(fn)
        cold = t(fn, 10, big)
        # MB / us = TB/s; the swept figure is one launch between one event pair (~2.5 us of it is the pair)
        print("%-20s %6.1f MB  %s back to back %7.2f us = %4.1f TB/s | after a 384 MB sweep %7.2f us (one event pair: ~2.5 us of it) = %4.1f TB/s"
              % (name, mb, label, warm, traffic / warm, cold, traffic / (cold - 2.5)))
