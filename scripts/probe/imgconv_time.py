"""res5's convs at batch 8: autotuned default kernels vs the image-resident kernel (variant 12), back to back and what the cold-L2
autotuner picks."""
import sys
import numpy as np
import torch

sys.path.insert(0, ".")
from anakin_amd import saber as S          # noqa: E402
from tests.test_gpu_stage import conv, dev, IMG_CASES  # noqa: E402
from oracle import oracle as O             # noqa: E402


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


N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rng = np.random.default_rng(3)
for cin, cout, k, relu, idt, odt, elt in IMG_CASES:
    c = conv(rng, N, 7, 7, cin, cout, k, relu, idt, odt, 0.03, 0.05, (0.04, 0.05) if elt else None)
    x = dev(rng.integers(0, 256, (N, 7, 7, cin)).astype(np.uint8) if idt == O.U8 else rng.integers(-128, 128, (N, 7, 7, cin)).astype(np.int8))
    res = dev(rng.integers(-128, 128, (N, 7, 7, cout)).astype(np.int8)) if elt else None
    y = c.new_output()
    c.dispatch(x, y, res)
    c.autotune(x, y, res)
    picked = c.algo()
    t_pick = timed(lambda: c.dispatch(x, y, res))
    c.set_tile(12 << 16)
    t_img = timed(lambda: c.dispatch(x, y, res))
    print("%4d -> %4d %dx%d%s: autotune picks %-36s %.2f us | image-resident %.2f us" % (cin, cout, k, k, " +elt" if elt else "", picked, t_pick, t_img))
