"""FP32 bf16x3 convolution: unsplit tiles vs split-K (2/4/8 workgroups per tile on one XCD) on ResNet50's deep-K, few-pixel
layers at batch 8. Cold-L2 style timing is the autotuner's job; this prints the back-to-back event time per launch."""
import sys
import numpy as np
import torch

sys.path.insert(0, ".")
from anakin_amd import lib as L, saber as S   # noqa: E402

CASES = [("res3 2b", 8, 28, 28, 128, 128, 3), ("res4 2b", 8, 14, 14, 256, 256, 3), ("res5 2b", 8, 7, 7, 512, 512, 3),
         ("res5 2a", 8, 7, 7, 2048, 512, 1), ("res4 2a", 8, 14, 14, 1024, 256, 1), ("res5 2c", 8, 7, 7, 512, 2048, 1)]


def timed(fn, iters=200):
    for _ in range(10):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000.0 / iters


rng = np.random.default_rng(0)
for name, N, H, W, C, K, k in CASES:
    x = torch.from_numpy((rng.random((N, H, W, C)) * 3).astype(np.float32)).cuda()
    w = (rng.standard_normal((K, C, k, k)) * np.sqrt(2.0 / (C * k * k))).astype(np.float32)
    b = np.zeros(K, np.float32)
    conv = S.SaberConv2D(int8=False).init((N, C, H, W), S.ConvParam(w, b, 1, (k // 2, k // 2), (1, 1), (1, 1), True), L.F32, L.F32,
                                          in_layout=L.NHWC, out_layout=L.NHWC)
    y = conv.new_output()
    rows = []
    for t in range(5):
        for ks in (1, 2):
            for sh in range(4):
                try:
                    conv.set_tile(t | ((ks | (sh << 4)) << 8) | (11 << 16))
                except L.SaberHipError:
                    continue
                rows.append((timed(lambda: conv.dispatch(x, y)), conv.algo()))
    rows.sort()
    flops = 2.0 * N * H * W * C * K * k * k
    base = min(r for r in rows if "split" not in r[1])
    print("%s: best unsplit %.2f us (%s) | best %.2f us (%s) %.0f TFLOP/s" % (name, base[0], base[1], rows[0][0], rows[0][1], flops / rows[0][0] / 1e6))
    print("    " + "  ".join("%s %.2f" % (a.replace("igemm_f32_bf16x3_", ""), t) for t, a in rows[:6]))
    conv.set_tile(2 | (1 << 8) | (11 << 16))
    conv.autotune(x, y)
    print("    autotune (cold-L2 timing) picks %s: %.2f us back to back" % (conv.algo(), timed(lambda: conv.dispatch(x, y))))

# large-pixel layers (VGG16 conv3_x / conv4_x, ResNet50 res2 3x3): 4-wave vs 8-wave tiles
for name, N, H, W, C, K, k in [("vgg conv3 56x56 256->256", 8, 56, 56, 256, 256, 3), ("vgg conv4 28x28 512->512", 8, 28, 28, 512, 512, 3),
                               ("vgg conv2 112x112 128->128", 8, 112, 112, 128, 128, 3), ("res2 3x3 56x56 64->64", 8, 56, 56, 64, 64, 3)]:
    x = torch.from_numpy((rng.random((N, H, W, C)) * 3).astype(np.float32)).cuda()
    w = (rng.standard_normal((K, C, k, k)) * np.sqrt(2.0 / (C * k * k))).astype(np.float32)
    conv = S.SaberConv2D(int8=False).init((N, C, H, W), S.ConvParam(w, np.zeros(K, np.float32), 1, (1, 1), (1, 1), (1, 1), True), L.F32, L.F32,
                                          in_layout=L.NHWC, out_layout=L.NHWC)
    y = conv.new_output()
    rows = []
    for t in range(10):
        for ks in (1, 2):
            try:
                conv.set_tile(t | (ks << 8) | (11 << 16))
            except L.SaberHipError:
                continue
            rows.append((timed(lambda: conv.dispatch(x, y), 50), conv.algo()))
    rows.sort()
    flops = 2.0 * N * H * W * C * K * k * k
    print("%s: " % name + "  ".join("%s %.1f us (%.0f TF)" % (a.replace("igemm_f32_bf16x3_", ""), t, flops / t / 1e6) for t, a in rows[:5]))

# the LDS-halo form (conv3x3_b3h.hip, variants 1..5) against the best implicit-GEMM bf16-plane kernel
for name, N, H, W, C, K in [("vgg conv2 112x112 128->128", 8, 112, 112, 128, 128), ("vgg conv3 56x56 256->256", 8, 56, 56, 256, 256),
                            ("vgg conv4 28x28 512->512", 8, 28, 28, 512, 512), ("vgg conv1_2 224x224 64->64", 8, 224, 224, 64, 64),
                            ("res2 3x3 56x56 64->64", 8, 56, 56, 64, 64), ("res3 3x3 28x28 128->128", 8, 28, 28, 128, 128),
                            ("res4 3x3 14x14 256->256", 8, 14, 14, 256, 256)]:
    x = torch.from_numpy((rng.random((N, H, W, C)) * 3).astype(np.float32)).cuda()
    w = (rng.standard_normal((K, C, 3, 3)) * np.sqrt(2.0 / (C * 9))).astype(np.float32)
    conv = S.SaberConv2D(int8=False).init((N, C, H, W), S.ConvParam(w, np.zeros(K, np.float32), 1, (1, 1), (1, 1), (1, 1), True), L.F32, L.F32,
                                          in_layout=L.NHWC, out_layout=L.NHWC)
    y = conv.new_output()
    rows = []
    for v in range(1, 6):      # (before the autotune: it releases the fragment-ordered planes of a family it does not select)
        conv.set_tile(v | (13 << 16))
        rows.append((timed(lambda: conv.dispatch(x, y), 50), conv.algo()))
    conv.autotune(x, y)
    t_auto, a_auto = timed(lambda: conv.dispatch(x, y), 50), conv.algo()
    flops = 2.0 * N * H * W * C * K * 9
    print("%s: autotune %s %.1f us (%.0f TF) | halo: " % (name, a_auto.replace("igemm_f32_bf16x3_", ""), t_auto, flops / t_auto / 1e6) +
          "  ".join("%s %.1f us (%.0f TF)" % (a.replace("halo3x3_f32_bf16x3_", ""), t, flops / t / 1e6) for t, a in rows))

# the pointwise forms (variants 6..8) on ResNet50's 1x1 layers at batch 8, against what the autotuner picks among the others
for name, N, HW, C, K, elt in [("res2 2c 64->256 +sum", 8, 56, 64, 256, True), ("res2 2a 256->64", 8, 56, 256, 64, False),
                               ("res3 2c 128->512 +sum", 8, 28, 128, 512, True), ("res3 2a 512->128", 8, 28, 512, 128, False),
                               ("res4 2c 256->1024 +sum", 8, 14, 256, 1024, True), ("res4 2a 1024->256", 8, 14, 1024, 256, False)]:
    x = torch.from_numpy((rng.random((N, HW, HW, C)) * 3).astype(np.float32)).cuda()
    w = (rng.standard_normal((K, C, 1, 1)) * np.sqrt(2.0 / C)).astype(np.float32)
    p = S.ConvParam(w, np.zeros(K, np.float32), 1, (0, 0), (1, 1), (1, 1), not elt)
    if elt:
        p.res_mode, p.res_relu, p.sum_scale = L.RES_SUM_INPLACE, True, 1.0
    conv = S.SaberConv2D(int8=False).init((N, C, HW, HW), p, L.F32, L.F32, in_layout=L.NHWC, out_layout=L.NHWC)
    y = conv.new_output()
    rows = []
    for v in (6, 7, 8):
        conv.set_tile(v | (13 << 16))
        rows.append((timed(lambda: conv.dispatch(x, y), 100), conv.algo()))
    conv.autotune(x, y)
    t_auto, a_auto = timed(lambda: conv.dispatch(x, y), 100), conv.algo()
    byt = 4.0 * N * HW * HW * (C + K * (2 if elt else 1))
    print("%s: autotune %s %.1f us (%.2f TB/s) | pw: " % (name, a_auto, t_auto, byt / t_auto / 1e6) +
          "  ".join("%s %.1f us" % (a.replace("pw1x1_f32_bf16x3_", ""), t) for t, a in rows))
