"""Generate the golden vectors in tests/golden/*.npz FROM THE COMPILED REFERENCE.

Run in the build container only (needs /root/reference + oracle/_ref):
    python tests/golden/make_golden.py
Every output array in a fixture is produced by the reference's own, unmodified x86 Saber sources
(oracle/_ref/libanakin_x86_ref.so — GemmX8S8S32XConv, reorder_nhwc_nchw, SaberEltwise,
scale_conv_weights_to_nchw_host, conv_basic_check, PackedMKLInt8Gemm), never by oracle/saber_oracle.c. The reference's
tests have no golden vectors of their own (inputs come from std::random_device), so
these seeded fixtures are what pins bit-level parity; the GPU box has no /root/reference and reads
only the committed .npz files.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))

# name: (N, H, W, C, K, k, pad, stride, dil, group, in_dtype, out_dtype, relu)
CONV_I8 = {
    # ResNet50 layer shapes at reduced spatial size / batch / out-channels (same C, k, stride, pad)
    "conv_i8_res2a_2b_3x3_u8u8": (1, 14, 14, 64, 64, 3, 1, 1, 1, 1, O.U8, O.U8, 1),
    "conv_i8_res3a_2a_1x1s2_s8u8": (1, 14, 14, 256, 128, 1, 0, 2, 1, 1, O.S8, O.U8, 1),
    "conv_i8_res4_2c_1x1_u8f32": (2, 7, 7, 256, 256, 1, 0, 1, 1, 1, O.U8, O.F32, 0),
    "conv_i8_res5_2a_1x1_s8u8": (1, 7, 7, 2048, 128, 1, 0, 1, 1, 1, O.S8, O.U8, 1),
    "conv_i8_res5_2b_3x3_u8u8": (1, 7, 7, 512, 64, 3, 1, 1, 1, 1, O.U8, O.U8, 1),
    "conv_i8_branch1_1x1s2_s8s8": (1, 14, 14, 256, 128, 1, 0, 2, 1, 1, O.S8, O.S8, 0),
    "conv_i8_conv1_7x7s2_s8u8": (1, 32, 32, 3, 64, 7, 3, 2, 1, 1, O.S8, O.U8, 1),
    # edge cases: ragged sizes, dilation, groups, non-multiple-of-16 channels, no bias
    "conv_i8_ragged_dil2_s8s8": (2, 11, 9, 24, 20, 3, 2, 1, 2, 1, O.S8, O.S8, 0),
    "conv_i8_group4_u8u8": (1, 9, 9, 32, 32, 3, 1, 1, 1, 4, O.U8, O.U8, 1),
    "conv_i8_5x5_s8f32": (1, 7, 7, 20, 24, 5, 2, 1, 1, 1, O.S8, O.F32, 1),
    # (appended in round 2; seeds follow the position in this dict, so new cases go at the end)
    # u8 -> s8 (all 16 branch2c expand convs and res2a_branch1): GemmX8S8S32XConv::sub_dispatch<uint8_t,int8_t>
    "conv_i8_res2_2c_1x1_u8s8": (2, 14, 14, 64, 256, 1, 0, 1, 1, 1, O.U8, O.S8, 0),
    "conv_i8_res5_2c_1x1_u8s8": (2, 7, 7, 512, 256, 1, 0, 1, 1, 1, O.U8, O.S8, 0),
    "conv_i8_res2a_branch1_1x1_u8s8": (1, 14, 14, 64, 128, 1, 0, 1, 1, 1, O.U8, O.S8, 0),
}


def gen_conv_i8(name, spec, seed):
    N, H, W, C, K, k, pad, stride, dil, group, idt, odt, relu = spec
    rng = np.random.default_rng(seed)
    x = (rng.integers(0, 256, (N, H, W, C)).astype(np.uint8) if idt == O.U8
         else rng.integers(-128, 128, (N, H, W, C)).astype(np.int8))
    w = (rng.standard_normal((K, C // group, k, k)) * np.sqrt(2.0 / (C // group * k * k))).astype(np.float32)
    bias = (rng.standard_normal(K) * 0.5).astype(np.float32)
    in_scale = np.float32(0.0213)
    wq, ws = O.ref_quant_conv_weights(w)
    # choose an in-range out_scale from the reference's own f32-output run
    real = O.ref_conv_i8(x, wq, ws, bias, in_scale, 1.0, O.F32, relu, (pad, pad), (stride, stride),
                         (dil, dil), group)
    out_scale = np.float32(np.abs(real).max() / (125.0 if odt == O.S8 else 250.0 * 127.0 / 255.0))
    if odt == O.F32:
        out_scale = np.float32(1.0)
    y = O.ref_conv_i8(x, w, None, bias, in_scale, out_scale, odt, relu, (pad, pad), (stride, stride),
                      (dil, dil), group)
    # the f32 weights (which pin the reference's weight quantisation) are kept only when small
    extra = {"w": w} if w.size <= 40000 else {}
    np.savez_compressed(os.path.join(OUT, name + ".npz"), x=x, wq=wq, w_scale=ws, bias=bias,
                        in_scale=in_scale, out_scale=out_scale, y=y,
                        spec=np.array(spec, np.int32), **extra)


def gen_quant(seed):
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal((2, 6, 9, 7)) * 2.5).astype(np.float32)
    scale = np.float32(0.0391)
    x[0, 0, 0, :6] = np.array([0.5, 1.5, 2.5, -0.5, -1.5, 1e5], np.float32) * scale  # ties, saturation
    q8 = O.ref_reorder(x, 0, O.S8, scale)
    qu = O.ref_reorder(x, 0, O.U8, scale)
    d8 = O.ref_reorder(q8, 1, O.F32, scale)
    du = O.ref_reorder(qu, 1, O.F32, scale)
    np.savez_compressed(os.path.join(OUT, "quant_dequant.npz"), x=x, scale=scale, q_s8=q8, q_u8=qu,
                        deq_s8=d8, deq_u8=du)


def gen_eltwise(seed):
    rng = np.random.default_rng(seed)
    a = rng.integers(-128, 128, (2, 7, 7, 64)).astype(np.int8)
    b = rng.integers(-128, 128, (2, 7, 7, 64)).astype(np.int8)
    sa, sb = np.float32(0.0312), np.float32(0.0457)
    c = np.float32(1.0 / 0.05)
    y_relu = O.ref_eltwise_i8(a, b, sa, sb, c, c, True)
    y_lin = O.ref_eltwise_i8(a, b, sa, sb, 1.0, 1.0, False)
    fa = rng.standard_normal((2, 16, 5, 5)).astype(np.float32)
    fb = rng.standard_normal((2, 16, 5, 5)).astype(np.float32)
    yf = O.ref_eltwise_f32(fa, fb, 1.0, 1.0, True)
    np.savez_compressed(os.path.join(OUT, "eltwise.npz"), a=a, b=b, sa=sa, sb=sb, c=c, y_relu=y_relu,
                        y_lin=y_lin, fa=fa, fb=fb, yf=yf)


def gen_conv_f32(seed):
    rng = np.random.default_rng(seed)
    for name, (N, C, H, W, K, k, pad, stride) in {
        "conv_f32_3x3": (1, 32, 10, 10, 48, 3, 1, 1),
        "conv_f32_1x1s2": (2, 64, 9, 9, 32, 1, 0, 2),
        "conv_f32_7x7s2": (1, 3, 30, 30, 16, 7, 3, 2),
    }.items():
        x = rng.standard_normal((N, C, H, W)).astype(np.float32)
        w = (rng.standard_normal((K, C, k, k)) * np.sqrt(2.0 / (C * k * k))).astype(np.float32)
        bias = (rng.standard_normal(K) * 0.5).astype(np.float32)
        y = O.ref_conv_basic_check_f32(x, w, bias, True, (pad, pad), (stride, stride))
        np.savez_compressed(os.path.join(OUT, name + ".npz"), x=x, w=w, bias=bias, y=y,
                            spec=np.array([N, C, H, W, K, k, pad, stride], np.int32))
    x = rng.standard_normal((2, 64, 7, 7)).astype(np.float32)
    w = (rng.standard_normal((96, 64, 1, 1)) * 0.2).astype(np.float32)
    bias = rng.standard_normal(96).astype(np.float32)
    res = rng.standard_normal((2, 96, 7, 7)).astype(np.float32)
    y = O.ref_conv1x1_f32(x, w, bias, True, residual=res)  # SaberConv1X1 + fused residual (MKL sgemm)
    np.savez_compressed(os.path.join(OUT, "conv_f32_1x1_residual_mkl.npz"), x=x, w=w, bias=bias,
                        res=res, y=y)


def gen_fc_i8(seed):
    """INT8 fc with an f32 input through the reference's PackedMKLInt8Gemm (k = 2048 as ResNet50's fc1000, 128 of its
    outputs, 2 rows). The weights are float16-representable so that the fixture can store them in half the bytes."""
    rng = np.random.default_rng(seed)
    M, N, K = 2, 128, 2048
    x = (rng.standard_normal((M, K)) * 0.8).astype(np.float32)
    in_scale = np.float32(np.abs(x).max() / 127.0)
    x[0, :3] = [0.5 * in_scale, -0.5 * in_scale, 2.5 * in_scale]   # exact ties: round half away from zero
    w16 = (rng.standard_normal((N, K)) * 0.03).astype(np.float16)
    w = w16.astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    y = O.ref_fc_i8_packed(x, w, bias, float(in_scale))
    np.savez_compressed(os.path.join(OUT, "fc_i8_f32in.npz"), x=x, w=w16, bias=bias, in_scale=in_scale, y=y)


if __name__ == "__main__":
    assert O.ref_available(), "build oracle/_ref first (make -C oracle ref)"
    only_new = "--all" not in sys.argv   # default: keep the committed fixtures, write only the missing ones
    for i, (name, spec) in enumerate(CONV_I8.items()):
        if only_new and os.path.exists(os.path.join(OUT, name + ".npz")):
            continue
        gen_conv_i8(name, spec, 1000 + i)
    if only_new:
        print("new conv fixtures written to", OUT)
        sys.exit(0)
    gen_quant(2000)
    gen_eltwise(2001)
    gen_conv_f32(2002)
    gen_fc_i8(2003)
    print("golden vectors written to", OUT)
