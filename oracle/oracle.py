"""ctypes/numpy bindings for the CPU oracle (oracle/libsaber_oracle.so) and, when it has been
built, the compiled reference (oracle/_ref/libanakin_x86_ref.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg. Nothing under anakin_amd/ may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
F32, S8, U8 = 0, 1, 2
NP_DTYPE = {F32: np.float32, S8: np.int8, U8: np.uint8}
RES_NONE, RES_JIT_SUM, RES_ELTWISE = 0, 1, 2

_P = C.c_void_p
_f = C.c_float


def _ptr(a):
    return None if a is None else a.ctypes.data_as(_P)


def code_of(arr):
    return {np.dtype(np.float32): F32, np.dtype(np.int8): S8, np.dtype(np.uint8): U8}[arr.dtype]


def build(force=False):
    """Compile the oracle (and oracle/_ref when /root/reference is present)."""
    so = os.path.join(HERE, "libsaber_oracle.so")
    src = os.path.join(HERE, "saber_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", HERE, "oracle"])
    if os.path.isdir("/root/reference/saber"):
        subprocess.check_call(["make", "-s", "-C", HERE, "ref"])


_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        so = os.environ.get("SABER_ORACLE_LIB")      # another build of saber_oracle.c (tests/test_oracle_sanitized.py: ASan + UBSan)
        if not so:
            so = os.path.join(HERE, "libsaber_oracle.so")
            if not os.path.exists(so):
                build()
        _lib = C.CDLL(so)
    return _lib


def ref_available():
    return os.path.exists(os.path.join(HERE, "_ref", "libanakin_x86_ref.so"))


def ref():
    """The reference's own x86 Saber objects (oracle/_ref). MKL wants the GNU threading layer
    because the objects are built with gcc -fopenmp."""
    global _ref
    if _ref is None:
        os.environ.setdefault("MKL_THREADING_LAYER", "GNU")
        # the reference's logger writes INFO lines to stdout on Env init; silence fd 1 briefly
        _ref = C.CDLL(os.path.join(HERE, "_ref", "libanakin_x86_ref.so"))
    return _ref


class Residual(C.Structure):
    _fields_ = [("mode", C.c_int), ("with_relu", C.c_int), ("sum_scale", _f), ("res_dtype", C.c_int),
                ("coeff_conv", _f), ("coeff_res", _f), ("scale_conv", _f), ("scale_res", _f)]


def conv_out_hw(H, W, kh, kw, pad, stride, dil):
    oh = (H + 2 * pad[0] - (dil[0] * (kh - 1) + 1)) // stride[0] + 1
    ow = (W + 2 * pad[1] - (dil[1] * (kw - 1) + 1)) // stride[1] + 1
    return oh, ow


# ---------------------------------------------------------------- oracle (restatement)

def weight_scales(w):
    w = np.ascontiguousarray(w, np.float32)
    K = w.shape[0]
    out = np.empty(K, np.float32)
    lib().orc_weight_scales(_ptr(w), K, int(w.size // K), _ptr(out))
    return out


def quant_weights(w, scale):
    w = np.ascontiguousarray(w, np.float32)
    scale = np.ascontiguousarray(scale, np.float32)
    K = w.shape[0]
    q = np.empty(w.shape, np.int8)
    lib().orc_quant_weights(_ptr(w), K, int(w.size // K), _ptr(scale), _ptr(q))
    return q


def conv_i8_prepare(w_scale, bias, in_scale, out_scale, in_dtype, out_dtype):
    w_scale = np.ascontiguousarray(w_scale, np.float32)
    K = w_scale.size
    bias_p = np.zeros(K, np.float32)
    scale = np.empty(K, np.float32)
    b = None if bias is None else np.ascontiguousarray(bias, np.float32)
    lib().orc_conv_i8_prepare(K, _ptr(w_scale), _ptr(b), _f(in_scale), _f(out_scale), in_dtype,
                              out_dtype, _ptr(bias_p), _ptr(scale))
    return (bias_p if bias is not None else None), scale


def conv_i8(x, wq, bias_p, scale, out_dtype, relu, pad=(0, 0), stride=(1, 1), dil=(1, 1), group=1,
            residual=None, res=None, out_init=None):
    """x NHWC s8/u8, wq OIHW s8 -> NHWC out_dtype. residual: Residual or None."""
    x = np.ascontiguousarray(x)
    wq = np.ascontiguousarray(wq, np.int8)
    N, H, W, Cc = x.shape
    K, _, kh, kw = wq.shape
    oh, ow = conv_out_hw(H, W, kh, kw, pad, stride, dil)
    out = np.zeros((N, oh, ow, K), NP_DTYPE[out_dtype]) if out_init is None else \
        np.ascontiguousarray(out_init).copy()
    bp = None if bias_p is None else np.ascontiguousarray(bias_p, np.float32)
    scale = np.ascontiguousarray(scale, np.float32)
    r = None if res is None else np.ascontiguousarray(res)
    rc = lib().orc_conv_i8(N, H, W, Cc, K, kh, kw, pad[0], pad[1], stride[0], stride[1], dil[0], dil[1],
                           group, code_of(x), out_dtype, int(relu), _ptr(x), _ptr(wq), _ptr(bp),
                           _ptr(scale), C.byref(residual) if residual is not None else None,
                           _ptr(r), _ptr(out))
    assert rc == 0, rc
    return out


def conv_i8_acc(x, wq, pad=(0, 0), stride=(1, 1), dil=(1, 1), group=1):
    x = np.ascontiguousarray(x)
    wq = np.ascontiguousarray(wq, np.int8)
    N, H, W, Cc = x.shape
    K, _, kh, kw = wq.shape
    oh, ow = conv_out_hw(H, W, kh, kw, pad, stride, dil)
    out = np.zeros((N, oh, ow, K), np.int32)
    lib().orc_conv_i8_acc(N, H, W, Cc, K, kh, kw, pad[0], pad[1], stride[0], stride[1], dil[0], dil[1],
                          group, code_of(x), _ptr(x), _ptr(wq), _ptr(out))
    return out


def conv_f32_nchw(x, w, bias, relu, pad=(0, 0), stride=(1, 1), dil=(1, 1), group=1, alpha=1.0,
                  beta=0.0, out_init=None):
    x = np.ascontiguousarray(x, np.float32)
    w = np.ascontiguousarray(w, np.float32)
    N, Cc, H, W = x.shape
    K, _, kh, kw = w.shape
    oh, ow = conv_out_hw(H, W, kh, kw, pad, stride, dil)
    out = np.zeros((N, K, oh, ow), np.float32) if out_init is None else \
        np.ascontiguousarray(out_init, np.float32).copy()
    b = None if bias is None else np.ascontiguousarray(bias, np.float32)
    rc = lib().orc_conv_f32_nchw(N, Cc, H, W, K, kh, kw, pad[0], pad[1], stride[0], stride[1], dil[0],
                                 dil[1], group, _ptr(x), _ptr(w), _ptr(b), int(relu), _f(alpha),
                                 _f(beta), _ptr(out))
    assert rc == 0, rc
    return out


def quant_nchw_to_nhwc(x, scale, out_dtype):
    x = np.ascontiguousarray(x, np.float32)
    N, Cc, H, W = x.shape
    out = np.empty((N, H, W, Cc), NP_DTYPE[out_dtype])
    lib().orc_quant_nchw_to_nhwc(N, Cc, H, W, out_dtype, _f(scale), _ptr(x), _ptr(out))
    return out


def dequant_nhwc_to_nchw(x, scale):
    x = np.ascontiguousarray(x)
    N, H, W, Cc = x.shape
    out = np.empty((N, Cc, H, W), np.float32)
    lib().orc_dequant_nhwc_to_nchw(N, Cc, H, W, code_of(x), _f(scale), _ptr(x), _ptr(out))
    return out


def quant_flat_s8(x, scale):
    x = np.ascontiguousarray(x, np.float32)
    out = np.empty(x.shape, np.int8)
    lib().orc_quant_flat_s8(C.c_size_t(x.size), _f(scale), _ptr(x), _ptr(out))
    return out


def eltwise_i8(a, b, sa, sb, c0=1.0, c1=1.0, relu=True):
    a = np.ascontiguousarray(a, np.int8)
    b = np.ascontiguousarray(b, np.int8)
    out = np.empty(a.shape, np.int8)
    lib().orc_eltwise_i8(C.c_size_t(a.size), _ptr(a), _ptr(b), _f(sa), _f(sb), _f(c0), _f(c1),
                         int(relu), _ptr(out))
    return out


def eltwise_f32(a, b, c0=1.0, c1=1.0, relu=True):
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    out = np.empty(a.shape, np.float32)
    lib().orc_eltwise_f32(C.c_size_t(a.size), _ptr(a), _ptr(b), _f(c0), _f(c1), int(relu), _ptr(out))
    return out


def pool_out_dim(inp, pad, win, stride, floor_mode=False):
    return lib().orc_pool_out_dim(inp, pad, win, stride, int(floor_mode))


def pool_i8_nhwc(x, win, stride, pad, ptype, out_dtype=None, global_pool=False, floor_mode=False):
    x = np.ascontiguousarray(x)
    N, H, W, Cc = x.shape
    if global_pool:
        win, stride, pad = (H, W), (H, W), (0, 0)
        oh = ow = 1
    else:
        oh = pool_out_dim(H, pad[0], win[0], stride[0], floor_mode)
        ow = pool_out_dim(W, pad[1], win[1], stride[1], floor_mode)
    od = code_of(x) if out_dtype is None else out_dtype
    out = np.empty((N, oh, ow, Cc), NP_DTYPE[od])
    lib().orc_pool_i8_nhwc(N, H, W, Cc, oh, ow, win[0], win[1], stride[0], stride[1], pad[0], pad[1],
                           ptype, code_of(x), od, _ptr(x), _ptr(out))
    return out


def pool_f32_nchw(x, win, stride, pad, ptype, global_pool=False, floor_mode=False):
    x = np.ascontiguousarray(x, np.float32)
    N, Cc, H, W = x.shape
    if global_pool:
        win, stride, pad = (H, W), (H, W), (0, 0)
        oh = ow = 1
    else:
        oh = pool_out_dim(H, pad[0], win[0], stride[0], floor_mode)
        ow = pool_out_dim(W, pad[1], win[1], stride[1], floor_mode)
    out = np.empty((N, Cc, oh, ow), np.float32)
    lib().orc_pool_f32_nchw(N, Cc, H, W, oh, ow, win[0], win[1], stride[0], stride[1], pad[0], pad[1],
                            ptype, _ptr(x), _ptr(out))
    return out


def gemm_f32(A, B, M, N, K, trans_a=False, trans_b=False, alpha=1.0, beta=0.0, Cinit=None):
    A = np.ascontiguousarray(A, np.float32)
    B = np.ascontiguousarray(B, np.float32)
    out = np.zeros((M, N), np.float32) if Cinit is None else np.ascontiguousarray(Cinit, np.float32).copy()
    lib().orc_gemm_f32(int(trans_a), int(trans_b), M, N, K, _f(alpha), _ptr(A), _ptr(B), _f(beta),
                       _ptr(out))
    return out


def fc_f32(x, w, bias, w_is_kn=False):
    x = np.ascontiguousarray(x, np.float32)
    w = np.ascontiguousarray(w, np.float32)
    M, K = x.shape
    N = w.shape[1] if w_is_kn else w.shape[0]
    out = np.empty((M, N), np.float32)
    b = None if bias is None else np.ascontiguousarray(bias, np.float32)
    lib().orc_fc_f32(M, N, K, _ptr(x), _ptr(w), int(w_is_kn), _ptr(b), _ptr(out))
    return out


def fc_i8(x, wq, w_scale, in_scale, bias, out_scale=1.0):
    """x [M,K] s8 or u8; wq [N,K] s8 -> f32 [M,N]."""
    x = np.ascontiguousarray(x)
    wq = np.ascontiguousarray(wq, np.int8)
    w_scale = np.ascontiguousarray(w_scale, np.float32)
    M, K = x.shape
    N = wq.shape[0]
    out = np.empty((M, N), np.float32)
    b = None if bias is None else np.ascontiguousarray(bias, np.float32)
    if x.dtype == np.uint8:
        lib().orc_fc_i8_u8in(M, N, K, _ptr(x), _ptr(wq), _ptr(w_scale), _f(in_scale), _f(out_scale),
                             _ptr(b), _ptr(out))
    else:
        lib().orc_fc_i8_s8in(M, N, K, _ptr(x), _ptr(wq), _ptr(w_scale), _f(in_scale), _ptr(b), _ptr(out))
    return out


def bn_fold(w, bias, bn_scale, eps, mean, var, scale_w, scale_b):
    w = np.ascontiguousarray(w, np.float32).copy()
    K = w.shape[0]
    has_bias = bias is not None
    b = np.ascontiguousarray(bias, np.float32).copy() if has_bias else np.zeros(K, np.float32)
    mean = np.ascontiguousarray(mean, np.float32)
    var = np.ascontiguousarray(var, np.float32)
    scale_w = np.ascontiguousarray(scale_w, np.float32)
    sb = None if scale_b is None else np.ascontiguousarray(scale_b, np.float32)
    lib().orc_bn_fold(K, int(w.size // K), _ptr(w), _ptr(b), int(has_bias), _f(bn_scale), _f(eps),
                      _ptr(mean), _ptr(var), _ptr(scale_w), _ptr(sb))
    return w, b


def activation_f32(x, active, negative_slope=0.0, coef=1.0):
    """SaberActivation<X86, AK_FLOAT>::dispatch (saber/funcs/impl/x86/saber_activation.cpp:136-262) and the scalar formulas the
    reference's own test checks it against (test/saber/test_saber_activation.cpp:17-115), per ActiveType value (saber_types.h:283-
    293). PARITY UNPINNED for these types: saber_activation.cpp needs xbyak (jit_generator.h) and cannot be compiled into
    oracle/_ref here; this is the published formula in f32 with numpy's libm."""
    x = np.asarray(x, np.float32)
    one = np.float32(1.0)
    if active == 2:       # relu (the standalone operator ignores negative_slope)
        return np.where(x > 0, x, np.float32(0)).astype(np.float32)
    if active == 1:       # sigmoid
        return (one / (one + np.exp(-x))).astype(np.float32)
    if active == 3:       # tanh
        return np.tanh(x).astype(np.float32)
    if active == 4:       # clipped relu, threshold = coef
        r = np.where(x > 0, x, np.float32(0))
        return np.where(r < np.float32(coef), r, np.float32(coef)).astype(np.float32)
    if active == 5:       # elu
        return np.where(x > 0, x, np.float32(coef) * (np.exp(x) - one)).astype(np.float32)
    if active == 9:       # stanh
        return (np.float32(coef) * np.tanh(np.float32(negative_slope) * x)).astype(np.float32)
    if active == 11:      # gelu: x * 0.5 * (erf(x / sqrt(2)) + 1)
        from math import erf, sqrt
        e = np.vectorize(lambda v: erf(float(v) / sqrt(2.0)))(x).astype(np.float32)
        return (x * (np.float32(0.5) * (e + one))).astype(np.float32)
    if active == 12:      # swish, beta = coef
        return (x / (one + np.exp(-x * np.float32(coef)))).astype(np.float32)
    raise ValueError("activation type %d" % active)


def prelu_f32(x, slope, channel_axis, channel_shared=False):
    """excute_prelu (saber_activation.cpp:38-132): y = x > 0 ? x : x * slope[channel] (slope[0] when channel_shared)"""
    x = np.asarray(x, np.float32)
    slope = np.asarray(slope, np.float32)
    if channel_shared:
        sl = slope.ravel()[0]
    else:
        shape = [1] * x.ndim
        shape[channel_axis] = x.shape[channel_axis]
        sl = slope.reshape(shape)
    return np.where(x > 0, x, x * sl).astype(np.float32)


def softmax_f32(x):
    x = np.ascontiguousarray(x, np.float32)
    outer, Cn = x.shape[0], x.shape[1]
    inner = int(x.size // (outer * Cn))
    out = np.empty(x.shape, np.float32)
    lib().orc_softmax_f32(outer, Cn, inner, _ptr(x), _ptr(out))
    return out


# ---------------------------------------------------------------- compiled reference (oracle/_ref)

def ref_conv_i8(x, w, w_scale, bias, in_scale, out_scale, out_dtype, relu, pad=(0, 0), stride=(1, 1),
                dil=(1, 1), group=1):
    """w: f32 OIHW (reference quantises it) or s8 OIHW with w_scale[K]."""
    x = np.ascontiguousarray(x)
    w = np.ascontiguousarray(w)
    N, H, W, Cc = x.shape
    K, _, kh, kw = w.shape
    oh, ow = conv_out_hw(H, W, kh, kw, pad, stride, dil)
    out = np.zeros((N, oh, ow, K), NP_DTYPE[out_dtype])
    ws = None if w_scale is None else np.ascontiguousarray(w_scale, np.float32)
    b = None if bias is None else np.ascontiguousarray(bias, np.float32)
    rc = ref().ref_conv_i8(N, H, W, Cc, K, kh, kw, pad[0], pad[1], stride[0], stride[1], dil[0], dil[1],
                           group, code_of(x), out_dtype, code_of(w), int(relu), _ptr(x), _ptr(w),
                           _ptr(ws), _ptr(b), _f(in_scale), _f(out_scale), _ptr(out))
    assert rc == 0, rc
    return out


def ref_quant_conv_weights(w):
    w = np.ascontiguousarray(w, np.float32)
    K, Cc, kh, kw = w.shape
    q = np.empty(w.shape, np.int8)
    s = np.empty(K, np.float32)
    ref().ref_quant_conv_weights(K, Cc, kh, kw, _ptr(w), _ptr(q), _ptr(s))
    return q, s


def ref_reorder(x, direction, dst_dtype, scale):
    """direction 0: NCHW array -> NHWC; 1: NHWC array -> NCHW."""
    x = np.ascontiguousarray(x)
    if direction == 0:
        N, Cc, H, W = x.shape
        out = np.empty((N, H, W, Cc), NP_DTYPE[dst_dtype])
    else:
        N, H, W, Cc = x.shape
        out = np.empty((N, Cc, H, W), NP_DTYPE[dst_dtype])
    ref().ref_reorder(direction, N, Cc, H, W, code_of(x), dst_dtype, _f(scale), _ptr(x), _ptr(out))
    return out


def ref_eltwise_i8(a, b, sa, sb, c0=1.0, c1=1.0, relu=True, out_scale=1.0):
    a = np.ascontiguousarray(a, np.int8)
    b = np.ascontiguousarray(b, np.int8)
    N, H, W, Cc = a.shape
    out = np.empty(a.shape, np.int8)
    rc = ref().ref_eltwise_i8(N, H, W, Cc, _ptr(a), _ptr(b), _f(sa), _f(sb), _f(c0), _f(c1), int(relu),
                              _f(out_scale), _ptr(out))
    assert rc == 0, rc
    return out


def ref_eltwise_f32(a, b, c0=1.0, c1=1.0, relu=True):
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    N, Cc, H, W = a.shape
    out = np.empty(a.shape, np.float32)
    rc = ref().ref_eltwise_f32(N, Cc, H, W, _ptr(a), _ptr(b), _f(c0), _f(c1), int(relu), _ptr(out))
    assert rc == 0, rc
    return out


def ref_conv1x1_f32(x, w, bias, relu, residual=None):
    x = np.ascontiguousarray(x, np.float32)
    w = np.ascontiguousarray(w, np.float32)
    N, Cc, H, W = x.shape
    K = w.shape[0]
    out = np.zeros((N, K, H, W), np.float32) if residual is None else \
        np.ascontiguousarray(residual, np.float32).copy()
    b = None if bias is None else np.ascontiguousarray(bias, np.float32)
    rc = ref().ref_conv1x1_f32(N, Cc, H, W, K, _ptr(x), _ptr(w), _ptr(b), int(relu),
                               int(residual is not None), _ptr(out))
    assert rc == 0, rc
    return out


def ref_fc_i8_packed(x_f32, w_nk_f32, bias, in_scale):
    """The reference's INT8 fc for an f32 input (VenderFc<X86,AK_INT8> -> PackedMKLInt8Gemm): f32 [M,N]."""
    x = np.ascontiguousarray(x_f32, np.float32)
    w = np.ascontiguousarray(w_nk_f32, np.float32)
    M, K = x.shape
    N = w.shape[0]
    out = np.empty((M, N), np.float32)
    b = None if bias is None else np.ascontiguousarray(bias, np.float32)
    rc = ref().ref_fc_i8_packed(M, N, K, _ptr(x), _ptr(w), _ptr(b), _f(in_scale), _ptr(out))
    assert rc == 0, rc
    return out


def ref_vender_fc_i8(x, w_nk_f32, bias, in_scale, out_scale=1.0):
    """The reference's whole VenderFc<X86,AK_INT8> operator (f32 / s8 / u8 input [M,K], f32 weights [N,K]) -> f32 [M,N]."""
    x = np.ascontiguousarray(x)
    w = np.ascontiguousarray(w_nk_f32, np.float32)
    M, K = x.shape
    N = w.shape[0]
    out = np.empty((M, N), np.float32)
    b = None if bias is None else np.ascontiguousarray(bias, np.float32)
    rc = ref().ref_vender_fc_i8(M, N, K, code_of(x), _ptr(x), _ptr(w), _ptr(b), _f(in_scale), _f(out_scale), _ptr(out))
    assert rc == 0, rc
    return out


def ref_bn_fold(w, bias, bn_scale, eps, mean, var, scale_w, scale_b):
    """WeightsFusion<float,X86>::update_weights on the compiled reference: (folded weights, folded bias)."""
    w = np.ascontiguousarray(w, np.float32).copy()
    K, Cc, kh, kw = w.shape
    has_bias = bias is not None
    b = np.ascontiguousarray(bias, np.float32).copy() if has_bias else np.zeros(K, np.float32)
    sb = None if scale_b is None else np.ascontiguousarray(scale_b, np.float32)
    rc = ref().ref_bn_fold(K, Cc, kh, kw, _ptr(w), _ptr(b), int(has_bias), _f(bn_scale), _f(eps),
                           _ptr(np.ascontiguousarray(mean, np.float32)), _ptr(np.ascontiguousarray(var, np.float32)),
                           _ptr(np.ascontiguousarray(scale_w, np.float32)), _ptr(sb))
    assert rc == 0, rc
    return w, b


def ref_gemm_s8s8s32(a, b, M, N, K, trans_a=False, trans_b=False):
    """MklDnnGemm<int8_t,int8_t,int> (packed B): s32 [M,N]."""
    a = np.ascontiguousarray(a, np.int8)
    b = np.ascontiguousarray(b, np.int8)
    out = np.empty((M, N), np.int32)
    rc = ref().ref_gemm_s8s8s32(int(trans_a), int(trans_b), M, N, K, _ptr(a), _ptr(b), _ptr(out))
    assert rc == 0, rc
    return out


def ref_conv_basic_check_f32(x, w, bias, relu, pad=(0, 0), stride=(1, 1), dil=(1, 1), group=1):
    x = np.ascontiguousarray(x, np.float32)
    w = np.ascontiguousarray(w, np.float32)
    N, Cc, H, W = x.shape
    K, _, kh, kw = w.shape
    oh, ow = conv_out_hw(H, W, kh, kw, pad, stride, dil)
    out = np.zeros((N, K, oh, ow), np.float32)
    b = None if bias is None else np.ascontiguousarray(bias, np.float32)
    ref().ref_conv_basic_check_f32(N, Cc, H, W, K, kh, kw, pad[0], pad[1], stride[0], stride[1], dil[0],
                                   dil[1], group, _ptr(x), _ptr(w), _ptr(b), int(relu), _ptr(out))
    return out


def ref_conv_basic_check_int8(x, wq, bias_i32, scale, relu, pad=(0, 0), stride=(1, 1), dil=(1, 1),
                              group=1):
    x = np.ascontiguousarray(x)
    wq = np.ascontiguousarray(wq, np.int8)
    N, H, W, Cc = x.shape
    K, _, kh, kw = wq.shape
    oh, ow = conv_out_hw(H, W, kh, kw, pad, stride, dil)
    out = np.zeros((N, oh, ow, K), np.int8)
    b = None if bias_i32 is None else np.ascontiguousarray(bias_i32, np.int32)
    scale = np.ascontiguousarray(scale, np.float32)
    ref().ref_conv_basic_check_int8(N, H, W, Cc, K, kh, kw, pad[0], pad[1], stride[0], stride[1],
                                    dil[0], dil[1], group, code_of(x), _ptr(x), _ptr(wq), _ptr(b),
                                    int(relu), _ptr(scale), _ptr(out))
    return out


def ref_conv_basic_check_int8_sum(x, wq, scale, relu, prev, sum_scale, pad=(0, 0), stride=(1, 1)):
    """conv_basic_check_int8 with its Eltwise_sum post-op: `prev` (s8 NHWC, the bytes already in the output) is summed in."""
    x = np.ascontiguousarray(x)
    wq = np.ascontiguousarray(wq, np.int8)
    N, H, W, Cc = x.shape
    K, _, kh, kw = wq.shape
    out = np.ascontiguousarray(prev, np.int8).copy()
    scale = np.ascontiguousarray(scale, np.float32)
    ref().ref_conv_basic_check_int8_sum.argtypes = [C.c_int] * 15 + [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                                                     C.c_float, C.c_void_p]
    ref().ref_conv_basic_check_int8_sum(N, H, W, Cc, K, kh, kw, pad[0], pad[1], stride[0], stride[1], 1, 1, 1, code_of(x),
                                        _ptr(x), _ptr(wq), None, int(relu), _ptr(scale), float(sum_scale), _ptr(out))
    return out


def ref_pool_basic_check_int8(x, oh, ow, win, stride, pad, ptype):
    x = np.ascontiguousarray(x)
    N, H, W, Cc = x.shape
    out = np.empty((N, oh, ow, Cc), x.dtype)
    ref().ref_pool_basic_check_int8(N, H, W, Cc, oh, ow, win[0], win[1], stride[0], stride[1], pad[0],
                                    pad[1], ptype, code_of(x), _ptr(x), _ptr(out))
    return out


# ---- round 6: the reference's FP32 PRODUCTION convolutions / fc (oracle/ref_driver.cpp: ref_conv_f32, ref_vender_fc_f32) ----
REF_F32_IM2COL, REF_F32_CONV1X1, REF_F32_WINOGRAD = 1, 2, 3
REF_F32_IMPL_NAME = {1: "SaberIm2colConv", 2: "SaberConv1X1", 3: "SaberConvWinograd"}


def ref_f32_conv_rule(cin, h, w, cout, k, pad, stride, dil=1, group=1):
    """the implementation SaberConv2D<X86,AK_FLOAT>::init (saber_conv.cpp:49-136) selects, JIT kernels absent"""
    return int(ref().ref_f32_conv_rule(cin, h, w, cout, k, k, pad, pad, stride, stride, dil, dil, group))


def ref_conv_f32(x, w, bias, relu, pad=(0, 0), stride=(1, 1), dil=(1, 1), group=1, impl=0):
    """NCHW f32 convolution through the reference's own x86 objects (impl 0: the dispatcher's rule). Returns (out, impl used)."""
    x = np.ascontiguousarray(x, np.float32)
    w = np.ascontiguousarray(w, np.float32)
    N, Cc, H, W = x.shape
    K, _, kh, kw = w.shape
    oh, ow = conv_out_hw(H, W, kh, kw, pad, stride, dil)
    out = np.zeros((N, K, oh, ow), np.float32)
    b = None if bias is None else np.ascontiguousarray(bias, np.float32)
    used = C.c_int(0)
    rc = ref().ref_conv_f32(int(impl), N, Cc, H, W, K, kh, kw, pad[0], pad[1], stride[0], stride[1], dil[0], dil[1], group,
                            _ptr(x), _ptr(w), _ptr(b), int(relu), _ptr(out), C.byref(used))
    assert rc == 0, rc
    return out, used.value


def ref_fc_f32(x, w_nk, bias):
    """x [m,k] . w[n,k]^T + bias through the reference's Gemm<X86,VENDER_IMPL,float> (ref_driver.cpp: ref_fc_f32 says why not VenderFc)"""
    x = np.ascontiguousarray(x, np.float32)
    w_nk = np.ascontiguousarray(w_nk, np.float32)
    m, k = x.shape
    n = w_nk.shape[0]
    out = np.zeros((m, n), np.float32)
    b = None if bias is None else np.ascontiguousarray(bias, np.float32)
    rc = ref().ref_fc_f32(m, n, k, _ptr(x), _ptr(w_nk), _ptr(b), _ptr(out))
    assert rc == 0, rc
    return out
