"""Host-side mirror of the Saber operator interface for the MI355X target, over the C ABI.

Names and argument meaning follow the reference (saber/funcs/*.h, saber/saber_funcs_param.h):
ConvParam / ActivationParam / EltwiseParam, SaberConv2D / SaberConvEltwise (init = create +
set_weights, dispatch = run), Fc, Gemm, Pooling, Eltwise, Softmax, and an op-list executor that plays
the role of Net::prediction (framework/core/net/net.cpp:417-509). torch is used only to own device
memory and streams; every computation is a call into libsaber_mi355x.so.
"""
import ctypes as C

import numpy as np
import torch

from . import lib as L

_TORCH_DT = {L.F32: torch.float32, L.S8: torch.int8, L.U8: torch.uint8}


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _np_ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def dtype_code(t):
    return {torch.float32: L.F32, torch.int8: L.S8, torch.uint8: L.U8}[t.dtype]


class ConvParam:
    """saber_funcs_param.h:470-581 (+ the EltwiseParam half of ConvEltwiseParam :586-615)."""

    def __init__(self, weight, bias=None, group=1, pad=(0, 0), stride=(1, 1), dilation=(1, 1), relu=False,
                 w_scale=None):
        self.weight = weight          # numpy OIHW, f32 or s8
        self.bias = bias              # numpy f32 [K] or None
        self.w_scale = w_scale        # numpy f32 [K] when weight is s8
        self.group = group
        self.pad, self.stride, self.dilation = pad, stride, dilation
        self.relu = relu
        # fused eltwise
        self.res_mode = L.RES_NONE
        self.res_relu = False
        self.sum_scale = 1.0
        self.negative_slope = 0.0     # ActivationParam::negative_slope of an Active_relu (FP32 convs)
        self.res_dtype = None        # RES_SUM_INPLACE: dtype of the bytes already in y (ConvParam.beta_type); None = out dtype
        self.coeff = (1.0, 1.0)
        self.scale_res = 1.0
        self.res_stride = 0           # RES_ELTWISE: (stride, res_h, res_w) of a 1x1 / stride-s shortcut pooling folded into the
        self.res_hw = (0, 0)          # residual read (saber_hip_conv_desc::res_stride)
        # an activation other than relu behind an FP32 conv: (ActiveType value, negative_slope, coef) - the conv runs without one and
        # the activation follows in place (include/saber_mi355x_impl.h: _post_act; the NV impl's _saber_act)
        self.post_act = None


class SaberConv2D:
    """SaberConv2D<MI355X, AK_INT8|AK_FLOAT> / SaberConvEltwise<MI355X, ...>.

    init(...)  = BaseFunc::init (create + weight quantise/repack, cold path)
    dispatch() = ImplBase::dispatch: enqueue on the current stream, no sync.
    """

    def __init__(self, int8=True):
        self.int8 = int8
        self.h = C.c_void_p()
        self.ws = None

    def init(self, in_shape_nchw, param, in_dtype, out_dtype, in_scale=1.0, out_scale=1.0, in_layout=None,
             out_layout=None):
        lib = L.load()
        n, c, h, w = in_shape_nchw
        k, _, kh, kw = param.weight.shape
        d = L.ConvDesc()
        d.n, d.h, d.w, d.c, d.k, d.kh, d.kw = n, h, w, c, k, kh, kw
        d.pad_h, d.pad_w = param.pad
        d.stride_h, d.stride_w = param.stride
        d.dil_h, d.dil_w = param.dilation
        d.group = param.group
        d.in_dtype, d.out_dtype = in_dtype, out_dtype
        d.in_layout = in_layout if in_layout is not None else (L.NCHW if in_dtype == L.F32 else L.NHWC)
        d.out_layout = out_layout if out_layout is not None else \
            (L.NCHW if (out_dtype == L.F32 and not self.int8) else L.NHWC)
        d.act = L.ACT_RELU if param.relu else L.ACT_NONE
        d.res_mode = param.res_mode
        d.res_act = L.ACT_RELU if param.res_relu else L.ACT_NONE
        d.sum_scale = param.sum_scale
        d.act_negative_slope = float(getattr(param, "negative_slope", 0.0))
        if getattr(param, "res_dtype", None) is not None:
            d.res_has_dtype, d.res_dtype = 1, int(param.res_dtype)
        d.coeff_conv, d.coeff_res = param.coeff
        d.scale_res = param.scale_res
        d.int8_weights = 1 if self.int8 else 0
        if getattr(param, "res_stride", 0) > 1:
            d.res_stride, (d.res_h, d.res_w) = int(param.res_stride), param.res_hw
        self.desc = d
        self.post_act = getattr(param, "post_act", None)
        if self.post_act is not None and (self.int8 or param.res_mode != L.RES_NONE or param.relu or out_dtype != L.F32):
            # (the adaptor's rule: SaberUnImplError - the x86 impls know only relu there)
            raise L.SaberHipError("an activation other than relu needs an FP32 conv with f32 output and no eltwise")
        L.check(lib.saber_hip_conv2d_create(C.byref(d), C.byref(self.h)))
        w_np = np.ascontiguousarray(param.weight)
        w_dt = L.F32 if w_np.dtype == np.float32 else L.S8
        ws = None if param.w_scale is None else np.ascontiguousarray(param.w_scale, np.float32)
        b = None if param.bias is None else np.ascontiguousarray(param.bias, np.float32)
        L.check(lib.saber_hip_conv2d_set_weights(self.h, _np_ptr(w_np), w_dt, _np_ptr(ws), _np_ptr(b),
                                                 float(in_scale), float(out_scale)))
        oh, ow = C.c_int(), C.c_int()
        lib.saber_hip_conv2d_out_shape(self.h, C.byref(oh), C.byref(ow))
        self.out_hw = (oh.value, ow.value)
        nbytes = lib.saber_hip_conv2d_workspace_bytes(self.h)
        self.ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device="cuda") if nbytes else None
        return self

    def out_shape(self):
        """Logical output shape in the op's output layout."""
        d = self.desc
        oh, ow = self.out_hw
        return (d.n, oh, ow, d.k) if d.out_layout == L.NHWC else (d.n, d.k, oh, ow)

    def new_output(self):
        return torch.empty(self.out_shape(), dtype=_TORCH_DT[self.desc.out_dtype], device="cuda")

    def dispatch(self, x, y, res=None):
        L.check(L.load().saber_hip_conv2d_run(self.h, _p(x), _p(y), _p(res), _p(self.ws), _stream()))
        if getattr(self, "post_act", None) is not None:
            active, slope, coef = self.post_act
            activation(active, y, y, slope, coef)
        return y

    def algo(self):
        return L.load().saber_hip_conv2d_algo(self.h).decode()

    def set_tile(self, tile):
        L.check(L.load().saber_hip_conv2d_set_tile(self.h, tile))

    def set_global_pooling(self):
        """fuse the global average Pooling<AK_INT8> of this conv's output (saber_hip_conv2d_set_global_pooling); dispatch_gpool then
        writes both tensors"""
        L.check(L.load().saber_hip_conv2d_set_global_pooling(self.h))
        return self

    def dispatch_gpool(self, x, y, y_pool, res=None):
        L.check(L.load().saber_hip_conv2d_run_gpool(self.h, _p(x), _p(y), _p(res), _p(y_pool), _stream()))
        return y, y_pool

    def tile_id(self):
        """The current block tile (low byte of saber_hip_conv2d_get_tile)."""
        return L.load().saber_hip_conv2d_get_tile(self.h) & 0xff

    def autotune(self, x, y, res=None, iters=5):
        L.check(L.load().saber_hip_conv2d_autotune(self.h, _p(x), _p(y), _p(res), _p(self.ws), _stream(), iters))

    def quantized_weights(self):
        d = self.desc
        wq = np.empty((d.k, d.c // d.group, d.kh, d.kw), np.int8)
        ws = np.empty(d.k, np.float32)
        L.check(L.load().saber_hip_conv2d_get_quantized_weights(self.h, _np_ptr(wq), _np_ptr(ws)))
        return wq, ws

    def __del__(self):
        try:
            if self.h:
                L.load().saber_hip_conv2d_destroy(self.h)
        except Exception:
            pass


class SaberConv2DPooling:
    """SaberConv2DPooling<MI355X, AK_INT8> (saber/funcs/conv_pooling.h, ConvPoolingParam): conv followed by pooling.
    One fused kernel where the library has one (saber_hip_conv2d_set_pooling: the ResNet stem + 3x3/2 max pooling);
    otherwise the conv runs into an inner tensor and the pooling is a second launch, exactly the structure of
    SaberConv2DPooling<X86,AK_FLOAT> (saber_conv_pooling.cpp:13-57). Same bytes either way."""

    def __init__(self, int8=True):
        self.int8 = int8
        self.conv = SaberConv2D(int8)
        self.fused = False
        self.inner = None

    def init(self, in_shape_nchw, conv_param, pool_type, window, stride, pad, in_dtype, out_dtype, in_scale=1.0,
             out_scale=1.0, floor_mode=False, in_layout=None):
        if self.int8:
            self.conv.init(in_shape_nchw, conv_param, in_dtype, out_dtype, in_scale, out_scale, in_layout=in_layout)
        else:   # FP32: NHWC in / out inside an op list
            self.conv.init(in_shape_nchw, conv_param, L.F32, L.F32, in_layout=L.NHWC if in_layout is None else in_layout,
                           out_layout=L.NHWC)
        self.pool = (pool_type, tuple(window), tuple(stride), tuple(pad), floor_mode)
        n = in_shape_nchw[0]
        ch, cw = self.conv.out_hw
        rc = L.load().saber_hip_conv2d_set_pooling(self.conv.h, pool_type, window[0], window[1], stride[0], stride[1],
                                                   pad[0], pad[1], 1 if floor_mode else 0)
        if rc == 0:
            self.fused = True
            oh, ow = C.c_int(), C.c_int()
            L.load().saber_hip_conv2d_out_shape(self.conv.h, C.byref(oh), C.byref(ow))
            self.out_hw = (oh.value, ow.value)
        elif rc == L.UNIMPL:
            self.inner = torch.empty((n, ch, cw, self.conv.desc.k), dtype=_TORCH_DT[out_dtype], device="cuda")
            self.out_hw = pool_out_hw(ch, cw, pad, window, stride, floor_mode)
        else:
            L.check(rc)
        self.h = self.conv.h
        return self

    def new_output(self):
        d = self.conv.desc
        return torch.empty((d.n, self.out_hw[0], self.out_hw[1], d.k), dtype=_TORCH_DT[d.out_dtype], device="cuda")

    def algo(self):
        return self.conv.algo()

    def dispatch(self, x, y):
        if self.fused:
            return self.conv.dispatch(x, y)
        self.conv.dispatch(x, self.inner)
        pt, win, st, pd, _ = self.pool
        d = self.conv.desc
        n, ch, cw, k = self.inner.shape
        if not self.int8:
            L.check(L.load().saber_hip_pool2d_f32(n, ch, cw, k, self.out_hw[0], self.out_hw[1], win[0], win[1], st[0], st[1],
                                                  pd[0], pd[1], pt, L.NHWC, _p(self.inner), _p(y), _stream()))
            return y
        L.check(L.load().saber_hip_pool2d_i8_nhwc(n, ch, cw, k, self.out_hw[0], self.out_hw[1], win[0], win[1], st[0],
                                                  st[1], pd[0], pd[1], pt, d.out_dtype, d.out_dtype, _p(self.inner),
                                                  _p(y), _stream()))
        return y


class SaberConvPair:
    """Two sibling INT8 convolutions over ONE input tensor (same kernel / pad / stride) in one launch
    (saber_hip_conv2d_create_pair): ResNet's stage-entry `branch1` + `branch2a`. Outputs are bit-identical
    to dispatching `a` and `b` one after the other, which is what the reference does (net.cpp:417-509)."""

    def __init__(self, a, b):
        self.a, self.b = a, b          # keep the parents alive (and usable)
        self.h = C.c_void_p()
        L.check(L.load().saber_hip_conv2d_create_pair(a.h, b.h, C.byref(self.h)))

    def dispatch(self, x, ya, yb):
        L.check(L.load().saber_hip_conv2d_run_pair(self.h, _p(x), _p(ya), _p(yb), _stream()))
        return ya, yb

    def algo(self):
        return L.load().saber_hip_conv2d_algo(self.h).decode()

    def set_tile(self, tile):
        L.check(L.load().saber_hip_conv2d_set_tile(self.h, tile))

    def autotune(self, x, ya, yb, iters=5):
        L.check(L.load().saber_hip_conv2d_autotune_pair(self.h, _p(x), _p(ya), _p(yb), _stream(), iters))

    def __del__(self):
        try:
            if self.h:
                L.load().saber_hip_conv2d_destroy(self.h)
        except Exception:
            pass


# ActiveType values (saber/saber_types.h:283-293)
ACTIVE_SIGMOID, ACTIVE_RELU, ACTIVE_TANH, ACTIVE_CLIPPED_RELU, ACTIVE_ELU, ACTIVE_STANH, ACTIVE_PRELU, ACTIVE_GELU, ACTIVE_SWISH = \
    1, 2, 3, 4, 5, 9, 10, 11, 12


def activation(active, x, y=None, negative_slope=0.0, coef=1.0):
    """Activation<MI355X, AK_FLOAT>::dispatch for every type but prelu (saber_hip_activation_f32); y defaults to a new tensor, y = x:
    in place."""
    import torch
    if y is None:
        y = torch.empty_like(x)
    L.check(L.load().saber_hip_activation_f32(int(active), x.numel(), float(negative_slope), float(coef), _p(x), _p(y), _stream()))
    return y


def prelu(x, slope, channels, inner, channel_shared=False, y=None):
    """Active_prelu (saber_hip_prelu_f32): channel = (i / inner) % channels; slope: a device f32 tensor"""
    import torch
    if y is None:
        y = torch.empty_like(x)
    L.check(L.load().saber_hip_prelu_f32(x.numel(), int(channels), int(inner), int(bool(channel_shared)), _p(slope), _p(x), _p(y), _stream()))
    return y


class SaberConvChain:
    """`a` (1x1 conv + fused eltwise, s8) followed by `b` (1x1 conv on a's output) in one launch
    (saber_hip_conv2d_chain_create): ResNet's `branch2c + sum + relu -> next branch2a`. Both outputs hold the bits of
    dispatching `a` and then `b`, which is what the reference does (net.cpp:417-509)."""

    def __init__(self, a, b, conv3x3=None, pair_b=None):
        """conv3x3: the block's 3x3 conv in front of `a` joins the launch (saber_hip_conv2d_chain_create3); dispatch() then
        takes ITS input as x and its own output edge is not written. pair_b (with a stride-2 conv3x3): `b` and `pair_b` are the
        sibling pair reading a's output (saber_hip_conv2d_chain_create3_pair); dispatch() then also takes yc."""
        self.a, self.b, self.c3, self.b2 = a, b, conv3x3, pair_b
        self.h = C.c_void_p()
        if pair_b is not None:
            L.check(L.load().saber_hip_conv2d_chain_create3_pair(conv3x3.h, a.h, b.h, pair_b.h, C.byref(self.h)))
        elif conv3x3 is None:
            L.check(L.load().saber_hip_conv2d_chain_create(a.h, b.h, C.byref(self.h)))
        else:   # b may be None: conv3x3 + `a` only
            L.check(L.load().saber_hip_conv2d_chain_create3(conv3x3.h, a.h, None if b is None else b.h, C.byref(self.h)))

    def dispatch(self, x, res, ya, yb=None, yc=None):
        L.check(L.load().saber_hip_conv2d_chain_run3(self.h, _p(x), _p(res), _p(ya), _p(yb), _p(yc), _stream()))
        return ya, yb

    def set_tile(self, tn):
        L.check(L.load().saber_hip_conv2d_chain_set_tile(self.h, tn))

    def tile(self):
        return L.load().saber_hip_conv2d_chain_get_tile(self.h)

    def __del__(self):
        try:
            if self.h:
                L.load().saber_hip_conv2d_chain_destroy(self.h)
        except Exception:
            pass


class SaberChainStage:
    """A run of 3x3-led C = 256 chains (SaberConvChain with conv3x3 and b) as ONE persistent launch (saber_hip_conv2d_stage_create):
    chain i + 1 reads chain i's two outputs. dispatch(x, res, y1s, y2s): chain 0's inputs, every chain's two output tensors. The
    outputs hold the bits of dispatching the chains (= the operators, net.cpp:417-509) one after the other."""

    def __init__(self, chains):
        self.chains = list(chains)                   # keep them alive: the stage borrows their weights
        arr = (C.c_void_p * len(chains))(*[c.h for c in chains])
        self.h = C.c_void_p()
        L.check(L.load().saber_hip_conv2d_stage_create(arr, len(chains), C.byref(self.h)))

    def dispatch(self, x, res, y1s, y2s):
        n = len(self.chains)
        a = (C.c_void_p * n)(*[t.data_ptr() for t in y1s])
        b = (C.c_void_p * n)(*[t.data_ptr() for t in y2s])
        L.check(L.load().saber_hip_conv2d_stage_run(self.h, _p(x), _p(res), a, b, _stream()))

    def __del__(self):
        try:
            if self.h:
                L.load().saber_hip_conv2d_stage_destroy(self.h)
        except Exception:
            pass


class SaberStemPair:
    """The fused stem conv + max pooling (SaberConv2DPooling, INT8, 64 channels) with the two 1x1 convs `a` and `b` that read the pooled
    tensor in ONE launch (saber_hip_conv2d_stem_pair_create). dispatch(x, y_a, y_b, y_pool=None): y_pool only when something else reads
    the pooled tensor. The outputs hold the bits of dispatching the three operators one after the other (net.cpp:417-509)."""

    def __init__(self, stem, a, b):
        self.ops = (stem, a, b)                      # keep them alive: the object borrows their weights
        self.h = C.c_void_p()
        L.check(L.load().saber_hip_conv2d_stem_pair_create(stem.h, a.h, b.h, C.byref(self.h)))

    def dispatch(self, x, y_a, y_b, y_pool=None):
        stem = self.ops[0]
        ws = stem.ws if hasattr(stem, "ws") else stem.conv.ws      # (a SaberConv2D with a fused pooling, or SaberConv2DPooling)
        L.check(L.load().saber_hip_conv2d_stem_pair_run(self.h, _p(x), _p(y_pool), _p(y_a), _p(y_b), _p(ws), _stream()))

    def __del__(self):
        try:
            if self.h:
                L.load().saber_hip_conv2d_stem_pair_destroy(self.h)
        except Exception:
            pass


class SaberStage:
    """XCD-resident stage (saber_hip_stage_create): `phases` = [(conv, in_slot, out_slot, res_slot | -1), ...] run as ONE
    persistent launch, image i on XCD i % 8; dispatch(tensors) takes the device tensors by slot. Every output slot holds
    the bits of dispatching the convs one after the other (the reference's op-by-op Net::prediction, net.cpp:417-509)."""

    def __init__(self, phases):
        self.convs = [p[0] for p in phases]          # keep the ops alive
        arr = (L.StagePhase * len(phases))()
        for i, (c, a, b, r) in enumerate(phases):
            arr[i].conv, arr[i].in_, arr[i].out, arr[i].res = c.h, a, b, r
        self.h = C.c_void_p()
        L.check(L.load().saber_hip_stage_create(arr, len(phases), C.byref(self.h)))
        self.n = L.load().saber_hip_stage_num_tensors(self.h)

    def dispatch(self, tensors):
        ptrs = (C.c_void_p * self.n)(*[None if t is None else t.data_ptr() for t in tensors[:self.n]])
        L.check(L.load().saber_hip_stage_run(self.h, ptrs, self.n, _stream()))

    def trace(self, arm=False):
        """arm=True: start recording; otherwise the stamps of the last launch as [256, n_phases, 8] (100 MHz ticks)"""
        if arm:
            return L.check_count(L.load().saber_hip_stage_trace(self.h, None, 0))
        n = 256 * len(self.convs) * 8
        out = np.zeros(n, np.uint64)
        L.check_count(L.load().saber_hip_stage_trace(self.h, _np_ptr(out), n))
        return out.reshape(256, len(self.convs), 8)

    def status(self):
        """synchronises; raises if a launch since the last call gave up waiting for its XCD"""
        L.check(L.load().saber_hip_stage_status(self.h))

    def __del__(self):
        try:
            if self.h:
                L.load().saber_hip_stage_destroy(self.h)
        except Exception:
            pass


class SaberFc:
    """Fc<MI355X, AK_FLOAT|AK_INT8> (saber/funcs/fc.h:48-127): out[m,n] = in[m,k] W[n,k]^T + bias."""

    def __init__(self, int8=True):
        self.int8 = int8
        self.h = C.c_void_p()
        self.ws = None

    def init(self, m, n, k, weight, bias, in_dtype, in_scale=1.0, out_scale=1.0, w_scale=None, w_is_kn=False):
        lib = L.load()
        d = L.FcDesc(m, n, k, in_dtype, 1 if self.int8 else 0, 1 if w_is_kn else 0)
        self.desc = d
        L.check(lib.saber_hip_fc_create(C.byref(d), C.byref(self.h)))
        w_np = np.ascontiguousarray(weight)
        w_dt = L.F32 if w_np.dtype == np.float32 else L.S8
        ws = None if w_scale is None else np.ascontiguousarray(w_scale, np.float32)
        b = None if bias is None else np.ascontiguousarray(bias, np.float32)
        L.check(lib.saber_hip_fc_set_weights(self.h, _np_ptr(w_np), w_dt, _np_ptr(ws), _np_ptr(b), float(in_scale),
                                             float(out_scale)))
        nbytes = lib.saber_hip_fc_workspace_bytes(self.h)
        self.ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device="cuda") if nbytes else None
        return self

    def dispatch(self, x, y):
        L.check(L.load().saber_hip_fc_run(self.h, _p(x), _p(y), _p(self.ws), _stream()))
        return y

    def algo(self):
        return L.load().saber_hip_fc_algo(self.h).decode()

    def set_tile(self, tile):
        L.check(L.load().saber_hip_fc_set_tile(self.h, int(tile)))

    def dispatch_softmax(self, x, y, prob):
        """fc + the Softmax over its output, one launch where the INT8 small-batch kernel runs the fc (saber_hip_fc_run_softmax)."""
        L.check(L.load().saber_hip_fc_run_softmax(self.h, _p(x), _p(y), _p(prob), _p(self.ws), _stream()))
        return y, prob

    def dispatch_q(self, xq, y):
        """Input already quantised to s8 with in_scale (saber_hip_fc_run_q)."""
        L.check(L.load().saber_hip_fc_run_q(self.h, _p(xq), _p(y), _stream()))
        return y

    def __del__(self):
        try:
            if self.h:
                L.load().saber_hip_fc_destroy(self.h)
        except Exception:
            pass


class GemmInt8:
    """MklDnnGemm<int8_t|uint8_t, int8_t, int> (saber/funcs/impl/x86/mkl_gemm.cpp:138-256), PACKED mode: row-major
    C[m,n] (int32) = op(A)[m,k] (s8 or u8, device) x op(B)[k,n] (s8, packed at init from a host array). Exact."""

    def __init__(self):
        self.h = C.c_void_p()
        self.ws = None

    def init(self, trans_a, trans_b, m, n, k, b_host, a_dtype=L.S8):
        b_np = np.ascontiguousarray(b_host, np.int8)
        L.check(L.load().saber_hip_gemm_i8_create(int(trans_a), int(trans_b), m, n, k, a_dtype, _np_ptr(b_np),
                                                  C.byref(self.h)))
        self.m, self.n = m, n
        nbytes = L.load().saber_hip_gemm_i8_workspace_bytes(self.h)
        self.ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda") if nbytes else None
        return self

    def dispatch(self, a, c=None):
        if c is None:
            c = torch.empty((self.m, self.n), dtype=torch.int32, device="cuda")
        L.check(L.load().saber_hip_gemm_i8_run(self.h, _p(a), _p(c), _p(self.ws), _stream()))
        return c

    def __del__(self):
        try:
            if self.h:
                L.load().saber_hip_gemm_i8_destroy(self.h)
        except Exception:
            pass


def gemm(trans_a, trans_b, m, n, k, alpha, a, b, beta, c):
    """Gemm<MI355X, SABER_IMPL, float, float>::dispatch (saber/funcs/gemm.h:30-40), raw row-major."""
    L.check(L.load().saber_hip_gemm_f32(int(trans_a), int(trans_b), m, n, k, float(alpha), _p(a), _p(b),
                                        float(beta), _p(c), _stream()))
    return c


def quantize_nchw_to_nhwc(x, scale, out_dtype, c_pad=None):
    n, c, h, w = x.shape
    c_pad = c if c_pad is None else c_pad
    y = torch.empty((n, h, w, c_pad), dtype=_TORCH_DT[out_dtype], device="cuda")
    L.check(L.load().saber_hip_quantize_nchw_to_nhwc(n, c, h, w, c_pad, out_dtype, float(scale), _p(x), _p(y),
                                                     _stream()))
    return y


def dequantize_nhwc_to_nchw(x, scale):
    n, h, w, c = x.shape
    y = torch.empty((n, c, h, w), dtype=torch.float32, device="cuda")
    L.check(L.load().saber_hip_dequantize_nhwc_to_nchw(n, c, h, w, dtype_code(x), float(scale), _p(x), _p(y),
                                                       _stream()))
    return y


def transpose_nchw_to_nhwc(x, c_pad=None):
    n, c, h, w = x.shape
    c_pad = c if c_pad is None else c_pad
    y = torch.empty((n, h, w, c_pad), dtype=torch.float32, device="cuda")
    L.check(L.load().saber_hip_transpose_nchw_to_nhwc_f32(n, c, h, w, c_pad, _p(x), _p(y), _stream()))
    return y


def transpose_nhwc_to_nchw(x, c=None):
    n, h, w, c_pad = x.shape
    c = c_pad if c is None else c
    y = torch.empty((n, c, h, w), dtype=torch.float32, device="cuda")
    L.check(L.load().saber_hip_transpose_nhwc_to_nchw_f32(n, c, h, w, c_pad, _p(x), _p(y), _stream()))
    return y


def quantize_flat_s8(x, scale):
    y = torch.empty(x.shape, dtype=torch.int8, device="cuda")
    L.check(L.load().saber_hip_quantize_flat_s8(x.numel(), float(scale), _p(x), _p(y), _stream()))
    return y


def eltwise_sum(a, b, coeff=(1.0, 1.0), relu=True, scale_a=1.0, scale_b=1.0):
    """Eltwise<MI355X, AK_INT8|AK_FLOAT> sum (EltwiseParam :1077-1140)."""
    y = torch.empty_like(a)
    if a.dtype == torch.int8:
        L.check(L.load().saber_hip_eltwise_sum_i8(a.numel(), _p(a), _p(b), float(scale_a), float(scale_b),
                                                  float(coeff[0]), float(coeff[1]), int(relu), _p(y), _stream()))
    else:
        L.check(L.load().saber_hip_eltwise_sum_f32(a.numel(), _p(a), _p(b), float(coeff[0]), float(coeff[1]),
                                                   int(relu), _p(y), _stream()))
    return y


def pool_out_dim(inp, pad, win, stride, floor_mode=False, any_pad=None):
    """Pooling output size along one dimension. any_pad: pad_h > 0 or pad_w > 0 - the reference clips the last window of BOTH
    dimensions back inside the input whenever EITHER pad is non-zero (saber/funcs/pooling.h:113-120); None = this pad only."""
    ap = int(pad > 0) if any_pad is None else int(bool(any_pad))
    return L.load().saber_hip_pool_out_dim2(inp, pad, win, stride, int(floor_mode), ap)


def pool_out_hw(h, w, pad, window, stride, floor_mode=False):
    ap = pad[0] > 0 or pad[1] > 0
    return (pool_out_dim(h, pad[0], window[0], stride[0], floor_mode, ap), pool_out_dim(w, pad[1], window[1], stride[1], floor_mode, ap))


def pooling_i8(x, window, stride, pad, pool_type, out_dtype=None, global_pooling=False, floor_mode=False):
    """Pooling<MI355X, AK_INT8> on NHWC s8/u8 (PoolingParam :2087)."""
    n, h, w, c = x.shape
    if global_pooling:
        window, stride, pad, oh, ow = (h, w), (h, w), (0, 0), 1, 1
    else:
        oh, ow = pool_out_hw(h, w, pad, window, stride, floor_mode)
    od = dtype_code(x) if out_dtype is None else out_dtype
    y = torch.empty((n, oh, ow, c), dtype=_TORCH_DT[od], device="cuda")
    L.check(L.load().saber_hip_pool2d_i8_nhwc(n, h, w, c, oh, ow, window[0], window[1], stride[0], stride[1],
                                              pad[0], pad[1], pool_type, dtype_code(x), od, _p(x), _p(y), _stream()))
    return y


def pooling_f32(x, window, stride, pad, pool_type, layout=L.NCHW, global_pooling=False, floor_mode=False):
    if layout == L.NCHW:
        n, c, h, w = x.shape
    else:
        n, h, w, c = x.shape
    if global_pooling:
        window, stride, pad, oh, ow = (h, w), (h, w), (0, 0), 1, 1
    else:
        oh, ow = pool_out_hw(h, w, pad, window, stride, floor_mode)
    shape = (n, c, oh, ow) if layout == L.NCHW else (n, oh, ow, c)
    y = torch.empty(shape, dtype=torch.float32, device="cuda")
    L.check(L.load().saber_hip_pool2d_f32(n, h, w, c, oh, ow, window[0], window[1], stride[0], stride[1], pad[0],
                                          pad[1], pool_type, layout, _p(x), _p(y), _stream()))
    return y


def pooling_f32_from_i8(x, scale, window, stride, pad, pool_type, global_pooling=False, floor_mode=False,
                        q_scale=None):
    """Pooling<MI355X, AK_FLOAT> fed an s8/u8 NHWC tensor: dequantise on entry, pool, f32 NCHW out.
    q_scale: also return the s8 quantisation of the result (the next INT8 op's quantise-on-entry, fused)."""
    n, h, w, c = x.shape
    if global_pooling:
        window, stride, pad, oh, ow = (h, w), (h, w), (0, 0), 1, 1
    else:
        oh, ow = pool_out_hw(h, w, pad, window, stride, floor_mode)
    y = torch.empty((n, c, oh, ow), dtype=torch.float32, device="cuda")
    if q_scale is not None:
        yq = torch.empty((n, c, oh, ow), dtype=torch.int8, device="cuda")
        L.check(L.load().saber_hip_pool2d_f32_from_i8_q(n, h, w, c, oh, ow, window[0], window[1], stride[0], stride[1],
                                                        pad[0], pad[1], pool_type, dtype_code(x), float(scale), _p(x),
                                                        _p(y), float(q_scale), _p(yq), _stream()))
        return y, yq
    L.check(L.load().saber_hip_pool2d_f32_from_i8(n, h, w, c, oh, ow, window[0], window[1], stride[0], stride[1],
                                                  pad[0], pad[1], pool_type, dtype_code(x), float(scale), _p(x), _p(y),
                                                  _stream()))
    return y


def softmax(x):
    rows, cols = x.shape[0], x.numel() // x.shape[0]
    y = torch.empty_like(x)
    L.check(L.load().saber_hip_softmax_f32(rows, cols, _p(x), _p(y), _stream()))
    return y


class Net:
    """Op-list executor: the device half of Net<T,P,R>::prediction (net.cpp:417-509). Tensors are edge
    buffers in one arena; ops run in insertion order on one stream; `capture()` turns the launch
    sequence into a hipGraph."""

    def __init__(self):
        self.h = C.c_void_p()
        L.check(L.load().saber_hip_net_create(C.byref(self.h)))
        self.keep = []          # conv/fc python objects whose handles the net references
        self.tensors = {}       # name -> (id, shape, torch dtype)
        self.finalized = False

    def add_tensor(self, name, shape, dtype_code_):
        nbytes = int(np.prod(shape)) * (4 if dtype_code_ == L.F32 else 1)
        tid = L.load().saber_hip_net_add_tensor(self.h, nbytes)
        self.tensors[name] = (tid, tuple(shape), _TORCH_DT[dtype_code_])
        return tid

    def tid(self, name):
        return self.tensors[name][0]

    def add_conv(self, conv, x, y, res=None):
        self.keep.append(conv)
        rc = L.load().saber_hip_net_add_conv(self.h, conv.h, self.tid(x), self.tid(y), -1 if res is None else self.tid(res))
        if rc < 0:
            L.check(rc)
        return rc

    def add_conv_pair(self, pair, x, ya, yb):
        self.keep.append(pair)
        return self._chk(L.load().saber_hip_net_add_conv_pair(self.h, pair.h, self.tid(x), self.tid(ya), self.tid(yb)))

    def add_fc(self, fc, x, y):
        self.keep.append(fc)
        rc = L.load().saber_hip_net_add_fc(self.h, fc.h, self.tid(x), self.tid(y))
        if rc < 0:
            L.check(rc)
        return rc

    def _chk(self, rc):
        if rc < 0:
            L.check(rc)
        return rc

    def add_quantize(self, n, c, h, w, c_pad, out_dtype, scale, x, y):
        return self._chk(L.load().saber_hip_net_add_quantize(self.h, n, c, h, w, c_pad, out_dtype, float(scale),
                                                             self.tid(x), self.tid(y)))

    def add_dequantize(self, n, c, h, w, in_dtype, scale, x, y):
        return self._chk(L.load().saber_hip_net_add_dequantize(self.h, n, c, h, w, in_dtype, float(scale),
                                                               self.tid(x), self.tid(y)))

    def add_transpose_in(self, n, c, h, w, c_pad, x, y):
        return self._chk(L.load().saber_hip_net_add_transpose_in_f32(self.h, n, c, h, w, c_pad, self.tid(x), self.tid(y)))

    def add_eltwise_i8(self, count, sa, sb, c0, c1, relu, a, b, y):
        return self._chk(L.load().saber_hip_net_add_eltwise_i8(self.h, count, float(sa), float(sb), float(c0),
                                                               float(c1), int(relu), self.tid(a), self.tid(b), self.tid(y)))

    def add_eltwise_f32(self, count, c0, c1, relu, a, b, y):
        return self._chk(L.load().saber_hip_net_add_eltwise_f32(self.h, count, float(c0), float(c1), int(relu),
                                                                self.tid(a), self.tid(b), self.tid(y)))

    def add_pool_i8(self, n, h, w, c, oh, ow, win, stride, pad, ptype, in_dtype, out_dtype, x, y):
        return self._chk(L.load().saber_hip_net_add_pool_i8(self.h, n, h, w, c, oh, ow, win[0], win[1], stride[0],
                                                            stride[1], pad[0], pad[1], ptype, in_dtype, out_dtype,
                                                            self.tid(x), self.tid(y)))

    def add_pool_f32(self, n, h, w, c, oh, ow, win, stride, pad, ptype, layout, x, y):
        return self._chk(L.load().saber_hip_net_add_pool_f32(self.h, n, h, w, c, oh, ow, win[0], win[1], stride[0],
                                                             stride[1], pad[0], pad[1], ptype, layout, self.tid(x),
                                                             self.tid(y)))

    def add_pool_f32_from_i8(self, n, h, w, c, oh, ow, win, stride, pad, ptype, in_dtype, scale, x, y):
        return self._chk(L.load().saber_hip_net_add_pool_f32_from_i8(self.h, n, h, w, c, oh, ow, win[0], win[1],
                                                                     stride[0], stride[1], pad[0], pad[1], ptype,
                                                                     in_dtype, float(scale), self.tid(x), self.tid(y)))

    def add_pool_f32_from_i8_q(self, n, h, w, c, oh, ow, win, stride, pad, ptype, in_dtype, scale, x, y, q_scale, yq):
        """pool + the s8 quantisation of its result (the next INT8 op's quantise-on-entry) in one launch."""
        return self._chk(L.load().saber_hip_net_add_pool_f32_from_i8_q(
            self.h, n, h, w, c, oh, ow, win[0], win[1], stride[0], stride[1], pad[0], pad[1], ptype, in_dtype,
            float(scale), self.tid(x), self.tid(y), float(q_scale), self.tid(yq)))

    def add_fc_q(self, fc, xq, y):
        """INT8 fc reading an input already quantised with its in_scale (see add_pool_f32_from_i8_q)."""
        self.keep.append(fc)
        return self._chk(L.load().saber_hip_net_add_fc_q(self.h, fc.h, self.tid(xq), self.tid(y)))

    def add_softmax(self, rows, cols, x, y):
        return self._chk(L.load().saber_hip_net_add_softmax(self.h, rows, cols, self.tid(x), self.tid(y)))

    def set_lane(self, op_index, lane):
        L.check(L.load().saber_hip_net_set_lane(self.h, op_index, lane))

    def optimize(self, flags=15):
        """Executor-level fusions in C++ (saber_hip_net_optimize) on an op list added unfused; before finalize().
        Returns the number of launches removed."""
        rc = L.load().saber_hip_net_optimize(self.h, int(flags))
        if rc < 0:
            L.check(rc)
        return rc

    def finalize(self):
        L.check(L.load().saber_hip_net_finalize(self.h))
        self.finalized = True

    def tensor(self, name):
        """A torch view (no copy) of an edge tensor inside the arena."""
        tid, shape, dt = self.tensors[name]
        ptr = L.load().saber_hip_net_tensor_ptr(self.h, tid)
        n = int(np.prod(shape))
        nbytes = n * (4 if dt == torch.float32 else 1)
        # wrap raw device memory via the __cuda_array_interface__ protocol
        holder = _RawCuda(ptr, nbytes)
        return torch.as_tensor(holder, device="cuda").view(dt).view(shape)

    def num_ops(self):
        return L.load().saber_hip_net_num_ops(self.h)

    def unwritten(self, name):
        """True when the named tensor is the output edge of a 3x3 conv that currently runs inside a conv3x3 + chain launch"""
        return bool(L.load().saber_hip_net_tensor_unwritten(self.h, self.tensors[name][0]))

    def num_launches(self):
        """kernel launches per forward (ops absorbed into a conv1x1-chain launch do not count)"""
        return L.load().saber_hip_net_num_launches(self.h)

    def op_name(self, i):
        return L.load().saber_hip_net_op_name(self.h, i).decode()

    def run(self):
        L.check(L.load().saber_hip_net_run(self.h, _stream()))

    def run_op(self, i):
        L.check(L.load().saber_hip_net_run_op(self.h, i, _stream()))

    def capture(self):
        """Stream capture is not permitted on the legacy default stream: capture on a side stream
        (ordered after the current stream); the instantiated graph can be replayed on any stream."""
        cur = torch.cuda.current_stream()
        if cur.cuda_stream == 0:
            side = torch.cuda.Stream()
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                L.check(L.load().saber_hip_net_capture(self.h, _stream()))
            cur.wait_stream(side)
        else:
            L.check(L.load().saber_hip_net_capture(self.h, _stream()))

    def replay(self):
        L.check(L.load().saber_hip_net_replay(self.h, _stream()))

    def autotune(self, iters=5):
        L.check(L.load().saber_hip_net_autotune(self.h, _stream(), iters))

    def choices(self):
        """The kernel selection of every op (saber_hip_conv2d_get_tile encoding, 0 = none)."""
        lib = L.load()
        return [int(lib.saber_hip_net_get_choice(self.h, i)) for i in range(self.num_ops())]

    def set_choices(self, choices):
        lib = L.load()
        assert len(choices) == self.num_ops()
        for i, c in enumerate(choices):
            L.check(lib.saber_hip_net_set_choice(self.h, i, int(c)))

    def status(self):
        """after a completed pass: raises SaberHipError when one of its cooperative launches failed (those sites then launch block by
        block: run the pass again) - saber_hip_net_status"""
        L.check(L.load().saber_hip_net_status(self.h))

    def coop_fallbacks(self):
        """cooperative launches of this net that reported a failed pass so far (saber_hip_net_coop_fallbacks)"""
        return L.load().saber_hip_net_coop_fallbacks(self.h)

    def stages(self):
        """[(op index, blocks, selected)] of the ops that head a stage (saber_hip_net_optimize flag 256)"""
        lib = L.load()
        out = []
        for i in range(self.num_ops()):
            n = int(lib.saber_hip_net_stage_blocks(self.h, i))
            if n:
                out.append((i, n, bool(int(lib.saber_hip_net_get_choice(self.h, i)) >> 30 & 1)))
        return out

    def select_stages(self, on):
        """every stage of the net on / off (what saber_hip_net_autotune decides per stage by timing)"""
        ch = self.choices()
        for i, _, _ in self.stages():
            ch[i] = (ch[i] | (1 << 30)) if on else (ch[i] & ~(1 << 30))
        self.set_choices(ch)

    def time_ops(self, iters=20):
        out = (C.c_float * self.num_ops())()
        L.check(L.load().saber_hip_net_time_ops(self.h, _stream(), iters, out))
        return list(out)

    def time_op_in_pass(self, index, iters=20):
        """one op's launch duration inside an otherwise untimed eager pass (two events around that launch only)"""
        out = C.c_float()
        L.check(L.load().saber_hip_net_time_op_in_pass(self.h, _stream(), int(index), int(iters), C.byref(out)))
        return float(out.value)

    def time_pass(self, iters=20):
        """Per-op microseconds INSIDE a forward pass (one event after every launch; saber_hip_net_time_pass)."""
        out = (C.c_float * self.num_ops())()
        L.check(L.load().saber_hip_net_time_pass(self.h, _stream(), iters, out))
        return list(out)

    def op_work(self, index):
        """(algorithmic bytes, flops) of one launch of op `index` (saber_hip_net_op_work)."""
        b, f = C.c_double(), C.c_double()
        L.check(L.load().saber_hip_net_op_work(self.h, int(index), C.byref(b), C.byref(f)))
        return b.value, f.value

    def arena_bytes(self):
        return L.load().saber_hip_net_arena_bytes(self.h)

    def compact(self, keep=()):
        """saber_hip_net_compact_arena: edges of disjoint lifetimes share memory from now on (the reference's MemoryScheduler role);
        inputs, outputs and the named `keep` tensors stay readable, every other edge holds garbage after a pass. Views from
        tensor() taken earlier are invalid: fetch them again. Call after autotune / set_choices / the last per-edge check."""
        ids = (C.c_int * max(1, len(keep)))(*[self.tensors[k][0] for k in keep])
        L.check(L.load().saber_hip_net_compact_arena(self.h, ids, len(keep)))
        return self.arena_bytes()

    def compacted(self):
        return bool(L.load().saber_hip_net_arena_compacted(self.h))

    # ---- caller-owned tensors / captured lists (saber_hip_capture_begin / _end) ----
    def tensor_of_ptr(self, t):
        """captured lists: id of the newest tensor the pass saw at torch tensor `t`'s address (-1: none)"""
        return int(L.load().saber_hip_net_tensor_of_ptr(self.h, t.data_ptr()))

    def bind(self, tid, t):
        """tensor `tid` lives in the caller's torch tensor `t` instead of the arena (the caller keeps `t` alive)"""
        self.keep.append(t)
        L.check(L.load().saber_hip_net_bind_tensor(self.h, int(tid), t.data_ptr()))

    def num_tensors(self):
        return int(L.load().saber_hip_net_num_tensors(self.h))


class Capture:
    """`with Capture() as cap:` every dispatch / streaming-op call of this thread is RECORDED instead of launched (the C-ABI
    analogue of hipStreamBeginCapture; what the MI355X target does with the reference's Net::prediction loop,
    integration/mi355x/framework/mi355x_net_plan.h). Afterwards `cap.net` is the op list as a Net (not finalized): tensors are
    the call's device pointers renamed SSA-style, addresses read before written are bound to the caller's memory."""

    def __init__(self, keep=()):
        self.net = None
        self._keep = list(keep)

    def __enter__(self):
        L.check(L.load().saber_hip_capture_begin())
        return self

    def __exit__(self, et, ev, tb):
        h = C.c_void_p()
        rc = L.load().saber_hip_capture_end(C.byref(h))
        if et is None:
            L.check(rc)
            net = Net.__new__(Net)
            net.h, net.keep, net.tensors, net.finalized = h, list(self._keep), {}, False
            self.net = net
        elif h:
            L.load().saber_hip_net_destroy(h)
        return False

    def __del__(self):
        try:
            if self.h:
                L.load().saber_hip_net_destroy(self.h)
        except Exception:
            pass


class _RawCuda:
    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (int(ptr), False),
                                         "version": 2}
