"""CPU oracle forward of the synthetic post-fusion op lists (anakin_amd/workloads.py), op by op with
oracle/saber_oracle.c. TEST INFRASTRUCTURE ONLY (parity tests, smoke, bench cpu_baseline)."""
import numpy as np

from . import oracle as O

F32, S8, U8 = O.F32, O.S8, O.U8


def prepare_int8(model):
    """Cold path, once per model: per-layer weight scales + quantised weights (the `init` work of the ops)."""
    prep = {}
    for l in model["spec"]:
        if l["kind"] in ("conv", "fc"):
            w, _ = model["params"][l["name"]]
            ws = O.weight_scales(w)
            prep[l["name"]] = (ws, O.quant_weights(w, ws))
    return prep


def set_threads(n):
    """OpenMP thread count of the oracle library (bench.py cpu_baseline states the number it used)."""
    import ctypes
    ctypes.CDLL("libgomp.so.1").omp_set_num_threads(int(n))


def run_int8(model, scales, x, keep=True, prep=None):
    """x: f32 NCHW batch. Returns {edge name: numpy tensor} (8-bit edges NHWC) following exactly the
    unfused reference op list (graph.cpp:423-436: conv+eltwise fusion is off for INT8)."""
    scales = dict(scales)
    prep = prep if prep is not None else prepare_int8(model)
    t = {"data": x}
    dt = {"data": F32}
    for l in model["spec"]:
        kd, nm = l["kind"], l["name"]
        if kd == "conv":
            w, b = model["params"][nm]
            src = t[l["src"]]
            if dt[l["src"]] == F32:  # quantise on entry (reorder_nhwc_nchw, saber_conv.cpp:308)
                src = O.quant_nchw_to_nhwc(src, scales[l["src"]], S8)
                in_dt = S8
            else:
                in_dt = dt[l["src"]]
            odt = l.get("odt", U8 if l["relu"] else S8)   # workloads.framework_spec: conv1's output follows its consumer
            ws, wq = prep[nm]
            bp, sc = O.conv_i8_prepare(ws, b, scales[l["src"]], scales[nm], in_dt, odt)
            t[nm] = O.conv_i8(src, wq, bp, sc, odt, l["relu"], (l["pad"],) * 2, (l["stride"],) * 2)
            dt[nm] = odt
        elif kd == "pool":
            t[nm] = O.pool_i8_nhwc(t[l["src"]], (l["win"],) * 2, (l["stride"],) * 2, (l["pad"],) * 2, l["type"],
                                   floor_mode=l.get("floor", False))
            dt[nm] = dt[l["src"]]
            scales[nm] = scales[l["src"]]
        elif kd == "eltwise":
            c = np.float32(1.0 / scales[nm])
            t[nm] = O.eltwise_i8(t[l["a"]], t[l["b"]], scales[l["a"]], scales[l["b"]], c, c, l["relu"])
            dt[nm] = S8
        elif kd == "gpool" and l.get("int8"):
            # INT8 global average pooling, s8 -> s8, the output inherits the input's scale (saber_pooling.cpp:571-582)
            t[nm] = O.pool_i8_nhwc(t[l["src"]], None, None, None, 1, global_pool=True)
            dt[nm] = dt[l["src"]]
            scales[nm] = scales[l["src"]]
        elif kd == "gpool":
            deq = O.dequant_nhwc_to_nchw(t[l["src"]], scales[l["src"]])
            t[nm] = O.pool_f32_nchw(deq, None, None, None, 1, global_pool=True)
            dt[nm] = F32
        elif kd == "fc":
            w, b = model["params"][nm]
            ws, wq = prep[nm]
            xin = t[l["src"]].reshape(t[l["src"]].shape[0], -1)
            # f32 input: quantised on entry; s8 input: used as it is (vender_fc.cpp:254-262, mkl_packed_int8_gemm.cpp:52-57)
            xq = xin if dt[l["src"]] == S8 else O.quant_flat_s8(xin, scales[l["src"]])
            t[nm] = O.fc_i8(xq, wq, ws, scales[l["src"]], b)
            dt[nm] = F32
        elif kd == "softmax":
            t[nm] = O.softmax_f32(t[l["src"]])
            dt[nm] = F32
    return t


def run_fp32(model, x):
    """FP32 forward, NCHW, naive reference order (conv_basic_check)."""
    t = {"data": x}
    for l in model["spec"]:
        kd, nm = l["kind"], l["name"]
        if kd == "conv":
            w, b = model["params"][nm]
            t[nm] = O.conv_f32_nchw(t[l["src"]], w, b, l["relu"], (l["pad"],) * 2, (l["stride"],) * 2)
        elif kd == "pool":
            t[nm] = O.pool_f32_nchw(t[l["src"]], (l["win"],) * 2, (l["stride"],) * 2, (l["pad"],) * 2, l["type"],
                                    floor_mode=l.get("floor", False))
        elif kd == "eltwise":
            t[nm] = O.eltwise_f32(t[l["a"]], t[l["b"]], 1.0, 1.0, l["relu"])
        elif kd == "gpool":
            t[nm] = O.pool_f32_nchw(t[l["src"]], None, None, None, 1, global_pool=True)
        elif kd == "fc":
            w, b = model["params"][nm]
            y = O.fc_f32(t[l["src"]].reshape(t[l["src"]].shape[0], -1), w, b)
            t[nm] = np.maximum(y, 0) if l.get("relu") else y
        elif kd == "softmax":
            t[nm] = O.softmax_f32(t[l["src"]])
    return t


class RefNet:
    """The ResNet INT8 op list run by the REFERENCE'S OWN compiled x86 objects (oracle/_ref, ref_net_* in
    oracle/ref_driver.cpp): GemmX8S8S32XConv per conv (init once, dispatch per forward), SaberEltwise<X86,AK_INT8>,
    PackedMKLInt8Gemm for the fc. Used by bench.py's cpu_baseline ("kind": "reference") and pinned against
    run_int8 above (same logits, bit for bit) in tests/test_oracle_vs_ref.py. TEST / BASELINE INFRASTRUCTURE ONLY."""

    def __init__(self, model, scales, batch, hw=224):
        import ctypes as C
        self.C = C
        self.R = R = O.ref()
        R.ref_net_new.restype = C.c_void_p
        R.ref_net_time_ms.restype = C.c_double
        R.ref_net_tensor.argtypes = [C.c_void_p] + [C.c_int] * 5 + [C.c_float]
        R.ref_net_conv.argtypes = [C.c_void_p] + [C.c_int] * 8 + [C.c_void_p, C.c_void_p]
        R.ref_net_eltwise.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int]
        R.ref_net_maxpool.argtypes = [C.c_void_p] + [C.c_int] * 5
        R.ref_net_gpool_fc.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_float]
        R.ref_net_avgpool_fc_s8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        R.ref_net_run.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        R.ref_net_read.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        R.ref_net_time_ms.argtypes = [C.c_void_p, C.c_int, C.c_int]
        R.ref_net_free.argtypes = [C.c_void_p]
        self.h = C.c_void_p(R.ref_net_new())
        scales = dict(scales)
        B = batch
        self.ids, self.shape, self.dt = {}, {}, {}
        self.keep = []

        def tensor(name, c, s, dtype, scale):
            self.ids[name] = R.ref_net_tensor(self.h, B, c, s, s, dtype, float(scale))
            self.shape[name], self.dt[name] = (c, s), dtype
        tensor("data", 3, hw, F32, scales["data"])
        spec = model["spec"]
        for li, l in enumerate(spec):
            kd, nm = l["kind"], l["name"]
            if kd == "conv":
                cin, hin = self.shape[l["src"]]
                ho = (hin + 2 * l["pad"] - l["k"]) // l["stride"] + 1
                w, b = model["params"][nm]
                tensor(nm, l["cout"], ho, l.get("odt", U8 if l["relu"] else S8), scales[nm])
                w = np.ascontiguousarray(w, np.float32)
                b = np.ascontiguousarray(b, np.float32)
                self.keep += [w, b]
                rc = R.ref_net_conv(self.h, self.ids[l["src"]], self.ids[nm], l["cout"], cin, l["k"], l["pad"], l["stride"],
                                    int(l["relu"]), w.ctypes.data, b.ctypes.data)
                assert rc == 0, (nm, rc)
            elif kd == "pool":
                c, hin = self.shape[l["src"]]
                ho = O.pool_out_dim(hin, l["pad"], l["win"], l["stride"], l.get("floor", False))
                scales[nm] = scales[l["src"]]
                tensor(nm, c, ho, self.dt[l["src"]], scales[nm])
                assert l["type"] == 0
                R.ref_net_maxpool(self.h, self.ids[l["src"]], self.ids[nm], l["win"], l["stride"], l["pad"])
            elif kd == "eltwise":
                c, s = self.shape[l["a"]]
                tensor(nm, c, s, S8, scales[nm])
                rc = R.ref_net_eltwise(self.h, self.ids[l["a"]], self.ids[l["b"]], self.ids[nm], 1.0 / scales[nm], int(l["relu"]))
                assert rc == 0
            elif kd == "gpool":
                nxt = spec[li + 1]
                assert nxt["kind"] == "fc" and nxt["src"] == nm
                w, b = model["params"][nxt["name"]]
                w = np.ascontiguousarray(w, np.float32)
                b = np.ascontiguousarray(b, np.float32)
                self.keep += [w, b]
                self.n_out = nxt["cout"]
                self.ids[nxt["name"]] = R.ref_net_tensor(self.h, B, self.n_out, 1, 1, F32, 1.0)
                if l.get("int8"):       # workloads.framework_spec: INT8 average pooling, the fc reads its s8 result
                    c, _ = self.shape[l["src"]]
                    tensor(nm, c, 1, S8, scales[l["src"]])
                    rc = R.ref_net_avgpool_fc_s8(self.h, self.ids[l["src"]], self.ids[nm], self.ids[nxt["name"]], self.n_out,
                                                 w.ctypes.data, b.ctypes.data)
                else:
                    rc = R.ref_net_gpool_fc(self.h, self.ids[l["src"]], self.ids[nxt["name"]], self.n_out, w.ctypes.data,
                                            b.ctypes.data, float(scales[nm]))
                assert rc == 0
                self.out_name = nxt["name"]
            # fc handled with gpool; softmax is not part of the timed reference list (a 1000-element row)
        self.batch, self.hw = B, hw

    def run(self, x):
        x = np.ascontiguousarray(x, np.float32)
        out = np.empty((self.batch, self.n_out), np.float32)
        rc = self.R.ref_net_run(self.h, self.ids["data"], x.ctypes.data, self.ids[self.out_name], out.ctypes.data)
        assert rc == 0, rc
        return out

    def read(self, name):
        c, s = self.shape[name]
        a = np.empty((self.batch, s, s, c), np.uint8 if self.dt[name] == U8 else np.int8)
        self.R.ref_net_read(self.h, self.ids[name], a.ctypes.data)
        return a

    def time_ms(self, warmup=10, iters=200):
        return float(self.R.ref_net_time_ms(self.h, warmup, iters))

    def __del__(self):
        try:
            self.R.ref_net_free(self.h)
        except Exception:
            pass


def ref_set_threads(n):
    O.ref().ref_set_threads(int(n))


class RefNetF32:
    """BASELINE.json configs[0]: the FP32 op list (ResNet50 / VGG16, NCHW f32) run by the REFERENCE'S OWN compiled x86 objects
    (oracle/_ref; ref_net_conv_f32 / ref_net_pool_f32 / ref_net_fc_f32 in oracle/ref_driver.cpp). Every convolution runs on the
    implementation SaberConv2D<X86,AK_FLOAT>::init (saber/funcs/impl/x86/saber_conv.cpp:49-136) selects for it when the xbyak JIT
    kernels are absent - SaberConvWinograd (3x3 / stride 1, maps >= 12), SaberConv1X1 (1x1 / stride 1), SaberIm2colConv (the rest:
    7x7 / stride 2, strided 1x1, 3x3 at 7x7) - or on `impl` when forced (1 = im2col everywhere it applies). `branch2c` + eltwise
    run as the ConvEltwise operator does (saber_conv_eltwise.cpp:68-151: SaberConv1X1 with beta = 1 onto the shortcut's buffer).
    Pooling is restated (its x86 source needs xbyak); the fc is the reference's float Gemm object + bias (ref_fc_f32 says why).
    Used by bench.py's cpu_baseline for the FP32 configurations and by tests/test_oracle_vs_ref.py. TEST / BASELINE INFRASTRUCTURE ONLY."""

    def __init__(self, model, batch, hw=224, impl=0):
        import ctypes as C
        self.R = R = O.ref()
        R.ref_net_new.restype = C.c_void_p
        R.ref_net_time_ms.restype = C.c_double
        R.ref_net_tensor.argtypes = [C.c_void_p] + [C.c_int] * 5 + [C.c_float]
        R.ref_net_conv_f32.argtypes = [C.c_void_p] + [C.c_int] * 8 + [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        R.ref_net_pool_f32.argtypes = [C.c_void_p] + [C.c_int] * 6
        R.ref_net_fc_f32.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        R.ref_net_run.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        R.ref_net_read.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        R.ref_net_time_ms.argtypes = [C.c_void_p, C.c_int, C.c_int]
        R.ref_net_free.argtypes = [C.c_void_p]
        self.h = C.c_void_p(R.ref_net_new())
        B = self.batch = batch
        self.ids, self.shape, self.keep, self.impl_of = {}, {}, [], {}
        alias = {}

        def T(n):
            return alias.get(n, n)

        def tensor(name, c, s):
            self.ids[name] = R.ref_net_tensor(self.h, B, c, s, s, F32, 1.0)
            self.shape[name] = (c, s)
        tensor("data", 3, hw)
        spec = model["spec"]
        by = {l["name"]: l for l in spec}
        for l in spec:
            kd, nm = l["kind"], l["name"]
            if kd == "conv":
                cin, hin = self.shape[T(l["src"])]
                ho = (hin + 2 * l["pad"] - l["k"]) // l["stride"] + 1
                w, b = model["params"][nm]
                w = np.ascontiguousarray(w, np.float32)
                b = np.ascontiguousarray(b, np.float32)
                self.keep += [w, b]
                if "eltwise" in l:           # ConvEltwise: accumulate onto the other eltwise input, relu of the eltwise
                    el = by[l["eltwise"]]
                    dst, res, relu = T(el["b"]), 1, int(el["relu"])
                    alias[el["name"]] = dst
                else:
                    tensor(nm, l["cout"], ho)
                    dst, res, relu = nm, 0, int(l["relu"])
                force = impl if (impl and not (impl == 1 and res)) else 0       # (the fused residual needs SaberConv1X1's beta = 1)
                rc = R.ref_net_conv_f32(self.h, self.ids[T(l["src"])], self.ids[dst], l["cout"], cin, l["k"], l["pad"], l["stride"],
                                        relu, w.ctypes.data, b.ctypes.data, res, force)
                assert rc > 0, (nm, rc)
                self.impl_of[nm] = rc
            elif kd == "pool":
                c, hin = self.shape[T(l["src"])]
                ho = O.pool_out_dim(hin, l["pad"], l["win"], l["stride"], l.get("floor", False))
                tensor(nm, c, ho)
                R.ref_net_pool_f32(self.h, self.ids[T(l["src"])], self.ids[nm], l["win"], l["stride"], l["pad"], l["type"])
            elif kd == "gpool":
                c, hin = self.shape[T(l["src"])]
                tensor(nm, c, 1)
                R.ref_net_pool_f32(self.h, self.ids[T(l["src"])], self.ids[nm], hin, hin, 0, 1)
            elif kd == "fc":
                w, b = model["params"][nm]
                w = np.ascontiguousarray(w, np.float32)
                b = np.ascontiguousarray(b, np.float32)
                self.keep += [w, b]
                tensor(nm, l["cout"], 1)
                rc = R.ref_net_fc_f32(self.h, self.ids[T(l["src"])], self.ids[nm], l["cout"], w.ctypes.data, b.ctypes.data,
                                      int(bool(l.get("relu"))))
                assert rc == 0, (nm, rc)
                self.out_name, self.n_out = nm, l["cout"]
            # eltwise: fused into branch2c above; softmax is not part of the timed reference list
        self.alias = alias

    def impl_counts(self):
        c = {}
        for v in self.impl_of.values():
            c[O.REF_F32_IMPL_NAME[v]] = c.get(O.REF_F32_IMPL_NAME[v], 0) + 1
        return c

    def run(self, x):
        x = np.ascontiguousarray(x, np.float32)
        out = np.empty((self.batch, self.n_out), np.float32)
        rc = self.R.ref_net_run(self.h, self.ids["data"], x.ctypes.data, self.ids[self.out_name], out.ctypes.data)
        assert rc == 0, rc
        return out

    def read(self, name):
        name = self.alias.get(name, name)
        c, s = self.shape[name]
        a = np.empty((self.batch, c, s, s), np.float32)
        self.R.ref_net_read(self.h, self.ids[name], a.ctypes.data)
        return a

    def time_ms(self, warmup=10, iters=200):
        return float(self.R.ref_net_time_ms(self.h, warmup, iters))

    def __del__(self):
        try:
            self.R.ref_net_free(self.h)
        except Exception:
            pass
