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
            odt = U8 if l["relu"] else S8
            ws, wq = prep[nm]
            bp, sc = O.conv_i8_prepare(ws, b, scales[l["src"]], scales[nm], in_dt, odt)
            t[nm] = O.conv_i8(src, wq, bp, sc, odt, l["relu"], (l["pad"],) * 2, (l["stride"],) * 2)
            dt[nm] = odt
        elif kd == "pool":
            t[nm] = O.pool_i8_nhwc(t[l["src"]], (l["win"],) * 2, (l["stride"],) * 2, (l["pad"],) * 2, l["type"])
            dt[nm] = dt[l["src"]]
            scales[nm] = scales[l["src"]]
        elif kd == "eltwise":
            c = np.float32(1.0 / scales[nm])
            t[nm] = O.eltwise_i8(t[l["a"]], t[l["b"]], scales[l["a"]], scales[l["b"]], c, c, l["relu"])
            dt[nm] = S8
        elif kd == "gpool":
            deq = O.dequant_nhwc_to_nchw(t[l["src"]], scales[l["src"]])
            t[nm] = O.pool_f32_nchw(deq, None, None, None, 1, global_pool=True)
            dt[nm] = F32
        elif kd == "fc":
            w, b = model["params"][nm]
            ws, wq = prep[nm]
            xin = t[l["src"]].reshape(t[l["src"]].shape[0], -1)
            xq = O.quant_flat_s8(xin, scales[l["src"]])
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
            t[nm] = O.pool_f32_nchw(t[l["src"]], (l["win"],) * 2, (l["stride"],) * 2, (l["pad"],) * 2, l["type"])
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
