"""ctypes binding of the C-ABI HIP library (include/saber_hip.h -> anakin_amd/libsaber_mi355x.so).

There is NO fallback: if the library is missing or cannot be loaded this module raises, and every
entry point that computes needs a gfx950 device. PyTorch is used by callers only for device memory
and streams (tensor.data_ptr(), torch.cuda.current_stream().cuda_stream).
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SABER_MI355X_LIB") or os.path.join(HERE, "libsaber_mi355x.so")   # override: A/B builds

F32, S8, U8, S32 = 0, 1, 2, 3
NHWC, NCHW = 0, 1
ACT_NONE, ACT_RELU = 0, 1
POOL_MAX, POOL_AVG_INCL, POOL_AVG_EXCL = 0, 1, 2
RES_NONE, RES_SUM_INPLACE, RES_ELTWISE = 0, 1, 2
TILES = ["32x32", "64x32", "64x64", "128x64", "64x128", "128x128"]

# every symbol include/saber_hip.h declares (tests/test_abi.py checks the .so exports each one)
SYMBOLS = [
    "saber_hip_last_error", "saber_hip_device_ok",
    "saber_hip_conv2d_create", "saber_hip_conv2d_set_weights", "saber_hip_conv2d_workspace_bytes",
    "saber_hip_conv2d_out_shape", "saber_hip_conv2d_run", "saber_hip_conv2d_destroy",
    "saber_hip_conv2d_get_quantized_weights", "saber_hip_conv2d_algo", "saber_hip_conv2d_set_tile",
    "saber_hip_conv2d_get_tile", "saber_hip_conv2d_autotune", "saber_hip_conv2d_set_pooling",
    "saber_hip_conv2d_create_pair", "saber_hip_conv2d_run_pair", "saber_hip_conv2d_autotune_pair",
    "saber_hip_net_add_conv_pair",
    "saber_hip_conv2d_chain_create", "saber_hip_conv2d_chain_create3", "saber_hip_conv2d_chain_create3_pair", "saber_hip_conv2d_chain_destroy", "saber_hip_conv2d_chain_run",
    "saber_hip_conv2d_chain_run3",
    "saber_hip_conv2d_chain_set_tile", "saber_hip_conv2d_chain_get_tile",
    "saber_hip_conv2d_stage_create", "saber_hip_conv2d_stage_destroy", "saber_hip_conv2d_stage_run",
    "saber_hip_conv2d_stem_pair_create", "saber_hip_conv2d_stem_pair_destroy", "saber_hip_conv2d_stem_pair_run",
    "saber_hip_conv2d_set_global_pooling", "saber_hip_conv2d_run_gpool",
    "saber_hip_stage_create", "saber_hip_stage_num_tensors", "saber_hip_stage_run", "saber_hip_stage_status", "saber_hip_stage_trace", "saber_hip_stage_destroy",
    "saber_hip_fc_create", "saber_hip_fc_set_weights", "saber_hip_fc_workspace_bytes", "saber_hip_fc_run",
    "saber_hip_fc_destroy", "saber_hip_fc_algo", "saber_hip_fc_set_tile", "saber_hip_gemm_f32", "saber_hip_gemm_f32_release_plans",
    "saber_hip_gemm_i8_create", "saber_hip_gemm_i8_workspace_bytes", "saber_hip_gemm_i8_run", "saber_hip_gemm_i8_destroy",
    "saber_hip_quantize_nchw_to_nhwc", "saber_hip_dequantize_nhwc_to_nchw",
    "saber_hip_transpose_nchw_to_nhwc_f32", "saber_hip_transpose_nhwc_to_nchw_f32",
    "saber_hip_quantize_flat_s8", "saber_hip_eltwise_sum_i8", "saber_hip_eltwise_sum_f32", "saber_hip_relu_f32", "saber_hip_activation_f32", "saber_hip_prelu_f32",
    "saber_hip_pool_out_dim", "saber_hip_pool_out_dim2", "saber_hip_pool2d_i8_nhwc", "saber_hip_pool2d_f32", "saber_hip_pool2d_f32_from_i8", "saber_hip_pool2d_f32_from_i8_q", "saber_hip_fc_run_q", "saber_hip_fc_run_softmax", "saber_hip_softmax_f32",
    "saber_hip_net_add_pool_f32_from_i8_q", "saber_hip_net_add_fc_q", "saber_hip_net_optimize", "saber_hip_net_num_launches", "saber_hip_net_tensor_unwritten", "saber_hip_net_get_choice", "saber_hip_net_set_choice", "saber_hip_net_stage_blocks", "saber_hip_net_time_op_in_pass", "saber_hip_net_status", "saber_hip_net_inject_coop_error", "saber_hip_net_coop_fallbacks", "saber_hip_coop_fallbacks_total",
    "saber_hip_net_create", "saber_hip_net_add_tensor", "saber_hip_net_add_conv", "saber_hip_net_add_fc",
    "saber_hip_net_add_quantize", "saber_hip_net_add_dequantize", "saber_hip_net_add_transpose_in_f32", "saber_hip_net_add_eltwise_i8",
    "saber_hip_net_add_eltwise_f32", "saber_hip_net_add_pool_i8", "saber_hip_net_add_pool_f32",
    "saber_hip_net_add_pool_f32_from_i8", "saber_hip_net_add_softmax", "saber_hip_net_set_lane", "saber_hip_net_finalize", "saber_hip_net_tensor_ptr",
    "saber_hip_net_arena_bytes", "saber_hip_net_num_ops", "saber_hip_net_run", "saber_hip_net_run_op",
    "saber_hip_net_capture", "saber_hip_net_replay", "saber_hip_net_time_ops", "saber_hip_net_time_pass", "saber_hip_net_op_work", "saber_hip_net_op_name",
    "saber_hip_net_autotune", "saber_hip_net_destroy",
    "saber_hip_net_add_relu_f32", "saber_hip_net_add_activation_f32", "saber_hip_net_bind_tensor", "saber_hip_net_num_tensors", "saber_hip_net_tensor_bytes",
    "saber_hip_capture_begin", "saber_hip_capture_end", "saber_hip_capture_active", "saber_hip_net_tensor_of_ptr",
    "saber_hip_net_compact_arena", "saber_hip_net_arena_compacted", "saber_hip_serving_streams",
]


class StagePhase(C.Structure):      # saber_hip_stage_phase
    _fields_ = [("conv", C.c_void_p), ("in_", C.c_int), ("out", C.c_int), ("res", C.c_int)]


class ConvDesc(C.Structure):
    _fields_ = [(n, C.c_int) for n in
                ("n", "h", "w", "c", "k", "kh", "kw", "pad_h", "pad_w", "stride_h", "stride_w", "dil_h", "dil_w",
                 "group", "in_dtype", "out_dtype", "in_layout", "out_layout", "act", "res_mode", "res_act")] + \
               [("sum_scale", C.c_float), ("coeff_conv", C.c_float), ("coeff_res", C.c_float),
                ("scale_res", C.c_float), ("int8_weights", C.c_int), ("act_negative_slope", C.c_float),
                ("res_has_dtype", C.c_int), ("res_dtype", C.c_int), ("res_stride", C.c_int), ("res_h", C.c_int),
                ("res_w", C.c_int)]


class FcDesc(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("m", "n", "k", "in_dtype", "int8_weights", "w_is_kn")]


class SaberHipError(RuntimeError):
    pass


_lib = None


def load():
    """Load the HIP library; raises if it has not been built (python anakin_amd/build.py)."""
    global _lib
    if _lib is not None:
        return _lib
    try:
        # torch bundles its own libamdhip64 with the same SONAME as /opt/rocm's: whichever loads first
        # serves the whole process. Import torch first so one HIP runtime owns the device.
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise SaberHipError(
            "anakin_amd/libsaber_mi355x.so is missing - build it with `python anakin_amd/build.py` "
            "(there is no CPU fallback for the MI355X Saber target)")
    lib = C.CDLL(LIB_PATH)
    P, I, F, Z = C.c_void_p, C.c_int, C.c_float, C.c_size_t
    lib.saber_hip_last_error.restype = C.c_char_p
    lib.saber_hip_conv2d_create.argtypes = [C.POINTER(ConvDesc), C.POINTER(P)]
    lib.saber_hip_conv2d_set_weights.argtypes = [P, P, I, P, P, F, F]
    lib.saber_hip_conv2d_workspace_bytes.argtypes = [P]
    lib.saber_hip_conv2d_workspace_bytes.restype = Z
    lib.saber_hip_conv2d_out_shape.argtypes = [P, C.POINTER(I), C.POINTER(I)]
    lib.saber_hip_conv2d_out_shape.restype = None
    lib.saber_hip_conv2d_run.argtypes = [P, P, P, P, P, P]
    lib.saber_hip_conv2d_destroy.argtypes = [P]
    lib.saber_hip_conv2d_destroy.restype = None
    lib.saber_hip_conv2d_get_quantized_weights.argtypes = [P, P, P]
    lib.saber_hip_conv2d_algo.argtypes = [P]
    lib.saber_hip_conv2d_algo.restype = C.c_char_p
    lib.saber_hip_conv2d_set_tile.argtypes = [P, I]
    lib.saber_hip_conv2d_get_tile.argtypes = [P]
    lib.saber_hip_conv2d_set_pooling.argtypes = [P] + [I] * 8
    lib.saber_hip_conv2d_autotune.argtypes = [P, P, P, P, P, P, I]
    lib.saber_hip_conv2d_create_pair.argtypes = [P, P, C.POINTER(P)]
    lib.saber_hip_conv2d_run_pair.argtypes = [P, P, P, P, P]
    lib.saber_hip_conv2d_autotune_pair.argtypes = [P, P, P, P, P, I]
    lib.saber_hip_net_add_conv_pair.argtypes = [P, P, I, I, I]
    lib.saber_hip_conv2d_chain_create.argtypes = [P, P, C.POINTER(P)]
    lib.saber_hip_conv2d_chain_create3.argtypes = [P, P, P, C.POINTER(P)]
    lib.saber_hip_net_tensor_unwritten.argtypes = [P, I]
    lib.saber_hip_net_num_launches.argtypes = [P]
    lib.saber_hip_conv2d_chain_destroy.argtypes = [P]
    lib.saber_hip_conv2d_chain_destroy.restype = None
    lib.saber_hip_conv2d_chain_run.argtypes = [P, P, P, P, P, P]
    lib.saber_hip_conv2d_chain_create3_pair.argtypes = [P, P, P, P, C.POINTER(P)]
    lib.saber_hip_conv2d_chain_run3.argtypes = [P, P, P, P, P, P, P]
    lib.saber_hip_conv2d_chain_set_tile.argtypes = [P, I]
    lib.saber_hip_conv2d_chain_get_tile.argtypes = [P]
    lib.saber_hip_conv2d_stage_create.argtypes = [C.POINTER(P), I, C.POINTER(P)]
    lib.saber_hip_conv2d_stage_destroy.argtypes = [P]
    lib.saber_hip_conv2d_stage_destroy.restype = None
    lib.saber_hip_conv2d_stage_run.argtypes = [P, P, P, C.POINTER(P), C.POINTER(P), P]
    lib.saber_hip_conv2d_stem_pair_create.argtypes = [P, P, P, C.POINTER(P)]
    lib.saber_hip_conv2d_stem_pair_destroy.argtypes = [P]
    lib.saber_hip_conv2d_stem_pair_destroy.restype = None
    lib.saber_hip_conv2d_stem_pair_run.argtypes = [P, P, P, P, P, P, P]
    lib.saber_hip_conv2d_set_global_pooling.argtypes = [P]
    lib.saber_hip_conv2d_run_gpool.argtypes = [P, P, P, P, P, P]
    lib.saber_hip_stage_create.argtypes = [C.POINTER(StagePhase), I, C.POINTER(P)]
    lib.saber_hip_stage_num_tensors.argtypes = [P]
    lib.saber_hip_stage_run.argtypes = [P, C.POINTER(P), I, P]
    lib.saber_hip_stage_status.argtypes = [P]
    lib.saber_hip_stage_trace.argtypes = [P, P, C.c_size_t]
    lib.saber_hip_stage_destroy.argtypes = [P]
    lib.saber_hip_stage_destroy.restype = None
    lib.saber_hip_fc_create.argtypes = [C.POINTER(FcDesc), C.POINTER(P)]
    lib.saber_hip_fc_set_weights.argtypes = [P, P, I, P, P, F, F]
    lib.saber_hip_fc_workspace_bytes.argtypes = [P]
    lib.saber_hip_fc_workspace_bytes.restype = Z
    lib.saber_hip_fc_run.argtypes = [P, P, P, P, P]
    lib.saber_hip_fc_destroy.argtypes = [P]
    lib.saber_hip_net_optimize.argtypes = [P, I]
    lib.saber_hip_net_stage_blocks.argtypes = [P, I]
    lib.saber_hip_net_status.argtypes = [P]
    lib.saber_hip_net_inject_coop_error.argtypes = [P]
    lib.saber_hip_net_coop_fallbacks.argtypes = [P]
    lib.saber_hip_coop_fallbacks_total.argtypes = []
    lib.saber_hip_net_time_op_in_pass.argtypes = [P, P, I, I, C.POINTER(C.c_float)]
    lib.saber_hip_net_get_choice.argtypes = [P, I]
    lib.saber_hip_net_set_choice.argtypes = [P, I, I]
    lib.saber_hip_fc_algo.argtypes = [P]
    lib.saber_hip_fc_algo.restype = C.c_char_p
    lib.saber_hip_fc_set_tile.argtypes = [P, I]
    lib.saber_hip_fc_destroy.restype = None
    lib.saber_hip_gemm_f32.argtypes = [I, I, I, I, I, F, P, P, F, P, P]
    lib.saber_hip_gemm_f32_release_plans.argtypes = []
    lib.saber_hip_gemm_f32_release_plans.restype = I
    lib.saber_hip_gemm_i8_create.argtypes = [I, I, I, I, I, I, P, C.POINTER(P)]
    lib.saber_hip_gemm_i8_workspace_bytes.argtypes = [P]
    lib.saber_hip_gemm_i8_workspace_bytes.restype = Z
    lib.saber_hip_gemm_i8_run.argtypes = [P, P, P, P, P]
    lib.saber_hip_gemm_i8_destroy.argtypes = [P]
    lib.saber_hip_gemm_i8_destroy.restype = None
    lib.saber_hip_quantize_nchw_to_nhwc.argtypes = [I, I, I, I, I, I, F, P, P, P]
    lib.saber_hip_dequantize_nhwc_to_nchw.argtypes = [I, I, I, I, I, F, P, P, P]
    lib.saber_hip_transpose_nchw_to_nhwc_f32.argtypes = [I, I, I, I, I, P, P, P]
    lib.saber_hip_transpose_nhwc_to_nchw_f32.argtypes = [I, I, I, I, I, P, P, P]
    lib.saber_hip_quantize_flat_s8.argtypes = [Z, F, P, P, P]
    lib.saber_hip_eltwise_sum_i8.argtypes = [Z, P, P, F, F, F, F, I, P, P]
    lib.saber_hip_eltwise_sum_f32.argtypes = [Z, P, P, F, F, I, P, P]
    lib.saber_hip_pool_out_dim.argtypes = [I, I, I, I, I]
    lib.saber_hip_pool_out_dim2.argtypes = [I, I, I, I, I, I]
    lib.saber_hip_pool2d_i8_nhwc.argtypes = [I] * 15 + [P, P, P]
    lib.saber_hip_pool2d_f32.argtypes = [I] * 14 + [P, P, P]
    lib.saber_hip_pool2d_f32_from_i8.argtypes = [I] * 14 + [F, P, P, P]
    lib.saber_hip_pool2d_f32_from_i8_q.argtypes = [I] * 14 + [F, P, P, F, P, P]
    lib.saber_hip_fc_run_q.argtypes = [P, P, P, P]
    lib.saber_hip_net_add_pool_f32_from_i8_q.argtypes = [P] + [I] * 14 + [F, I, I, F, I]
    lib.saber_hip_net_add_fc_q.argtypes = [P, P, I, I]
    lib.saber_hip_softmax_f32.argtypes = [I, I, P, P, P]
    lib.saber_hip_relu_f32.argtypes = [Z, P, P, P]
    lib.saber_hip_activation_f32.argtypes = [I, Z, F, F, P, P, P]
    lib.saber_hip_prelu_f32.argtypes = [Z, I, I, I, P, P, P, P]
    lib.saber_hip_net_create.argtypes = [C.POINTER(P)]
    lib.saber_hip_net_add_tensor.argtypes = [P, Z]
    lib.saber_hip_net_add_conv.argtypes = [P, P, I, I, I]
    lib.saber_hip_net_add_fc.argtypes = [P, P, I, I]
    lib.saber_hip_net_add_quantize.argtypes = [P, I, I, I, I, I, I, F, I, I]
    lib.saber_hip_net_add_dequantize.argtypes = [P, I, I, I, I, I, F, I, I]
    lib.saber_hip_net_add_transpose_in_f32.argtypes = [P, I, I, I, I, I, I, I]
    lib.saber_hip_net_add_eltwise_i8.argtypes = [P, Z, F, F, F, F, I, I, I, I]
    lib.saber_hip_net_add_eltwise_f32.argtypes = [P, Z, F, F, I, I, I, I]
    lib.saber_hip_net_add_pool_i8.argtypes = [P] + [I] * 17
    lib.saber_hip_net_add_pool_f32.argtypes = [P] + [I] * 16
    lib.saber_hip_net_add_pool_f32_from_i8.argtypes = [P] + [I] * 14 + [F, I, I]
    lib.saber_hip_net_add_softmax.argtypes = [P, I, I, I, I]
    lib.saber_hip_net_set_lane.argtypes = [P, I, I]
    lib.saber_hip_net_finalize.argtypes = [P]
    lib.saber_hip_net_tensor_ptr.argtypes = [P, I]
    lib.saber_hip_net_tensor_ptr.restype = P
    lib.saber_hip_net_arena_bytes.argtypes = [P]
    lib.saber_hip_net_arena_bytes.restype = Z
    lib.saber_hip_net_compact_arena.argtypes = [P, P, I]
    lib.saber_hip_net_arena_compacted.argtypes = [P]
    lib.saber_hip_net_num_ops.argtypes = [P]
    lib.saber_hip_net_run.argtypes = [P, P]
    lib.saber_hip_net_run_op.argtypes = [P, I, P]
    lib.saber_hip_net_capture.argtypes = [P, P]
    lib.saber_hip_net_replay.argtypes = [P, P]
    lib.saber_hip_net_time_ops.argtypes = [P, P, I, P]
    lib.saber_hip_net_time_pass.argtypes = [P, P, I, P]
    lib.saber_hip_net_op_work.argtypes = [P, I, P, P]
    lib.saber_hip_net_op_name.argtypes = [P, I]
    lib.saber_hip_net_op_name.restype = C.c_char_p
    lib.saber_hip_net_autotune.argtypes = [P, P, I]
    lib.saber_hip_net_destroy.argtypes = [P]
    lib.saber_hip_net_destroy.restype = None
    lib.saber_hip_net_add_relu_f32.argtypes = [P, Z, I, I]
    lib.saber_hip_net_add_activation_f32.argtypes = [P, I, Z, F, F, I, I]
    lib.saber_hip_net_bind_tensor.argtypes = [P, I, P]
    lib.saber_hip_net_num_tensors.argtypes = [P]
    lib.saber_hip_net_tensor_bytes.argtypes = [P, I]
    lib.saber_hip_net_tensor_bytes.restype = Z
    lib.saber_hip_capture_begin.argtypes = []
    lib.saber_hip_capture_end.argtypes = [C.POINTER(P)]
    lib.saber_hip_capture_active.argtypes = []
    lib.saber_hip_net_tensor_of_ptr.argtypes = [P, P]
    lib.saber_hip_serving_streams.argtypes = [I, C.POINTER(P), C.POINTER(I)]
    _lib = lib
    return lib


UNIMPL = -3   # SABER_HIP_UNIMPL -> SaberUnImplError


def check(rc):
    if rc != 0:
        e = SaberHipError("saber_hip status %d: %s" % (rc, load().saber_hip_last_error().decode()))
        e.status = rc
        raise e


def check_count(rc):
    """entry points that return a count (>= 0) or a negative status"""
    if rc < 0:
        check(rc)
    return rc


def source_sha():
    """Hash of the kernel + host sources that determine what a forward pass launches (csrc/*, workloads.py): profiles
    record it, bench.py reports a profile's PMC traffic only while it still matches."""
    import glob
    import hashlib
    h = hashlib.sha256()
    here = os.path.dirname(os.path.abspath(__file__))
    for f in sorted(glob.glob(os.path.join(here, "csrc", "*"))) + [os.path.join(here, "workloads.py")]:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def require_device():
    if not load().saber_hip_device_ok():
        raise SaberHipError("no gfx950 (MI355X) device visible: the MI355X Saber target has no CPU fallback")
