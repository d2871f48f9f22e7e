"""Pin the CPU restatement (oracle/saber_oracle.c) to the reference's own x86 Saber objects
(oracle/_ref, compiled unmodified from /root/reference by oracle/Makefile). Bit-exact."""
import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.skipif(not O.ref_available(), reason="oracle/_ref not built")

CONV_CASES = [
    # N, H, W, C, K, k, pad, stride, dil, group
    (2, 14, 14, 64, 64, 3, 1, 1, 1, 1),
    (1, 14, 14, 64, 128, 1, 0, 1, 1, 1),
    (2, 15, 13, 32, 48, 1, 0, 2, 1, 1),
    (1, 17, 17, 16, 32, 3, 1, 2, 1, 1),
    (1, 12, 12, 16, 16, 3, 2, 1, 2, 1),
    (1, 20, 20, 3, 16, 7, 3, 2, 1, 1),
    (2, 9, 9, 32, 32, 3, 1, 1, 1, 4),
    (1, 7, 7, 20, 24, 5, 2, 1, 1, 1),
]


def _mk(case, in_dtype, seed):
    N, H, W, C, K, k, pad, stride, dil, group = case
    rng = np.random.default_rng(seed)
    if in_dtype == O.U8:
        x = rng.integers(0, 256, (N, H, W, C)).astype(np.uint8)
    else:
        x = rng.integers(-128, 128, (N, H, W, C)).astype(np.int8)
    w = (rng.standard_normal((K, C // group, k, k)) * 0.1).astype(np.float32)
    b = rng.standard_normal(K).astype(np.float32)
    return x, w, b


@pytest.mark.parametrize("case", CONV_CASES)
@pytest.mark.parametrize("in_dtype", [O.S8, O.U8])
@pytest.mark.parametrize("out_dtype", [O.S8, O.U8, O.F32])
@pytest.mark.parametrize("relu", [0, 1])
def test_conv_i8_matches_reference(case, in_dtype, out_dtype, relu):
    if out_dtype == O.U8 and not relu:
        pytest.skip("u8 output without relu is undefined in the reference GEMM path")
    # u8 -> s8 (every ResNet50 branch2c and res2a_branch1): the reference's dispatch() has no branch for it
    # (gemm_x8s8s32x_conv.cpp:290-308), so oracle/_ref calls the public member template
    # GemmX8S8S32XConv::sub_dispatch<uint8_t,int8_t> (gemm_x8s8s32x_conv.h:72) after the reference's own init()/create()
    # computed the scale (:163-166) - see oracle/ref_gemm_conv_u8s8.cpp
    N, H, W, C, K, k, pad, stride, dil, group = case
    x, w, b = _mk(case, in_dtype, seed=hash((case, in_dtype)) % 2**31)
    in_scale = 0.02
    ws = O.weight_scales(w)
    wq = O.quant_weights(w, ws)
    # in-range output scale: calibrate from the f32-output run of the oracle itself
    bp, sc = O.conv_i8_prepare(ws, b, in_scale, 1.0, in_dtype, O.F32)
    real = O.conv_i8(x, wq, bp, sc, O.F32, relu, (pad, pad), (stride, stride), (dil, dil), group)
    out_scale = float(np.abs(real).max()) / (120.0 if out_dtype == O.S8 else 240.0 * 127 / 255)
    bp, sc = O.conv_i8_prepare(ws, b, in_scale, out_scale, in_dtype, out_dtype)
    got = O.conv_i8(x, wq, bp, sc, out_dtype, relu, (pad, pad), (stride, stride), (dil, dil), group)
    # reference, fed the f32 weights (it quantises them itself) ...
    want = O.ref_conv_i8(x, w, None, b, in_scale, out_scale, out_dtype, relu, (pad, pad),
                         (stride, stride), (dil, dil), group)
    assert np.array_equal(got, want)
    # ... and fed pre-quantised s8 weights + scales
    want2 = O.ref_conv_i8(x, wq, ws, b, in_scale, out_scale, out_dtype, relu, (pad, pad),
                          (stride, stride), (dil, dil), group)
    assert np.array_equal(got, want2)


def test_weight_quant_matches_reference():
    rng = np.random.default_rng(3)
    w = (rng.standard_normal((64, 32, 3, 3)) * 0.2).astype(np.float32)
    q_ref, s_ref = O.ref_quant_conv_weights(w)
    s = O.weight_scales(w)
    assert np.array_equal(s, s_ref)
    assert np.array_equal(O.quant_weights(w, s), q_ref)


@pytest.mark.parametrize("dt", [O.S8, O.U8])
def test_quant_dequant_reorder_matches_reference(dt):
    rng = np.random.default_rng(4)
    x = (rng.standard_normal((2, 5, 7, 9)) * 3).astype(np.float32)
    x[0, 0, 0, :4] = [0.5 * 0.05, 1.5 * 0.05, -0.5 * 0.05, 1e6]  # ties + saturation
    q = O.quant_nchw_to_nhwc(x, 0.05, dt)
    assert np.array_equal(q, O.ref_reorder(x, 0, dt, 0.05))
    f = O.dequant_nhwc_to_nchw(q, 0.05)
    assert np.array_equal(f, O.ref_reorder(q, 1, O.F32, 0.05))


@pytest.mark.parametrize("relu", [0, 1])
def test_eltwise_i8_matches_reference(relu):
    rng = np.random.default_rng(5)
    a = rng.integers(-128, 128, (2, 7, 7, 32)).astype(np.int8)
    b = rng.integers(-128, 128, (2, 7, 7, 32)).astype(np.int8)
    for (sa, sb, c0, c1) in [(0.031, 0.047, 1.0, 1.0), (0.5, 0.25, 1.0, 1.0), (0.02, 0.03, 20.0, 20.0)]:
        got = O.eltwise_i8(a, b, sa, sb, c0, c1, relu)
        assert np.array_equal(got, O.ref_eltwise_i8(a, b, sa, sb, c0, c1, relu))


def test_eltwise_f32_matches_reference():
    rng = np.random.default_rng(6)
    a = rng.standard_normal((2, 8, 5, 5)).astype(np.float32)
    b = rng.standard_normal((2, 8, 5, 5)).astype(np.float32)
    assert np.array_equal(O.eltwise_f32(a, b, 1.0, 1.0, True), O.ref_eltwise_f32(a, b, 1.0, 1.0, True))
    assert np.array_equal(O.eltwise_f32(a, b, 0.5, 2.0, False), O.ref_eltwise_f32(a, b, 0.5, 2.0, False))


@pytest.mark.parametrize("case", CONV_CASES[:6])
def test_conv_f32_matches_reference_naive(case):
    N, H, W, C, K, k, pad, stride, dil, group = case
    rng = np.random.default_rng(7)
    x = rng.standard_normal((N, C, H, W)).astype(np.float32)
    w = (rng.standard_normal((K, C // group, k, k)) * 0.1).astype(np.float32)
    b = rng.standard_normal(K).astype(np.float32)
    got = O.conv_f32_nchw(x, w, b, True, (pad, pad), (stride, stride), (dil, dil), group)
    want = O.ref_conv_basic_check_f32(x, w, b, True, (pad, pad), (stride, stride), (dil, dil), group)
    assert np.array_equal(got, want)


def test_conv1x1_f32_production_path_within_tolerance():
    """SaberConv1X1 (MKL sgemm) vs the naive order: FP32 parity is tolerance-only (1e-4 rel)."""
    rng = np.random.default_rng(8)
    x = rng.standard_normal((2, 64, 14, 14)).astype(np.float32)
    w = (rng.standard_normal((128, 64, 1, 1)) * 0.1).astype(np.float32)
    b = rng.standard_normal(128).astype(np.float32)
    res = rng.standard_normal((2, 128, 14, 14)).astype(np.float32)
    got = O.conv_f32_nchw(x, w, b, True)
    want = O.ref_conv1x1_f32(x, w, b, True)
    assert np.abs(got - want).max() <= 1e-4 * np.abs(want).max()
    got = O.conv_f32_nchw(x, w, b, True, beta=1.0, out_init=res)
    want = O.ref_conv1x1_f32(x, w, b, True, residual=res)
    assert np.abs(got - want).max() <= 1e-4 * np.abs(want).max()


def test_conv_i8_matches_reference_integer_test_oracle():
    """conv_basic_check_int8 (the only integer oracle in the reference's tests) accumulates in float
    and has no pre-scaled float bias; with bias=None and |acc| < 2^24 it must agree exactly."""
    rng = np.random.default_rng(9)
    x = rng.integers(-128, 128, (1, 9, 9, 32)).astype(np.int8)
    wq = rng.integers(-127, 128, (16, 32, 3, 3)).astype(np.int8)
    scale = (rng.random(16) * 1e-3 + 1e-4).astype(np.float32)
    got = O.conv_i8(x, wq, None, scale, O.S8, True, (1, 1))
    want = O.ref_conv_basic_check_int8(x, wq, None, scale, True, (1, 1))
    assert np.array_equal(got, want)


def test_pool_matches_reference_test_oracle():
    rng = np.random.default_rng(10)
    # 13x13: every 3x3/s2 window is fully in bounds (the helper does not clip when pad == 0)
    x = rng.integers(0, 128, (2, 13, 13, 16)).astype(np.int8)
    oh = O.pool_out_dim(13, 0, 3, 2)
    got = O.pool_i8_nhwc(x, (3, 3), (2, 2), (0, 0), 0)
    want = O.ref_pool_basic_check_int8(x, oh, oh, (3, 3), (2, 2), (0, 0), 0)
    assert np.array_equal(got, want)
    got = O.pool_i8_nhwc(x[:, :12, :12], (2, 2), (2, 2), (0, 0), 1)
    want = O.ref_pool_basic_check_int8(x[:, :12, :12], 6, 6, (2, 2), (2, 2), (0, 0), 1)
    assert np.array_equal(got, want)  # /4 is exact both ways
    got = O.pool_i8_nhwc(x, (3, 3), (2, 2), (0, 0), 1).astype(np.int32)
    want = O.ref_pool_basic_check_int8(x, oh, oh, (3, 3), (2, 2), (0, 0), 1).astype(np.int32)
    assert np.abs(got - want).max() <= 1 and (got != want).mean() < 0.01


@pytest.mark.parametrize("shape", [(8, 1000, 2048), (3, 17, 50), (1, 64, 512)])
@pytest.mark.parametrize("with_bias", [False, True])
def test_fc_i8_matches_reference_packed_gemm(shape, with_bias):
    """INT8 fc, f32 input: the restatement (flat quantise -> per-row truncating weight quantisation -> exact s32 ->
    (float)acc * (w_scale*in_scale) + bias) equals VenderFc<X86,AK_INT8>'s PackedMKLInt8Gemm bit for bit."""
    M, N, K = shape
    rng = np.random.default_rng(M * 1000 + N + K)
    x = rng.standard_normal((M, K)).astype(np.float32)
    x[0, :4] = [0.5 * 0.031, -0.5 * 0.031, 1.5 * 0.031, 200.0]      # ties (round half away) and saturation
    w = (rng.standard_normal((N, K)) * 0.05).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32) if with_bias else None
    in_scale = 0.031
    want = O.ref_fc_i8_packed(x, w, b, in_scale)
    ws = O.weight_scales(w.reshape(N, K, 1, 1))
    wq = O.quant_weights(w.reshape(N, K, 1, 1), ws).reshape(N, K)
    got = O.fc_i8(O.quant_flat_s8(x, in_scale), wq, ws, in_scale, b)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("ta,tb", [(0, 0), (0, 1)])
def test_gemm_s8s8s32_matches_reference(ta, tb):
    """INT8 GEMM: the integer product the HIP path is tested against equals MklDnnGemm<int8_t,int8_t,int>."""
    rng = np.random.default_rng(9 + ta * 2 + tb)
    M, N, K = 33, 70, 130
    A = rng.integers(-128, 128, (M, K)).astype(np.int8)
    B = rng.integers(-128, 128, (K, N)).astype(np.int8)
    want = (A.astype(np.int64) @ B.astype(np.int64)).astype(np.int32)
    got = O.ref_gemm_s8s8s32(np.ascontiguousarray(A.T) if ta else A, np.ascontiguousarray(B.T) if tb else B, M, N, K, ta, tb)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("has_bias", [False, True])
@pytest.mark.parametrize("scale_bias", [False, True])
@pytest.mark.parametrize("bn_scale", [0.0, 1.0, 0.999])
def test_bn_fold_matches_reference(has_bias, scale_bias, bn_scale):
    """BatchNorm + Scale folding: the restatement (and the numpy version the workloads use) == the compiled
    WeightsFusion<float,X86>::update_weights, bit for bit."""
    from anakin_amd import workloads as W
    rng = np.random.default_rng(77 + int(has_bias) + 2 * int(scale_bias))
    K, C, k = 24, 16, 3
    w = rng.standard_normal((K, C, k, k)).astype(np.float32)
    bias = rng.standard_normal(K).astype(np.float32) if has_bias else None
    mean = rng.uniform(-0.5, 0.5, K).astype(np.float32)
    var = rng.uniform(0.2, 2.0, K).astype(np.float32)
    gamma = rng.uniform(0.5, 1.5, K).astype(np.float32)
    beta = rng.uniform(-0.3, 0.3, K).astype(np.float32) if scale_bias else None
    want_w, want_b = O.ref_bn_fold(w, bias, bn_scale, 1e-5, mean, var, gamma, beta)
    got_w, got_b = O.bn_fold(w, bias, bn_scale, 1e-5, mean, var, gamma, beta)
    assert np.array_equal(got_w, want_w) and np.array_equal(got_b, want_b)
    np_w, np_b = W.fold_bn(w, bias, bn_scale, 1e-5, mean, var, gamma, beta)
    assert np.array_equal(np_w, want_w) and np.array_equal(np_b, want_b)


@pytest.mark.parametrize("in_kind", ["f32", "u8"])
@pytest.mark.parametrize("with_bias", [False, True])
def test_vender_fc_i8_operator_matches_reference(in_kind, with_bias):
    """The whole VenderFc<X86,AK_INT8> operator (init + dispatch) with an f32 output: f32 input (quantised on entry)
    through PackedMKLInt8Gemm; u8 input through the cblas_gemm_s8u8s32 path with its truncated integer bias and
    scale = in_scale*w_scale/out_scale. Restatement == compiled reference, bit for bit. (An s8 input with an f32
    output is not a reference combination: PackedMKLInt8Gemm::dispatch only offers s8 -> s32 for it, which is the
    INT8 GEMM pinned above.)"""
    M, N, K = 5, 40, 96
    rng = np.random.default_rng(hash(in_kind) % 1000 + int(with_bias))
    w = (rng.standard_normal((N, K)) * 0.05).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32) if with_bias else None
    in_scale, out_scale = 0.031, 0.5
    ws = O.weight_scales(w.reshape(N, K, 1, 1))
    wq = O.quant_weights(w.reshape(N, K, 1, 1), ws).reshape(N, K)
    if in_kind == "f32":
        x = rng.standard_normal((M, K)).astype(np.float32)
        got = O.fc_i8(O.quant_flat_s8(x, in_scale), wq, ws, in_scale, b)
    else:
        x = rng.integers(0, 256, (M, K)).astype(np.uint8)
        got = O.fc_i8(x, wq, ws, in_scale, b, out_scale)
    want = O.ref_vender_fc_i8(x, w, b, in_scale, out_scale)
    assert np.array_equal(got, want)


def test_resnet50_int8_op_list_through_reference_objects():
    """The whole unfused ResNet50 INT8 op list (batch 1, 224x224) through the reference's own compiled objects
    (oracle/net_oracle.RefNet: GemmX8S8S32XConv incl. sub_dispatch<uint8_t,int8_t> for the 17 u8 -> s8 convs,
    SaberEltwise<X86,AK_INT8>, PackedMKLInt8Gemm) equals the restated oracle op list on EVERY edge and on the logits,
    bit for bit. Scales carry 25 % head-room so that no requantisation saturates: the GEMM path casts without
    saturating (gemm_x8s8s32x_conv.cpp:278, undefined on overflow) where the pinned contract saturates (JIT semantics)."""
    from anakin_amd import workloads as W
    from oracle import net_oracle as NO
    model = W.build_model("resnet50")
    xs = W.make_input(1)
    scales = {k: v * 1.25 for k, v in W.calibrate(model, W.make_input(2)).items()}
    rn = NO.RefNet(model, scales, 1)
    y = rn.run(xs)
    ref = NO.run_int8(model, dict(scales), xs)
    assert np.array_equal(y, ref["fc1000"].reshape(y.shape))
    edges = [nm for nm in rn.shape if nm != "data" and nm in ref]
    assert len(edges) >= 70
    for nm in edges:
        assert np.array_equal(rn.read(nm), ref[nm]), nm


def test_resnet50_int8_FRAMEWORK_op_list_through_reference_objects():
    """The same pin for the list the reference's own optimiser + edge rules emit (workloads.framework_spec, proved equal to
    Graph::Optimize + Net::init by tests/test_net_oplist.py): three stride-2 3x3 convolutions with 1x1 / stride-2 max
    poolings on their shortcuts (apply_stride_up), conv1 writing s8 (its consumer's dtype), INT8 global average pooling and
    the fc on its s8 result — 76 operators through GemmX8S8S32XConv (incl. <int8_t,int8_t> with relu for conv1),
    SaberEltwise<X86,AK_INT8> and PackedMKLInt8Gemm (s8 input): every edge and the logits equal the restated oracle's."""
    from anakin_amd import workloads as W
    from oracle import net_oracle as NO
    model = W.framework_model(W.build_model("resnet50"), "int8")
    assert len(model["spec"]) == 76
    xs = W.make_input(1)
    scales = {k: v * 1.25 for k, v in W.calibrate(model, W.make_input(2)).items()}
    rn = NO.RefNet(model, scales, 1)
    y = rn.run(xs)
    ref = NO.run_int8(model, dict(scales), xs)
    assert np.array_equal(y, ref["fc1000"].reshape(y.shape))
    edges = [nm for nm in rn.shape if nm != "data" and nm in ref]
    assert len(edges) >= 74
    for nm in edges:
        assert np.array_equal(rn.read(nm).reshape(ref[nm].shape), ref[nm]), nm
    assert ref["conv1"].dtype == np.int8 and ref["conv1"].min() >= 0          # relu'd, stored as s8
    assert ref["res2c"].shape == (1, 28, 28, 256)                              # stride-up: the block already runs at 28 x 28


@pytest.mark.parametrize("sum_scale", [1.0, 0.5, 2.0, 0.37])
def test_conv_i8_with_sum_matches_reference_integer_test_oracle(sum_scale):
    """The INT8 conv + in-place sum post-op (the JIT `with_sum`, kernel/jit_avx512_core_x8s8s32x_conv_kernel.cpp:156-177 —
    xbyak, not buildable here) pinned with reference CODE that is: conv_basic_check_int8's Eltwise_sum branch
    (test/saber/conv_func_helper.h:127-130,168-186), compiled into oracle/_ref. Same structure — acc * scale, + prev *
    sum_scale, relu, round to nearest even, saturate — with |acc| < 2^24 so that the helper's float accumulation is exact.
    For sum_scale in {1, 1/2, 2} (prev * sum_scale exact) the two must agree BIT FOR BIT. For a general factor the JIT fuses
    the multiply-add (vfmadd231ps: one rounding, restated with fmaf) where the helper rounds twice: the results may differ
    by one unit where the sum lands within an ulp of a rounding boundary — at most 1, on under 0.1 % of the outputs."""
    rng = np.random.default_rng(21)
    x = rng.integers(-128, 128, (2, 9, 9, 32)).astype(np.int8)
    wq = rng.integers(-127, 128, (16, 32, 3, 3)).astype(np.int8)
    scale = (rng.random(16) * 1e-3 + 1e-4).astype(np.float32)
    prev = rng.integers(-128, 128, (2, 9, 9, 16)).astype(np.int8)
    rp = O.Residual(O.RES_JIT_SUM, 0, sum_scale, O.S8, 0, 0, 0, 0)
    got = O.conv_i8(x, wq, None, scale, O.S8, True, (1, 1), residual=rp, out_init=prev.copy())
    want = O.ref_conv_basic_check_int8_sum(x, wq, scale, True, prev, sum_scale, (1, 1))
    if sum_scale in (1.0, 0.5, 2.0):
        assert np.array_equal(got, want)
    else:
        d = np.abs(got.astype(np.int32) - want.astype(np.int32))
        assert d.max() <= 1 and (d != 0).mean() < 1e-3, (d.max(), (d != 0).mean())
    assert (got > 0).any() and (got == 0).any()      # the relu and the sum both matter in this sample


def test_int8_average_pooling_contract_is_the_jit_kernels():
    """INT8 average pooling: the x86 implementation is an xbyak JIT kernel (kernel/jit_avx512_core_8bit_pooling_kernel.cpp,
    not buildable here) that multiplies the int32 window sum by a precomputed 1/count and rounds to nearest even; the
    reference's test helper pool_basic_check_int8 (conv_func_helper.h:29-100, compiled into oracle/_ref) DIVIDES. The pinned
    contract is the kernel's (`(float)sum * (1.f / count)`): it equals the helper wherever count is a power of two (1/count
    exact) and differs by at most one unit elsewhere, only where sum / count sits within an ulp of a .5 boundary. Both facts
    are tested; the deviation is a documented choice, not an unknown."""
    rng = np.random.default_rng(12)
    x = rng.integers(0, 128, (2, 16, 16, 32)).astype(np.int8)
    for win, stride in (((2, 2), (2, 2)), ((4, 4), (4, 4)), ((8, 8), (8, 8))):      # count 4 / 16 / 64: exact reciprocals
        oh = O.pool_out_dim(16, 0, win[0], stride[0])
        assert np.array_equal(O.pool_i8_nhwc(x, win, stride, (0, 0), 1), O.ref_pool_basic_check_int8(x, oh, oh, win, stride, (0, 0), 1))
    x7 = rng.integers(0, 128, (4, 7, 7, 256)).astype(np.int8)                          # count 49 (ResNet's pool5)
    got = O.pool_i8_nhwc(x7, (7, 7), (7, 7), (0, 0), 1).astype(np.int32)
    want = O.ref_pool_basic_check_int8(x7, 1, 1, (7, 7), (7, 7), (0, 0), 1).astype(np.int32)
    assert np.abs(got - want).max() <= 1 and (got != want).mean() < 0.02
    # and the kernel's formula, restated independently in numpy float32
    s = x7.astype(np.int32).sum((1, 2), keepdims=True)
    f = (s.astype(np.float32) * np.float32(1.0 / 49.0)).astype(np.float32)
    assert np.array_equal(got, np.clip(np.rint(f), -128, 127).astype(np.int32))


# ---------------------------------------------------------------------------------------------------------------------
# Round 6: FP32 convolutions pinned to the reference's PRODUCTION x86 implementations (the ones SaberConv2D<X86,AK_FLOAT>::init
# chooses between, saber/funcs/impl/x86/saber_conv.cpp:49-136), not only to its test helper conv_basic_check.
def _f32_criteria(got, want):
    """the two FP32 criteria of the GPU parity tests: max-norm and element-wise (|ref| + mean|ref| in the denominator)"""
    d = np.abs(got.astype(np.float64) - want.astype(np.float64))
    return d.max() / np.abs(want).max(), (d / (np.abs(want) + np.abs(want).mean())).max()


F32_PROD_CASES = [
    # N, C, H, K, k, pad, stride, dil   - what each is in ResNet50 / VGG16
    (1, 3, 40, 64, 7, 3, 2, 1),        # conv1 7x7 / stride 2 (im2col)
    (2, 64, 14, 64, 3, 1, 1, 1),       # 3x3 / stride 1 on a map >= 12 (Winograd by the rule)
    (1, 128, 7, 128, 3, 1, 1, 1),      # 3x3 at 7x7: res5 (im2col: the map is below Winograd's 12)
    (1, 48, 17, 32, 3, 1, 2, 1),       # 3x3 / stride 2: the stride-up form of res3a/4a/5a branch2b (im2col)
    (2, 64, 15, 96, 1, 0, 2, 1),       # 1x1 / stride 2: Caffe-topology branch1 / branch2a (im2col)
    (1, 16, 13, 24, 3, 2, 1, 2),       # dilated 3x3 (im2col)
    (1, 256, 14, 256, 3, 1, 1, 1),     # res4 branch2b at full width
]


@pytest.mark.parametrize("case", F32_PROD_CASES)
def test_conv_f32_oracle_vs_reference_production_im2col(case):
    """oracle.conv_f32_nchw (naive order, == conv_basic_check bit for bit above) against SaberIm2colConv<AK_FLOAT>
    (saber_im2col_conv.cpp:93-219: im2col_cpu_par + Gemm<X86,VENDER_IMPL,float> = MKL cblas_sgemm, bias / relu loops): 1e-4 on both
    criteria (north_star's FP32 tolerance; the summation order is MKL's, so FP32 parity is tolerance-only)."""
    N, C, H, K, k, pad, stride, dil = case
    rng = np.random.default_rng(1000 + C + k)
    x = rng.standard_normal((N, C, H, H)).astype(np.float32)
    w = (rng.standard_normal((K, C, k, k)) * np.sqrt(2.0 / (C * k * k))).astype(np.float32)
    b = (rng.standard_normal(K) * 0.1).astype(np.float32)
    for relu, bias in ((True, b), (False, b), (True, None)):
        want, used = O.ref_conv_f32(x, w, bias, relu, (pad, pad), (stride, stride), (dil, dil), impl=O.REF_F32_IM2COL)
        assert used == O.REF_F32_IM2COL
        got = O.conv_f32_nchw(x, w, bias, relu, (pad, pad), (stride, stride), (dil, dil))
        a, e = _f32_criteria(got, want)
        assert a <= 1e-4 and e <= 1e-4, (case, relu, a, e)


def test_conv_f32_reference_dispatcher_rule_and_winograd():
    """ref_f32_conv_rule restates which implementation saber_conv.cpp:92-136 ends with (JIT kernels absent); the reference's
    Winograd F(6x6, 3x3) kernels (winograd_avx2.cpp) against the naive order: 1e-3, the tolerance of the reference's own conv test
    (test/saber/test_saber_conv.cpp:189,201) - they are NOT a 1e-4 path (measured 5e-5 .. 1.4e-4 element-wise), which is why the
    GPU target's FP32 parity is pinned to the naive / im2col order and Winograd is out of scope."""
    assert O.ref_f32_conv_rule(64, 56, 56, 64, 3, 1, 1) == O.REF_F32_WINOGRAD
    assert O.ref_f32_conv_rule(512, 7, 7, 512, 3, 1, 1) == O.REF_F32_IM2COL        # map below 12
    assert O.ref_f32_conv_rule(3, 224, 224, 64, 7, 3, 2) == O.REF_F32_IM2COL
    assert O.ref_f32_conv_rule(256, 56, 56, 64, 1, 0, 1) == O.REF_F32_CONV1X1
    assert O.ref_f32_conv_rule(256, 56, 56, 512, 1, 0, 2) == O.REF_F32_IM2COL      # strided 1x1
    assert O.ref_f32_conv_rule(128, 28, 28, 128, 3, 1, 2) == O.REF_F32_IM2COL      # strided 3x3 (stride-up form)
    rng = np.random.default_rng(77)
    for (N, C, H, K) in ((2, 64, 14, 64), (1, 64, 28, 32), (1, 16, 12, 16)):
        x = rng.standard_normal((N, C, H, H)).astype(np.float32)
        w = (rng.standard_normal((K, C, 3, 3)) * np.sqrt(2.0 / (C * 9))).astype(np.float32)
        b = (rng.standard_normal(K) * 0.1).astype(np.float32)
        want, used = O.ref_conv_f32(x, w, b, True, (1, 1), (1, 1))
        assert used == O.REF_F32_WINOGRAD
        a, e = _f32_criteria(O.conv_f32_nchw(x, w, b, True, (1, 1), (1, 1)), want)
        assert a <= 1e-3 and e <= 1e-3, (a, e)


def test_fc_f32_oracle_vs_reference_gemm():
    rng = np.random.default_rng(5)
    x = rng.standard_normal((4, 2048)).astype(np.float32)
    w = (rng.standard_normal((1000, 2048)) * 0.02).astype(np.float32)
    b = rng.standard_normal(1000).astype(np.float32)
    a, e = _f32_criteria(O.fc_f32(x, w, b), O.ref_fc_f32(x, w, b))
    assert a <= 1e-4 and e <= 1e-4


@pytest.mark.parametrize("impl", [0, 1])
def test_resnet50_fp32_op_list_through_reference_production_objects(impl):
    """BASELINE.json configs[0] in shape: the ResNet50 FP32 op list through the reference's own x86 objects
    (net_oracle.RefNetF32: Winograd / SaberConv1X1 (+ beta = 1 residual, the ConvEltwise operator) / SaberIm2colConv as
    saber_conv.cpp's rule selects them, or im2col forced) against the restated oracle's naive-order pass (run_fp32), at 96 x 96 so that
    the CPU suite stays short: every edge and the logits within 1e-4 on both criteria with im2col forced, and within the reference's
    own 1e-3 with its Winograd kernels in the list."""
    from anakin_amd import workloads as W
    from oracle import net_oracle as NO
    model = W.build_model("resnet50")
    x = W.make_input(1, hw=96)
    rn = NO.RefNetF32(model, 1, hw=96, impl=impl)
    counts = rn.impl_counts()
    if impl == 0:      # 96 x 96: res2 (24 x 24) and res3 (12 x 12) 3x3 layers are Winograd's, res4 / res5 are below its 12 x 12
        assert counts == {"SaberConvWinograd": 7, "SaberConv1X1": 30, "SaberIm2colConv": 16}, counts
    else:
        assert counts == {"SaberConv1X1": 16, "SaberIm2colConv": 37}, counts
    y = rn.run(x)
    ref = NO.run_fp32(model, x)
    tol = 1e-4 if impl == 1 else 1e-3
    a, e = _f32_criteria(y, ref["fc1000"].reshape(y.shape))
    assert a <= tol and e <= tol, (a, e)
    # the last edge of every stage (the in-place ConvEltwise results are overwritten block by block; the final ones survive)
    for nm in ("pool1", "res5c", "pool5"):
        a, e = _f32_criteria(rn.read(nm), ref[nm].reshape(rn.read(nm).shape))
        assert a <= tol and e <= tol, (nm, a, e)
