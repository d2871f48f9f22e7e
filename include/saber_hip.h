/* include/saber_hip.h — the C-ABI drop-in boundary of the MI355X (gfx950) Saber target.
 *
 * This is what a `SaberConv2D<MI355X, ...>` / `SaberConvEltwise<MI355X, ...>` / `SaberFc<MI355X, ...>` /
 * `Gemm<MI355X, SABER_IMPL, ...>` specialisation under saber/funcs/impl/ would bind to
 * (INTEGRATION.md shows the adaptor a maintainer adds). The reference has no C ABI today; the
 * interface each group of functions replaces is its C++ template API:
 *
 *   ImplBase<TargetType, OpDtype, Param>::init / create / dispatch
 *                                              saber/funcs/impl/impl_base.h:33-69
 *   BaseFunc<...>::init / operator()           saber/funcs/base.h:85-162
 *   Conv<T,D> / ConvEltwise<T,D>               saber/funcs/conv.h:56-131, conv_eltwise.h:44-110
 *   ConvParam / ConvEltwiseParam / ActivationParam / EltwiseParam
 *                                              saber/saber_funcs_param.h:470-581,586-615,48-110,1077-1140
 *   Fc<T,D>, FcParam                           saber/funcs/fc.h:48-127, saber_funcs_param.h:1236-1279
 *   Gemm<T,impl,in,out>::init / dispatch       saber/funcs/gemm.h:27-66
 *   Pooling<T,D>, Eltwise<T,D>, Softmax<T,D>   saber/funcs/pooling.h:69-130, eltwise.h, softmax.h
 *   reorder_nhwc_nchw (quantise / dequantise)  saber/funcs/saber_util.h:637-803
 *   Net<T,P,R>::prediction (the caller)        framework/core/net/net.cpp:417-509
 *
 * Call protocol (mirrors init-once / dispatch-many):
 *   *_create           once per operator ("init"/"create": shapes, algorithm choice)
 *   *_set_weights      once ("init": quantise + repack weights, pre-scale bias — HOST pointers, cold path)
 *   *_run              every inference ("dispatch"): enqueues on the given hipStream_t, never syncs.
 * All tensor pointers given to *_run are DEVICE pointers owned by the caller. Returns 0 on success
 * or a negative saber_hip_status; the adaptor maps 0 -> SaberSuccess(-1, saber_types.h:224) and
 * the others onto SaberInvalidValue / SaberUnImplError / SaberOutOfMem.
 *
 * Plain C: no C++ types, no torch types. `hipStream_t` is passed as void*.
 */
#ifndef SABER_HIP_H
#define SABER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* saber_hip_stream_t; /* a hipStream_t */

typedef enum {
    SABER_HIP_OK = 0,
    SABER_HIP_INVALID_VALUE = -2, /* -> SaberInvalidValue */
    SABER_HIP_UNIMPL = -3,        /* -> SaberUnImplError  */
    SABER_HIP_OUT_OF_MEM = -4,    /* -> SaberOutOfMem     */
    SABER_HIP_RUNTIME_ERROR = -5  /* a HIP call failed; see saber_hip_last_error() */
} saber_hip_status;

/* DataType (saber_types.h:205-222) subset on this path */
typedef enum { SABER_HIP_F32 = 0, SABER_HIP_S8 = 1, SABER_HIP_U8 = 2, SABER_HIP_S32 = 3 } saber_hip_dtype;
/* LayoutType (saber_types.h:69-87) subset */
typedef enum { SABER_HIP_NHWC = 0, SABER_HIP_NCHW = 1 } saber_hip_layout;
/* ActiveType: only what the x86 path implements on this route */
typedef enum { SABER_HIP_ACT_NONE = 0, SABER_HIP_ACT_RELU = 1 } saber_hip_act;
/* PoolingType (saber_types.h) */
typedef enum { SABER_HIP_POOL_MAX = 0, SABER_HIP_POOL_AVG_INCL = 1, SABER_HIP_POOL_AVG_EXCL = 2 } saber_hip_pool_type;

/* Fused residual modes of the convolution epilogue (ConvEltwiseParam) */
typedef enum {
    SABER_HIP_RES_NONE = 0,
    /* INT8: the x86 JIT `with_sum` post-op (jit_avx512_core_x8s8s32x_conv_kernel.cpp:156-177):
     *   d = sum_scale==1 ? d + prev : fmaf(prev, sum_scale, d); relu.  prev is read from y (in place).
     * FP32: out = act(conv + bias + 1*y)  (SaberConvEltwise<X86,AK_FLOAT>, saber_conv_eltwise.cpp:40-151). */
    SABER_HIP_RES_SUM_INPLACE = 1,
    /* INT8 only: bit-exact fusion of  conv(->s8)  +  SaberEltwise<X86,AK_INT8> sum(+relu)  (the two
     * ops of the unfused INT8 graph, graph.cpp:423-436; saber_eltwise.cpp:98-111). res is a separate s8
     * NHWC tensor; y is s8. */
    SABER_HIP_RES_ELTWISE = 2
} saber_hip_res_mode;

const char* saber_hip_last_error(void);
/* 1 when a gfx950 device is visible to this process. */
int saber_hip_device_ok(void);

/* ------------------------------------------------------------------------------------------- */
/* Convolution (SaberConv2D / SaberConvEltwise), FP32 and INT8                                  */
/* ------------------------------------------------------------------------------------------- */
typedef struct {
    int n, h, w, c;              /* input  [n,c,h,w] logical */
    int k;                       /* out channels */
    int kh, kw;
    int pad_h, pad_w, stride_h, stride_w, dil_h, dil_w, group;   /* ConvParam:470-581 */
    int in_dtype, out_dtype;     /* saber_hip_dtype. f32/f32 = FP32 conv; s8|u8 in = INT8 conv;
                                    f32 in + 8-bit weights = INT8 conv that quantises on entry
                                    (SaberConv2D<X86,AK_INT8>::dispatch, saber_conv.cpp:293-324) */
    int in_layout, out_layout;   /* 8-bit tensors are NHWC (calibrator_parse.cpp:194-244); f32 NCHW or NHWC */
    int act;                     /* ActivationParam of the conv (relu) */
    int res_mode;                /* saber_hip_res_mode */
    int res_act;                 /* activation of the eltwise (RES_ELTWISE / FP32 SUM_INPLACE) */
    float sum_scale;             /* RES_SUM_INPLACE INT8: multiplies the bytes already in y. The reference derives it in the
                                    impl, not the caller: beta / out_scale with a 255/127 (s8 added into u8) or 127/255
                                    (u8 into s8) factor (jit_avx512_core_x8s8s32x_conv.cpp:174-189); the adaptors do that */
    float coeff_conv, coeff_res; /* RES_ELTWISE: EltwiseParam.coeff[0], coeff[1] */
    float scale_res;             /* RES_ELTWISE: scale of the residual tensor */
    int int8_weights;            /* 1: INT8 arithmetic (AK_INT8 op), 0: FP32 arithmetic */
    float act_negative_slope;    /* FP32 convs, act == RELU: ActivationParam::negative_slope, "if (t < 0) t *= slope" (leaky ReLU;
                                    saber_im2col_conv.cpp:153-207, saber_conv_1x1.cpp:61-64). The x86 INT8 conv ignores it
                                    (gemm_x8s8s32x_conv.cpp:269-271 clamps to 0): INT8 ops reject a non-zero slope */
    int res_has_dtype;           /* RES_SUM_INPLACE INT8: 1 = res_dtype below is the dtype of the bytes already in y */
    int res_dtype;               /* ... ConvParam.beta_type (s8 or u8; may differ from out_dtype, same element size).
                                    res_has_dtype == 0: the bytes in y have out_dtype */
    int res_stride;              /* RES_ELTWISE: > 1 = the residual tensor is [n, res_h, res_w, k] and output pixel (oy, ox) adds
                                    res[oy * res_stride][ox * res_stride] — the 1x1 / stride-s max pooling that the reference's
                                    graph_strategy::apply_stride_up puts on a shortcut (optimize_strategy.h:213-248; one element
                                    per window, floor mode) folded into the epilogue's read. 0 / 1: res is [n, oh, ow, k] */
    int res_h, res_w;            /* ... the pooling's input dims: (res_h - 1) / res_stride + 1 == oh, likewise res_w */
} saber_hip_conv_desc;

typedef struct saber_hip_conv saber_hip_conv_t;

int saber_hip_conv2d_create(const saber_hip_conv_desc* desc, saber_hip_conv_t** out);
/* w: HOST pointer, OIHW [k, c/group, kh, kw]; w_dtype f32 or s8.
 *   INT8 op + f32 weights: quantised here exactly as scale_conv_weights_to_nchw_host does
 *   (per-out-channel max|w|/127, truncating cast; x86_utils.h:141-166,293-322), w_scale ignored.
 *   INT8 op + s8 weights : w_scale[k] required.
 * bias: HOST f32 [k] or NULL. in_scale / out_scale: the edge tensors' scales (Tensor::get_scale()[0]). */
int saber_hip_conv2d_set_weights(saber_hip_conv_t* op, const void* w, int w_dtype, const float* w_scale,
                                 const float* bias, float in_scale, float out_scale);
size_t saber_hip_conv2d_workspace_bytes(const saber_hip_conv_t* op);
void saber_hip_conv2d_out_shape(const saber_hip_conv_t* op, int* oh, int* ow);
/* x, y, res, workspace: DEVICE pointers. res may be NULL unless res_mode == RES_ELTWISE. */
int saber_hip_conv2d_run(saber_hip_conv_t* op, const void* x, void* y, const void* res, void* workspace,
                         saber_hip_stream_t stream);
void saber_hip_conv2d_destroy(saber_hip_conv_t* op);
/* SaberConv2DPooling (saber/funcs/conv_pooling.h; ConvPoolingParam, saber_funcs_param.h:647-677): attaches a pooling
 * stage to an INT8 conv. On success the op's output IS the pooled tensor (out_shape reports the pooled dims;
 * Pooling<>::compute_output_shape, pooling.h:92-121) and `run` is one kernel; the bytes equal pooling(conv(x)).
 * Returns SABER_HIP_UNIMPL when no fused kernel covers the combination — the caller then dispatches the conv into an
 * inner tensor and the pooling as a second op, as SaberConv2DPooling<X86,AK_FLOAT> does (saber_conv_pooling.cpp:13-57).
 * Fused today: (INT8) 7x7 / stride-2 conv with <= 4 input channels (the ResNet stem, optionally quantising its f32 input)
 * + 3x3 / stride-2 / pad-0 max pooling, s8 or u8 output; (FP32) any relu'd implicit-GEMM conv with NHWC output and even
 * output dims + 2x2 / stride-2 / pad-0 max pooling (VGG16's conv+relu+pool stages; reference: sass_funcs.h:366-427,
 * saber_conv_pooling.cpp). Call before the first run. */
int saber_hip_conv2d_set_pooling(saber_hip_conv_t* op, int pool_type, int kh, int kw, int stride_h, int stride_w,
                                 int pad_h, int pad_w, int floor_mode);
/* Debug / parity helpers: copy out the quantised weights (OIHW s8) and their scales. */
int saber_hip_conv2d_get_quantized_weights(const saber_hip_conv_t* op, int8_t* wq_oihw, float* w_scale);
/* Name of the kernel variant `run` will launch (e.g. "igemm_i8_64x64"), for profiling/tests. */
const char* saber_hip_conv2d_algo(const saber_hip_conv_t* op);
/* Implementation selection. `create` picks a tile statically (BaseFunc STATIC strategy,
 * saber/funcs/base.h:173-192); `autotune` is the RUNTIME strategy (base.h:194,205-247): it times every
 * tile of the implicit-GEMM kernel on the given device tensors and keeps the fastest.
 * set_tile argument: tile id in the low byte, stage depth (1/2/4) in bits 8..15, variant in bits 16..23:
 *   1 register-staged implicit GEMM, 2 LDS-DMA ring, 3 / 4 ring with 2 / 4 wave groups, 5 / 6 LDS-halo 3x3 (4 / 8 rows),
 *   7 / 8 stem kernel on / off, 9 small-image 3x3 (low byte = output rows per slab, bits 8..15 = images per slab),
 *   10 small-batch fc;
 *   11 FP32 implicit GEMM on three bf16 operand planes: tile 0..5, or 6..9 = the 8-wave forms of 64x64 / 128x64 / 128x128 / 256x128;
 *      bits 8..11 stage depth (1 | 2), bits 12..15 log2 of the split-K factor (2 / 4 / 8 workgroups per tile on one XCD);
 *   13 FP32 3x3 stride-1 pad-1 LDS-halo kernel on the bf16 planes, variant 1..5 in the low byte (channels per workgroup x tile rows x
 *      waves: 128x8x8, 64x8x4, 64x8x8, 64x4x4, 128x4x8); C % 32 == 0, NHWC;
 *      6..8: the pointwise (1x1) forms of the same kernel, C % 64 == 0;
 *   12 image-resident kernel (INT8 1x1 / 3x3 stride-1 convs on <= 64 pixels per image: workgroup = one image x a channel group,
 *      the image in LDS, the weight slice in registers);
 *   14 FP32 pointwise (1x1 / stride 1, NHWC, K % 64 == 0) kernels without LDS staging: low byte 0 = persistent waves with their
 *      weight planes in registers (C = 64 / 128), 1..4 = the reduction split over the four waves of a workgroup (C % 128 == 0;
 *      output channels x pixels per workgroup, 32-deep slabs in flight per wave: 64x32 d1 two workgroups per CU, 32x32 d3, 64x32 d2,
 *      64x64 d2; a variant that keeps more slabs in flight than a wave has - C / 128 - is refused). */
int saber_hip_conv2d_set_tile(saber_hip_conv_t* op, int tile);
int saber_hip_conv2d_get_tile(const saber_hip_conv_t* op);
int saber_hip_conv2d_autotune(saber_hip_conv_t* op, const void* x, void* y, const void* res, void* workspace,
                              saber_hip_stream_t stream, int iters);

/* SaberConv2D + global average Pooling<AK_INT8> in one launch (the last residual block of a ResNet feeding pool5): for an
 * INT8 NHWC conv on <= 64 pixels per image that has the image-resident kernel (1x1 / 3x3, stride 1, ResNet res5 channel
 * shapes; plain or fused-eltwise epilogue). After set_weights; the op then has this one kernel. run_gpool writes the conv's
 * output y as usual AND y_pool = [n][k] 8-bit of y's dtype: (float)(int32 sum over the h*w pixels of y) * (1 / (h*w)), round
 * to nearest even, saturate - the bytes of saber_hip_pool2d_i8_nhwc on y (the reference's JIT average pooling,
 * saber/funcs/impl/x86/kernel/jit_uni_pool_kernel: int32 accumulate, one multiply, cvt). SABER_HIP_UNIMPL where no such kernel exists. */
int saber_hip_conv2d_set_global_pooling(saber_hip_conv_t* op);
int saber_hip_conv2d_run_gpool(saber_hip_conv_t* op, const void* x, void* y, const void* res, void* y_pool,
                               saber_hip_stream_t stream);

/* Sibling pair: two convolutions (both INT8, or both FP32 with NHWC tensors) that read the SAME input tensor with the same geometry (kernel,
 * pad, stride, dilation; e.g. ResNet's stage-entry `branch1` projection and `branch2a`) executed by ONE
 * launch — the input tile is staged once and multiplied against both weight sets. An executor-level
 * optimisation (the reference dispatches the two Saber ops one after the other, net.cpp:417-509);
 * the outputs are bit-identical to running `a` and `b` separately. Both ops must already have their
 * weights set, be plain convs (no residual mode; INT8: 8-bit NHWC in- and outputs, FP32: NHWC in- and outputs),
 * a.k % 128 == 0 and b.k % 16 == 0. The pair keeps its own copy of the packed weights; a and b stay usable. */
int saber_hip_conv2d_create_pair(const saber_hip_conv_t* a, const saber_hip_conv_t* b, saber_hip_conv_t** out);
int saber_hip_conv2d_run_pair(saber_hip_conv_t* pair, const void* x, void* y_a, void* y_b, saber_hip_stream_t stream);
int saber_hip_conv2d_autotune_pair(saber_hip_conv_t* pair, const void* x, void* y_a, void* y_b,
                                   saber_hip_stream_t stream, int iters);

/* 1x1 chain: `a` = 1x1 stride-1 INT8 conv with the fused SaberEltwise epilogue (RES_ELTWISE, s8 residual and output: the
 * ResNet branch2c + sum + relu) and `b` = the 1x1 stride-1 s8-input conv that reads a's output (the next block's
 * branch2a) run as ONE launch on shared pixel tiles. Shapes C -> 4C -> C with C in {64, 128, 256, 512}. Both outputs are
 * written and hold the same bits as  saber_hip_conv2d_run(a) ; saber_hip_conv2d_run(b)  (reference: the two
 * GemmX8S8S32XConv::dispatch calls + SaberEltwise, gemm_x8s8s32x_conv.cpp:184-257). The chain refers to a and b (they
 * must outlive it and keep their weights) and owns only the repacked weight stream. */
typedef struct saber_hip_chain saber_hip_chain_t;
int saber_hip_conv2d_chain_create(saber_hip_conv_t* a, saber_hip_conv_t* b, saber_hip_chain_t** out);
/* ... led by the block's 3x3 conv (stride 1, pad 1, C -> C with C in {64, 128, 256}, 8-bit output = a's input): three
 * operators, one launch. saber_hip_conv2d_chain_run then takes the 3x3 conv's INPUT as x; the 3x3 conv's own output edge
 * is not written (it lives in LDS), y_a / y_b hold the bits of running the three ops one after the other. The pixel
 * tile is `tn` rows x 16 columns (set_tile: 4 | 2 for C = 64, 2 | 1 for C = 128, 1 for C = 256). */
int saber_hip_conv2d_chain_create3(saber_hip_conv_t* conv3x3, saber_hip_conv_t* a, saber_hip_conv_t* b, saber_hip_chain_t** out);
/* ... and the last block of a stage after the reference's stride-up (conv3x3 / stride 2 + conv1x1 + eltwise on a sub-sampled
 * shortcut, C = 64: ResNet's res2c) followed in the SAME launch by the next stage's sibling pair `pair_a` / `pair_b` - the two 1x1
 * stride-1 convs that read a's output (res3a_branch1 / res3a_branch2a; 256 -> k_a | k_b, k_a + k_b = 640, k_a % 32 == 0, 8-bit
 * outputs): four operators, one launch. saber_hip_conv2d_chain_run3 writes y_a (a's output), y_b / y_c (the pair's). */
int saber_hip_conv2d_chain_create3_pair(saber_hip_conv_t* conv3x3, saber_hip_conv_t* a, saber_hip_conv_t* pair_a, saber_hip_conv_t* pair_b,
                                        saber_hip_chain_t** out);
void saber_hip_conv2d_chain_destroy(saber_hip_chain_t* chain);
int saber_hip_conv2d_chain_run(saber_hip_chain_t* chain, const void* x, const void* res, void* y_a, void* y_b,
                               saber_hip_stream_t stream);
int saber_hip_conv2d_chain_run3(saber_hip_chain_t* chain, const void* x, const void* res, void* y_a, void* y_b, void* y_c,
                                saber_hip_stream_t stream);
/* pixel fragments (16 pixels each) per workgroup: 4 | 2 for C = 64, 2 | 1 for C = 128, 1 otherwise; C >= 256 also takes
 * 9 = 1 fragment with the second conv's output channels split over two workgroups (both run the first conv);
 * C = 256 also 11 = the same with 8 waves per workgroup; C = 128 also 6 | 5 = 2 | 1 fragments with 8 waves */
int saber_hip_conv2d_chain_set_tile(saber_hip_chain_t* chain, int tn);
int saber_hip_conv2d_chain_get_tile(const saber_hip_chain_t* chain);

/* A RUN of such 3x3-led chains at C = 256 (the blocks of ResNet's res4 stage; chain i + 1 reads chain i's two outputs) as ONE
 * persistent launch: four cooperating workgroups per tile of 2 x 16 pixels, all tiles of an image on one XCD, an XCD-local barrier
 * between two blocks instead of a kernel boundary (anakin_amd/csrc/conv_stage_coop.hip). Like the chains an executor-level
 * fusion with no counterpart in the reference (framework/core/net/net.cpp:417-509 dispatches one operator at a time); results
 * bit-identical to the separate launches. Batch <= 8, <= 8 tiles per image (14 x 14). Also for runs of >= 2 chains at C = 128
 * (res3: one workgroup per tile, width <= 64, <= 32 tiles per image; measured slower than its chain launches on MI355X - an
 * autotune candidate). The chains are not owned and must outlive
 * the stage. y1[i] / y2[i]: chain i's outputs (saber_hip_conv2d_chain_run's y_a / y_b); x / res: chain 0's inputs. */
typedef struct saber_hip_chain_stage saber_hip_chain_stage_t;
int saber_hip_conv2d_stage_create(saber_hip_chain_t* const* chains, int n, saber_hip_chain_stage_t** out);
void saber_hip_conv2d_stage_destroy(saber_hip_chain_stage_t* stage);
int saber_hip_conv2d_stage_run(saber_hip_chain_stage_t* stage, const void* x, const void* res, void* const* y1, void* const* y2,
                               saber_hip_stream_t stream);

/* The ResNet stem with its first two consumers in ONE launch: `stem` (an INT8 conv with saber_hip_conv2d_set_pooling's fused 3x3 /
 * stride-2 max pooling, 64 output channels: SaberConv2DPooling<AK_INT8>, saber_conv_pooling.cpp:60-160) followed, on the workgroup's
 * pooled pixels, by the two 1x1 / stride-1 INT8 convs `a` and `b` that read the pooled tensor (res2a's branch1 and branch2a; 64 -> k,
 * k % 32 == 0, k_a + k_b <= 320, 8-bit NHWC outputs; GemmX8S8S32XConv::dispatch, gemm_x8s8s32x_conv.cpp:184-257) - an executor-level
 * fusion like the chains, results bit-identical to the three launches. y_pool may be null when nothing else reads the pooled tensor.
 * The ops are not owned and must outlive the object. */
typedef struct saber_hip_stem_pair saber_hip_stem_pair_t;
int saber_hip_conv2d_stem_pair_create(saber_hip_conv_t* stem, const saber_hip_conv_t* a, const saber_hip_conv_t* b, saber_hip_stem_pair_t** out);
void saber_hip_conv2d_stem_pair_destroy(saber_hip_stem_pair_t* sp);
int saber_hip_conv2d_stem_pair_run(saber_hip_stem_pair_t* sp, const void* x, void* y_pool, void* y_a, void* y_b, void* workspace,
                                   saber_hip_stream_t stream);

/* XCD-resident stage: a run of INT8 convolutions over SMALL feature maps (h * w <= 64 pixels per image: ResNet's res5) as
 * ONE persistent launch. Image i is computed entirely on XCD i % 8 (32 CUs, one workgroup each); the convolutions
 * ("phases") follow each other inside the kernel, separated by an XCD-local barrier where one reads what an earlier one
 * wrote - the edge tensors are handed over through that XCD's L2 instead of across a kernel boundary. Every edge is written
 * to its tensor as usual and holds the bits of running the phases' ops one after the other with saber_hip_conv2d_run
 * (reference: the same sequence of GemmX8S8S32XConv::dispatch / SaberEltwise calls, net.cpp:417-509).
 * Phase i runs convs `conv` on tensor slot `in` -> slot `out` (+ slot `res` for an op with the fused eltwise epilogue, -1
 * otherwise). The ops: 1x1 (pad 0) or 3x3 (pad 1) INT8 NHWC convs, stride 1, group 1, 8-bit in / out, weights set, all
 * with the same n / h / w; channel shapes as in ResNet's res5 (1024 -> 2048, 1024 -> 512, 3x3 512 -> 512, 512 -> 2048,
 * 2048 -> 512). The stage refers to the ops (they must outlive it) and owns the repacked weights.
 * The launch needs all 256 CUs to itself for its duration (a second stream's kernels holding CUs delay it; two stages in
 * flight on different streams can starve each other): a workgroup that waits ~20 ms gives up, the outputs are then
 * garbage and saber_hip_stage_status (synchronises the device) returns SABER_HIP_RUNTIME_ERROR once and re-arms the stage.
 * One stage object must not be in flight on two streams at once.
 * MEASURED (profiles/r03/stage_trace.txt): correct, but at batch 8 / 16 slower than the separate launches (104 vs 73 us for
 * res5): every XCD pulls the whole stage's weights (8 x 15 MB through the per-XCD fabric ports, ~2.7 us + 1.5 us / MB per
 * phase) - saber_hip_net_optimize therefore does not form stages; the entry points stay for the measurement. */
typedef struct saber_hip_stage saber_hip_stage_t;
typedef struct {
    saber_hip_conv_t* conv;
    int in, out, res;
} saber_hip_stage_phase;
int saber_hip_stage_create(const saber_hip_stage_phase* phases, int n_phases, saber_hip_stage_t** out);
int saber_hip_stage_num_tensors(const saber_hip_stage_t* stage);
int saber_hip_stage_run(saber_hip_stage_t* stage, void* const* tensors, int n_tensors, saber_hip_stream_t stream);
int saber_hip_stage_status(saber_hip_stage_t* stage);
/* Diagnostics: out == NULL arms the trace (returns the number of 64-bit words a read needs); afterwards every launch
 * records, per workgroup and phase, 8 stamps of the 100 MHz wall clock (phase start, arrived, weights issued, barrier
 * passed, image DMA issued, image in LDS, matrix loop done, phase end); a call with a buffer synchronises and copies them
 * out as [256][n_phases][8]. */
int saber_hip_stage_trace(saber_hip_stage_t* stage, unsigned long long* out, size_t cap);
void saber_hip_stage_destroy(saber_hip_stage_t* stage);

/* ------------------------------------------------------------------------------------------- */
/* Fully connected (SaberFc / VenderFc), FP32 and INT8                                          */
/* ------------------------------------------------------------------------------------------- */
typedef struct {
    int m, n, k;          /* out[m,n] = in[m,k] * W[n,k]^T + bias[n] */
    int in_dtype;         /* f32 (FP32 op, or INT8 op quantising on entry), s8, u8 */
    int int8_weights;
    int w_is_kn;          /* FcParam.is_transpose_weights: weights stored [k,n] */
} saber_hip_fc_desc;
typedef struct saber_hip_fc saber_hip_fc_t;
int saber_hip_fc_create(const saber_hip_fc_desc* desc, saber_hip_fc_t** out);
int saber_hip_fc_set_weights(saber_hip_fc_t* op, const void* w, int w_dtype, const float* w_scale,
                             const float* bias, float in_scale, float out_scale);
size_t saber_hip_fc_workspace_bytes(const saber_hip_fc_t* op);
/* out is f32 [m,n]. */
int saber_hip_fc_run(saber_hip_fc_t* op, const void* x, float* y, void* workspace, saber_hip_stream_t stream);
/* INT8 fc on an input that is ALREADY quantised to s8 with the op's in_scale (e.g. by pool2d_f32_from_i8_q): skips
 * the quantise-on-entry kernel of an f32-input INT8 fc; identical result. */
int saber_hip_fc_run_q(saber_hip_fc_t* op, const int8_t* xq, float* y, saber_hip_stream_t stream);
/* Fc followed by Softmax over its [m][n] output (the tail of every classifier: framework/operators/dense.cpp -> softmax.cpp; Saber:
 * saber/funcs/fc.h:48-127 + saber/funcs/softmax.h) as ONE launch where the INT8 small-batch weight-streaming kernel runs the fc
 * (m <= 16, n <= 1024, 8-bit operand, reduction 512 / 1024 / 2048 / 4096): the workgroup that arrives last on a device-wide counter
 * normalises the rows. y (the logits) is written as by saber_hip_fc_run; prob = softmax(y) per row, the arithmetic of
 * saber_hip_softmax_f32 with the row sum taken lane-major (within the 1e-4 the softmax output is held to). Every other case runs
 * saber_hip_fc_run + saber_hip_softmax_f32: the result is the same either way. saber_hip_net_optimize flag 4096 forms it.
 * The arrival counter belongs to `op`: at most ONE launch of an fc object may be in flight at a time (launches of one stream are
 * ordered; the same object on two streams / in two concurrently replayed graphs is the caller's error - the reference never calls an
 * impl instance concurrently). SABER_HIP_FC_SOFTMAX_FENCED=1 selects the release / acquire-fence form of the hand-off (A/B, slower). */
int saber_hip_fc_run_softmax(saber_hip_fc_t* op, const void* x, float* y, float* prob, void* workspace, saber_hip_stream_t stream);
/* Kernel selection of the INT8 fc, same encoding as saber_hip_conv2d_set_tile: variant 10 (<< 16) = the small-batch
 * weight-streaming kernel (m <= 16, k <= 4096; the STATIC choice when eligible), 1..4 = implicit-GEMM variants. */
const char* saber_hip_fc_algo(const saber_hip_fc_t* op);
int saber_hip_fc_set_tile(saber_hip_fc_t* op, int tile);
void saber_hip_fc_destroy(saber_hip_fc_t* op);

/* ------------------------------------------------------------------------------------------- */
/* GEMM (Gemm<T, SABER_IMPL, float, float>): row-major C = alpha*op(A)*op(B) + beta*C, raw pointers */
/* ------------------------------------------------------------------------------------------- */
int saber_hip_gemm_f32(int trans_a, int trans_b, int m, int n, int k, float alpha, const float* a,
                       const float* b, float beta, float* c, saber_hip_stream_t stream);
/* The entry point above is stateless for the caller, its device scratch (weight planes, transpose buffer = Gemm<>::init's state) is
 * not: it lives in plans cached per calling thread, keyed by (device, stream, transposes, m, n, k, beta class), at most 16 unpinned
 * ones (least recently used evicted after draining its stream). A call recorded into a hipGraph (the stream is capturing) pins the
 * plan it uses - never evicted, not shared with eager calls - or, when the shape was never run eagerly on that stream, records the
 * f32-MFMA kernel that owns no state. This frees the calling thread's plans (pinned ones included: the caller's graphs over them
 * must be gone) after draining their streams; returns how many. */
int saber_hip_gemm_f32_release_plans(void);

/* INT8 GEMM (MklDnnGemm<int8_t | uint8_t, int8_t, int>, saber/funcs/impl/x86/mkl_gemm.cpp:138-256; its test
 * test/saber/test_saber_gemm_int8.cpp): row-major C[m,n] (int32) = op(A)[m,k] x op(B)[k,n], exact integer
 * arithmetic (i8 MFMA, int32 accumulate). B (s8, HOST pointer) is constant and packed at create time — the
 * reference's PACKED_MKLGEMM mode; A (s8 or u8) and C are device pointers. trans_a: A is stored [k,m];
 * trans_b: B is stored [n,k]. Workspace (device) is needed when trans_a or k % 16 != 0. */
typedef struct saber_hip_gemm_i8 saber_hip_gemm_i8_t;
int saber_hip_gemm_i8_create(int trans_a, int trans_b, int m, int n, int k, int a_dtype, const int8_t* b_host,
                             saber_hip_gemm_i8_t** out);
size_t saber_hip_gemm_i8_workspace_bytes(const saber_hip_gemm_i8_t* g);
int saber_hip_gemm_i8_run(saber_hip_gemm_i8_t* g, const void* a, int32_t* c, void* workspace, saber_hip_stream_t stream);
void saber_hip_gemm_i8_destroy(saber_hip_gemm_i8_t* g);

/* ------------------------------------------------------------------------------------------- */
/* Quantise / dequantise + layout (reorder_nhwc_nchw)                                           */
/* ------------------------------------------------------------------------------------------- */
/* f32 NCHW [n,c,h,w] -> s8/u8 NHWC [n,h,w,c_pad] (channels c..c_pad-1 written as 0; c_pad >= c). */
int saber_hip_quantize_nchw_to_nhwc(int n, int c, int h, int w, int c_pad, int out_dtype, float scale,
                                    const float* x, void* y, saber_hip_stream_t stream);
/* s8/u8 NHWC -> f32 NCHW */
int saber_hip_dequantize_nhwc_to_nchw(int n, int c, int h, int w, int in_dtype, float scale, const void* x,
                                      float* y, saber_hip_stream_t stream);
/* f32 layout transforms (c_pad: channel padding of the NHWC side, zero filled) */
int saber_hip_transpose_nchw_to_nhwc_f32(int n, int c, int h, int w, int c_pad, const float* x, float* y,
                                         saber_hip_stream_t stream);
int saber_hip_transpose_nhwc_to_nchw_f32(int n, int c, int h, int w, int c_pad, const float* x, float* y,
                                         saber_hip_stream_t stream);
/* flat f32 -> s8 with ScaleUtils::scale_fp32_int8 semantics (x86_utils.h:325-346) */
int saber_hip_quantize_flat_s8(size_t count, float scale, const float* x, int8_t* y, saber_hip_stream_t stream);

/* ------------------------------------------------------------------------------------------- */
/* Eltwise sum (+relu), pooling, softmax — the "next" rows of SURVEY.md §8(f)                   */
/* ------------------------------------------------------------------------------------------- */
int saber_hip_eltwise_sum_i8(size_t count, const int8_t* a, const int8_t* b, float scale_a, float scale_b,
                             float coeff_a, float coeff_b, int relu, int8_t* y, saber_hip_stream_t stream);
int saber_hip_eltwise_sum_f32(size_t count, const float* a, const float* b, float coeff_a, float coeff_b,
                              int relu, float* y, saber_hip_stream_t stream);
/* standalone ReLU op (framework/operators/relu.cpp -> Activation<T,D>, Active_relu; saber_activation.cpp:136-154):
 * y = x > 0 ? x : 0 over `count` f32 elements; x == y (in place) is allowed */
int saber_hip_relu_f32(size_t count, const float* x, float* y, saber_hip_stream_t stream);
/* The other activation types of the standalone Activation operator (Activation<T, AK_FLOAT>: saber/funcs/activation.h,
 * x86: saber/funcs/impl/x86/saber_activation.cpp:156-262), elementwise on f32, in place allowed. `active` is the reference's
 * ActiveType value (saber/saber_types.h:283-293): sigmoid 1, relu 2 (= saber_hip_relu_f32), tanh 3, clipped relu 4 (threshold =
 * coef), elu 5 (coef), stanh 9 (coef * tanh(negative_slope * x)), gelu 11, swish 12 (beta = coef); anything else: UNIMPL.
 * PReLU (Active_prelu 10; excute_prelu :38-132): y = x > 0 ? x : x * slope[channel], channel = (i / inner) % channels
 * (NCHW: inner = H * W; NHWC: inner = 1), channel_shared: slope[0]; `slope` is device memory. Also what a Conv with a non-relu
 * activation runs after the convolution (the adaptor, as the NV impl does: saber/funcs/impl/cuda/saber_conv.cpp _saber_act). */
int saber_hip_activation_f32(int active, size_t count, float negative_slope, float coef, const float* x, float* y,
                             saber_hip_stream_t stream);
int saber_hip_prelu_f32(size_t count, int channels, int inner, int channel_shared, const float* slope, const float* x, float* y,
                        saber_hip_stream_t stream);
/* Pooling<>::compute_output_shape (pooling.h:69-130) */
int saber_hip_pool_out_dim(int in, int pad, int window, int stride, int floor_mode);
/* The same for one dimension of a pooling whose OTHER dimension may be padded: the reference clips the last window of
 * both dimensions whenever pad_h || pad_w (pooling.h:113-120); any_pad = (pad_h > 0 || pad_w > 0). */
int saber_hip_pool_out_dim2(int in, int pad, int window, int stride, int floor_mode, int any_pad);
/* NHWC s8/u8 -> s8/u8 (max, avg) or f32 (avg) */
int saber_hip_pool2d_i8_nhwc(int n, int h, int w, int c, int oh, int ow, int kh, int kw, int stride_h,
                             int stride_w, int pad_h, int pad_w, int pool_type, int in_dtype, int out_dtype,
                             const void* x, void* y, saber_hip_stream_t stream);
/* f32, NHWC or NCHW (same layout in and out) */
int saber_hip_pool2d_f32(int n, int h, int w, int c, int oh, int ow, int kh, int kw, int stride_h,
                         int stride_w, int pad_h, int pad_w, int pool_type, int layout, const float* x,
                         float* y, saber_hip_stream_t stream);
/* FP32 pooling op fed an 8-bit NHWC tensor (SaberPooling<X86,AK_FLOAT> dequantises on entry,
 * saber_pooling.cpp:399-402): s8/u8 NHWC in, f32 NCHW out; same float sequence as dequantize + pool2d_f32. */
int saber_hip_pool2d_f32_from_i8(int n, int h, int w, int c, int oh, int ow, int kh, int kw, int stride_h,
                                 int stride_w, int pad_h, int pad_w, int pool_type, int in_dtype, float scale,
                                 const void* x, float* y, saber_hip_stream_t stream);
/* Same, and additionally writes yq = saturate_s8(roundf(y * (1/q_scale))) (flat, y's order): the quantise-on-entry of
 * a following INT8 op (PackedMKLInt8Gemm::dispatch -> scale_fp32_int8, mkl_packed_int8_gemm.cpp:52-57) fused into
 * the pooling's store. yq may be NULL. */
int saber_hip_pool2d_f32_from_i8_q(int n, int h, int w, int c, int oh, int ow, int kh, int kw, int stride_h,
                                   int stride_w, int pad_h, int pad_w, int pool_type, int in_dtype, float scale,
                                   const void* x, float* y, float q_scale, int8_t* yq, saber_hip_stream_t stream);
/* softmax over the last axis of [rows, cols] */
int saber_hip_softmax_f32(int rows, int cols, const float* x, float* y, saber_hip_stream_t stream);

/* ------------------------------------------------------------------------------------------- */
/* Op-list executor: the device-side half of Net<T,P,R>::prediction (net.cpp:417-509)           */
/* ------------------------------------------------------------------------------------------- */
typedef struct saber_hip_net saber_hip_net_t;
int saber_hip_net_create(saber_hip_net_t** out);
/* Declares an edge tensor of `bytes` bytes; returns its id (>= 0). */
int saber_hip_net_add_tensor(saber_hip_net_t* net, size_t bytes);
/* The ops take ownership of nothing; conv/fc handles must outlive the net. */
int saber_hip_net_add_conv(saber_hip_net_t* net, saber_hip_conv_t* op, int in_id, int out_id, int res_id);
int saber_hip_net_add_conv_pair(saber_hip_net_t* net, saber_hip_conv_t* pair, int in_id, int out_a_id, int out_b_id);
int saber_hip_net_add_fc(saber_hip_net_t* net, saber_hip_fc_t* op, int in_id, int out_id);
int saber_hip_net_add_quantize(saber_hip_net_t* net, int n, int c, int h, int w, int c_pad, int out_dtype,
                               float scale, int in_id, int out_id);
int saber_hip_net_add_dequantize(saber_hip_net_t* net, int n, int c, int h, int w, int in_dtype, float scale,
                                 int in_id, int out_id);
int saber_hip_net_add_transpose_in_f32(saber_hip_net_t* net, int n, int c, int h, int w, int c_pad, int in_id,
                                       int out_id);
int saber_hip_net_add_eltwise_i8(saber_hip_net_t* net, size_t count, float scale_a, float scale_b,
                                 float coeff_a, float coeff_b, int relu, int a_id, int b_id, int out_id);
int saber_hip_net_add_eltwise_f32(saber_hip_net_t* net, size_t count, float coeff_a, float coeff_b, int relu,
                                  int a_id, int b_id, int out_id);
int saber_hip_net_add_pool_i8(saber_hip_net_t* net, int n, int h, int w, int c, int oh, int ow, int kh, int kw,
                              int stride_h, int stride_w, int pad_h, int pad_w, int pool_type, int in_dtype,
                              int out_dtype, int in_id, int out_id);
int saber_hip_net_add_pool_f32(saber_hip_net_t* net, int n, int h, int w, int c, int oh, int ow, int kh, int kw,
                               int stride_h, int stride_w, int pad_h, int pad_w, int pool_type, int layout,
                               int in_id, int out_id);
int saber_hip_net_add_pool_f32_from_i8(saber_hip_net_t* net, int n, int h, int w, int c, int oh, int ow, int kh,
                                       int kw, int stride_h, int stride_w, int pad_h, int pad_w, int pool_type,
                                       int in_dtype, float scale, int in_id, int out_id);
int saber_hip_net_add_pool_f32_from_i8_q(saber_hip_net_t* net, int n, int h, int w, int c, int oh, int ow, int kh, int kw,
                                         int stride_h, int stride_w, int pad_h, int pad_w, int pool_type, int in_dtype,
                                         float scale, int in_id, int out_id, float q_scale, int q_out_id);
int saber_hip_net_add_fc_q(saber_hip_net_t* net, saber_hip_fc_t* op, int in_q_id, int out_id);
/* Executor-level fusions on an op list added UNFUSED (one op per reference operator), before finalize: the counterpart
 * of the reference's graph optimiser (framework/graph/llvm/fusion/fusion_op_register.cpp:45-175) for this executor.
 * flags: 1 conv + INT8 eltwise -> fused epilogue, 2 sibling convs -> one pair launch, 4 conv + max pooling ->
 * SaberConv2DPooling where a fused kernel exists, 8 global pooling also writes the INT8 fc's quantised operand;
 * 16 a 1x1 conv with the fused eltwise epilogue + the 1x1 conv that reads its output -> one conv1x1-chain launch (both ops
 * stay in the list; while the chain is selected the second one launches nothing; saber_hip_net_autotune keeps whichever
 * form is faster); 32 (with 16) the block's 3x3 conv leads that chain launch when the chain head is its only consumer
 * (its output edge is then not written: saber_hip_net_tensor_unwritten); 64 a 1x1 / stride-s max pooling (the shortcut
 * pooling the reference's stride-up pass inserts) whose only reader is a fused eltwise epilogue -> folded into that read
 * (saber_hip_conv_desc::res_stride; the pooled edge no longer exists); 128 a conv on <= 64-pixel images followed by the global
 * average Pooling<AK_INT8> of its output -> saber_hip_conv2d_set_global_pooling (both tensors still written, one launch);
 * 255 = all of these. 256 (with 16 | 32; NOT in 255): a run of 3x3-led C = 256 chains whose blocks feed each other (ResNet's res4
 * stage) -> one persistent launch (saber_hip_conv2d_stage_create); every op stays in the list, saber_hip_net_autotune keeps the
 * faster form. Only for a net that has the GPU to itself while it runs: the launch needs all workgroups of an image resident on
 * one XCD together, and two such launches in flight on different streams can starve each other (they time out after ~20 ms,
 * the next run returns SABER_HIP_RUNTIME_ERROR and the chains launch one by one from then on).
 * 512 (with 2 | 4; NOT in 255): the fused stem conv + max pooling whose pooled tensor is read by one sibling pair of 1x1 convs only
 * (ResNet's conv1 + pool1 -> res2a_branch1 / res2a_branch2a) runs that pair in its own launch (saber_hip_conv2d_stem_pair_create);
 * the pooled edge is then not written (saber_hip_net_tensor_unwritten), the pair stays in the list and launches nothing.
 * 1024 (with 2 | 16 | 32; NOT in 255): the strided head of a stage at C = 64 (conv3x3 / stride 2 + conv1x1 + eltwise: ResNet's res2c)
 * whose output is read by one sibling pair only (res3a_branch1 / res3a_branch2a) runs that pair in its chain launch
 * (saber_hip_conv2d_chain_create3_pair); the pair stays in the list and launches nothing while the chain form is selected.
 * 4096 (NOT in 255): an fc (FC / FC on a quantised operand) whose only reader is a Softmax over its output -> saber_hip_fc_run_softmax
 * (one launch where the small-batch INT8 kernel is selected; the softmax op stays in the list and launches nothing).
 * 2048 (SABER_HIP_NET_SHARED_DEVICE; NOT in 255; may be passed on its own, sticks to the net): the net does NOT have the device to
 * itself - other streams, Worker threads or processes run kernels there while it does (framework/core/net/worker.h:38-60: one Net
 * per pool thread). Every kernel variant whose completion or speed depends on where the hardware places workgroups relative to
 * each other is then excluded AT SELECTION TIME - flag 256 is ignored, the cooperating-workgroup chains (tile codes 7 / 15) and
 * FP32 split-K through one XCD's L2 are neither chosen statically, nor offered to saber_hip_net_autotune, nor accepted from a
 * restored selection (saber_hip_net_set_choice maps them to their plain forms) - instead of being found out by a timed-out
 * hand-off (~20 ms) at run time.
 * 8192 (SABER_HIP_NET_REPRODUCIBLE_FP32; NOT in 255; may be passed on its own, sticks to the net): FP32 ops keep their STATIC kernel
 * selection - saber_hip_net_autotune skips them and saber_hip_net_set_choice leaves them alone. The FP32 kernel families differ in
 * accumulation order (all within the 1e-4 contract), and a timing-based choice depends on the box and the moment: with this flag two
 * nets built from one model answer bit-identically (the reference's x86 FP32 path is deterministic for a fixed thread count; this is
 * the switch that gives a maintainer the same property, at the static selection's speed). INT8 ops are exact under every selection
 * and stay tunable.
 * Bytes of every surviving edge are unchanged. Returns the number of launches removed (>= 0) or a status < 0. */
#define SABER_HIP_NET_SHARED_DEVICE 2048
#define SABER_HIP_NET_REPRODUCIBLE_FP32 8192
int saber_hip_net_optimize(saber_hip_net_t* net, int flags);
/* How many cooperative launches of this net have reported a failed pass since it was created (each made saber_hip_net_status /
 * the site's next launch return SABER_HIP_RUNTIME_ERROR once and its site fall back to single-workgroup launches): 0 on a net that
 * owns its device or was optimised with SABER_HIP_NET_SHARED_DEVICE. */
int saber_hip_net_coop_fallbacks(const saber_hip_net_t* net);
/* ... summed over every net this process has created (a Worker's Nets live inside its pool threads): what a serving loop reports. */
int saber_hip_coop_fallbacks_total(void);
/* After a forward pass has COMPLETED (the caller synchronised the stream): SABER_HIP_RUNTIME_ERROR when one of its cooperative
 * launches (a stage launch, a two-workgroup chain launch) found its workgroups on different XCDs or timed out in a hand-off - its
 * outputs are not valid; those sites launch block by block from then on (a captured graph is dropped): run the pass again.
 * Without this call the site's NEXT launch returns the error. saber_hip_net_inject_coop_error: a testing aid that makes the next
 * saber_hip_net_status report such a failure at the net's first cooperative site. */
int saber_hip_net_status(saber_hip_net_t* net);
int saber_hip_net_inject_coop_error(saber_hip_net_t* net);
/* > 0: op `index` heads a stage of that many blocks (flag 256); bit 30 of its saber_hip_net_get_choice / _set_choice value says
 * whether the stage launch is selected */
int saber_hip_net_stage_blocks(const saber_hip_net_t* net, int index);
/* kernel launches of one forward pass (ops minus the ones absorbed into a chain launch) */
int saber_hip_net_num_launches(const saber_hip_net_t* net);
/* 1 when tensor `id` is never written: the output edge of a 3x3 conv currently running inside a conv3x3 + chain launch, or an
 * edge saber_hip_net_optimize removed (a fused conv's own output, an absorbed pooling's output: it has no storage) */
int saber_hip_net_tensor_unwritten(const saber_hip_net_t* net, int id);
int saber_hip_net_add_softmax(saber_hip_net_t* net, int rows, int cols, int in_id, int out_id);
/* Lane of an op (graph::Lane, framework/core/net/operator_func.h:103-114; ParallScheduler): 0 = the caller's
 * stream, 1 = the net's side stream. Cross-lane tensor dependencies are ordered with events automatically and
 * become parallel branches of the captured hipGraph. Must be set before the first run. */
int saber_hip_net_set_lane(saber_hip_net_t* net, int op_index, int lane);
/* Allocates every edge tensor + the shared workspace (one hipMalloc arena). */
int saber_hip_net_finalize(saber_hip_net_t* net);
void* saber_hip_net_tensor_ptr(saber_hip_net_t* net, int id);
size_t saber_hip_net_arena_bytes(const saber_hip_net_t* net);
/* Lifetime aliasing of the arena - what the reference's memory planner does for the edges of a Net
 * (framework/graph/llvm/optimizer/memory_scheduler.cpp; Graph::Optimize, graph.cpp:351-477): after finalize (and after autotune - the
 * autotuner re-runs single ops on the operands a whole pass left behind) re-lay the arena out so that tensors whose lifetimes do not
 * overlap share memory; ops that may run as one launch in some kernel selection count as one time step. Never aliased: caller-owned
 * tensors, the pass's inputs (no op writes them), its outputs (no op reads them) and the `keep` ids (may be NULL / 0). Afterwards only
 * those tensors hold defined data after a pass, pointers from saber_hip_net_tensor_ptr must be fetched again, a captured hipGraph is
 * dropped. A net with a side lane is left as it is. saber_hip_net_arena_bytes reports the new footprint. */
int saber_hip_net_compact_arena(saber_hip_net_t* net, const int* keep, int n_keep);
int saber_hip_net_arena_compacted(const saber_hip_net_t* net);
int saber_hip_net_num_ops(const saber_hip_net_t* net);
/* Enqueue every op in order on `stream` (eager launches). */
int saber_hip_net_run(saber_hip_net_t* net, saber_hip_stream_t stream);
/* Enqueue only op `index` (per-op timing / parity checks). */
int saber_hip_net_run_op(saber_hip_net_t* net, int index, saber_hip_stream_t stream);
/* Capture the op list into a hipGraph once; later runs replay it with one launch. */
int saber_hip_net_capture(saber_hip_net_t* net, saber_hip_stream_t stream);
int saber_hip_net_replay(saber_hip_net_t* net, saber_hip_stream_t stream);
/* Per-op device time in microseconds measured with hipEvents on `stream` (iters launches each,
 * eager). out_us has saber_hip_net_num_ops() entries. */
int saber_hip_net_time_ops(saber_hip_net_t* net, saber_hip_stream_t stream, int iters, float* out_us);
/* The same inside a forward pass: one event after every launch of an eager pass, out_us[i] = event[i] - event[i-1] averaged
 * over `iters` passes (the op in its place in the pipeline, kernel boundary included; ops absorbed into a chain launch: 0).
 * The events lengthen the pass: use the shares, scaled to an untimed step. */
int saber_hip_net_time_pass(saber_hip_net_t* net, saber_hip_stream_t stream, int iters, float* out_us);
/* ONE op's launch duration inside a forward pass, undisturbed: whole eager passes with only two events, in front of and behind
 * that op's launch (the per-launch markers of saber_hip_net_time_pass stretch the pass; this figure is the one that agrees with
 * a rocprofv3 kernel trace - bench.py uses it for the dominant kernel's roofline) */
int saber_hip_net_time_op_in_pass(saber_hip_net_t* net, saber_hip_stream_t stream, int index, int iters, float* out_us);
/* Algorithmic work of one launch of op `index` (input + output + residual activations once, weights once;
 * 2 x MACs), summed over the operators the launch covers (chain / pair launches). */
int saber_hip_net_op_work(const saber_hip_net_t* net, int index, double* bytes, double* flops);
const char* saber_hip_net_op_name(const saber_hip_net_t* net, int index);
/* RUNTIME strategy over every conv/fc op of the list (on whatever the edge tensors currently hold). */
/* Kernel selection of op `index` in the saber_hip_conv2d_get_tile / set_tile encoding (0: the op has none). set applies a
 * selection taken from get (e.g. in an earlier process: every profiling pass can run the same autotuned kernels). */
int saber_hip_net_get_choice(saber_hip_net_t* net, int index);
int saber_hip_net_set_choice(saber_hip_net_t* net, int index, int choice);
int saber_hip_net_autotune(saber_hip_net_t* net, saber_hip_stream_t stream, int iters);
void saber_hip_net_destroy(saber_hip_net_t* net);
/* standalone ReLU operator inside an op list (Activation<T,D>, Active_relu; VGG16's fc6 / fc7) */
int saber_hip_net_add_relu_f32(saber_hip_net_t* net, size_t count, int in_id, int out_id);
int saber_hip_net_add_activation_f32(saber_hip_net_t* net, int active, size_t count, float negative_slope, float coef, int in_id,
                                     int out_id);
/* Caller-owned storage for tensor `id` instead of a slot of the net's arena (the net's inputs and outputs when the caller
 * has buffers of its own: the reference's Net owns its edge tensors). Before finalize any tensor can be bound (ptr != NULL)
 * or returned to the arena (NULL); after finalize only the address of an already external tensor can change. A captured
 * hipGraph is dropped (capture again). */
int saber_hip_net_bind_tensor(saber_hip_net_t* net, int id, void* ptr);
int saber_hip_net_num_tensors(const saber_hip_net_t* net);
size_t saber_hip_net_tensor_bytes(const saber_hip_net_t* net, int id);

/* ------------------------------------------------------------------------------------------- */
/* Op-list capture: a caller's OWN op loop (Net<T,P,R>::prediction, net.cpp:417-509) -> saber_hip_net */
/* ------------------------------------------------------------------------------------------- */
/* The analogue of hipStreamBeginCapture one level up: between begin and end every saber_hip_conv2d_run / saber_hip_fc_run[_q] /
 * quantize / dequantize / transpose_nchw_to_nhwc / eltwise / relu / pool2d / softmax call made BY THE CALLING THREAD is
 * recorded into a new op list instead of being launched (stream arguments are ignored, workspaces come from the net's
 * arena later). Tensors are identified by the device pointers of the calls, renamed SSA-style: a write to an address starts
 * a new tensor of the list (the reference's memory planner aliases edges whose lifetimes do not overlap onto a few buffers:
 * one pointer is many tensors over a pass), a read refers to the newest tensor written there; an address that is read before
 * the pass wrote it is an INPUT of the pass and becomes a tensor bound to that address (saber_hip_net_bind_tensor); every
 * other tensor gets its own slot of the arena at finalize. The conv + sum post-op (RES_SUM_INPLACE) reads and writes one
 * tensor. The operator handles recorded are the caller's: they must outlive the net and keep their weights.
 * capture_end returns the list (not finalized: bind the outputs the caller wants at its own addresses with
 * saber_hip_net_tensor_of_ptr + saber_hip_net_bind_tensor, then saber_hip_net_optimize / finalize / autotune / capture as
 * for a hand-built list) or SABER_HIP_UNIMPL when something in the loop cannot be expressed - an entry point without an
 * op-list form (GEMM, pair / chain / stage launches), or an access overlapping a live tensor at a different base address -
 * in which case the caller keeps running its own loop. */
int saber_hip_capture_begin(void);
int saber_hip_capture_end(saber_hip_net_t** out);
int saber_hip_capture_active(void);
/* captured lists: id of the NEWEST tensor the pass saw at `ptr` (for an output address: the tensor the last writer produced), -1: none */
int saber_hip_net_tensor_of_ptr(const saber_hip_net_t* net, const void* ptr);

/* ---- serving: streams that do not share a hardware queue --------------------------------------------------------------------
 * Replaces, for a server that keeps several Nets in flight on one device, the streams Env<T> hands every Context<T>
 * (saber/core/context.h:38-77, saber/core/env.h; one Net per Worker pool thread: framework/core/worker.h). The HIP runtime serves all
 * streams of a process from GPU_MAX_HW_QUEUES (default 4) hardware queues assigned by creation order; two Nets whose streams share a
 * queue run one after the other. Fills out[0 .. n) with non-blocking streams of the CURRENT device taken round-robin from a per-device
 * set of up to FOUR streams no two of which share a queue (found once per device by overlapping spin kernels, ~3 ms, on an otherwise
 * idle device: call it before the serving threads start); *distinct (may be null) = the size of that set. Streams with the same index
 * modulo *distinct are the same stream: passes enqueued on it run in order. The library owns the streams (never destroy them). */
int saber_hip_serving_streams(int n, saber_hip_stream_t* out, int* distinct);

#ifdef __cplusplus
}
#endif
#endif /* SABER_HIP_H */
