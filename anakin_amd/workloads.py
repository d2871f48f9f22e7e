"""Synthetic post-fusion op lists for the benchmark configurations of BASELINE.json.

The reference loads a `.anakin.bin` (protobuf) and runs its graph fusion; neither the model files
nor protobuf exist here, so the SAME post-fusion structure is generated directly:
Caffe-topology ResNet50 / ResNet101 (stride 2 on the 1x1 branch2a/branch1 of the first block of
stages 3-5) and VGG16, seeded weights, BatchNorm+Scale folded into the convolutions with the float
sequence of WeightsFusion::update_weights (framework/utils/parameter_fusion.cpp:88-131), and — for
INT8 — per-edge MAXABS activation scales (saber_types.h:357-360) from one FP32 pass.

Edge dtype / layout rules for the INT8 graph follow the x86 calibrator
(framework/core/net/calibrator_parse.cpp:82-128,194-244): 8-bit edges are NHWC; conv+relu outputs are
u8, conv without relu and eltwise outputs are s8; the first conv takes the f32 NCHW image and
quantises on entry (saber_conv.cpp:223-242,308); conv+eltwise fusion is OFF for INT8 (graph.cpp:423-436)
so `build_int8_net(fuse=False)` reproduces the reference op list one to one, while the default hands that list to the
C++ executor-level fuser (saber_hip_net_optimize), whose fused epilogue (include/saber_hip.h RES_ELTWISE) is bit-identical.
"""
import numpy as np

F32, S8, U8 = 0, 1, 2


# --------------------------------------------------------------------------------------------- topology
def resnet_spec(depth=50):
    """Returns the layer list: dicts with kind in {conv, pool, eltwise, gpool, fc, softmax}."""
    blocks = {50: [3, 4, 6, 3], 101: [3, 4, 23, 3]}[depth]
    L = []
    L.append(dict(kind="conv", name="conv1", src="data", cin=3, cout=64, k=7, stride=2, pad=3, relu=True))
    L.append(dict(kind="pool", name="pool1", src="conv1", win=3, stride=2, pad=0, type=0))
    prev, cin, hw = "pool1", 64, 56
    for si, nb in enumerate(blocks):
        mid = 64 << si
        cout = mid * 4
        for bi in range(nb):
            stride = 2 if (bi == 0 and si > 0) else 1
            # Caffe naming: res4a, res4b, ... for short stages; res4a, res4b1 .. res4b22 for ResNet101's long stage
            tag = "res%d%s" % (si + 2, chr(ord("a") + bi) if nb <= 6 else ("a" if bi == 0 else "b%d" % bi))
            if bi == 0:
                L.append(dict(kind="conv", name=tag + "_branch1", src=prev, cin=cin, cout=cout, k=1, stride=stride,
                              pad=0, relu=False))
                shortcut = tag + "_branch1"
            else:
                shortcut = prev
            L.append(dict(kind="conv", name=tag + "_branch2a", src=prev, cin=cin, cout=mid, k=1, stride=stride, pad=0,
                          relu=True))
            L.append(dict(kind="conv", name=tag + "_branch2b", src=tag + "_branch2a", cin=mid, cout=mid, k=3, stride=1,
                          pad=1, relu=True))
            L.append(dict(kind="conv", name=tag + "_branch2c", src=tag + "_branch2b", cin=mid, cout=cout, k=1,
                          stride=1, pad=0, relu=False, eltwise=tag))
            L.append(dict(kind="eltwise", name=tag, a=tag + "_branch2c", b=shortcut, relu=True))
            prev, cin = tag, cout
    L.append(dict(kind="gpool", name="pool5", src=prev))
    L.append(dict(kind="fc", name="fc1000", src="pool5", cin=cin, cout=1000))
    L.append(dict(kind="softmax", name="prob", src="fc1000"))
    return L


def vgg16_spec():
    cfg = [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M"]
    L, prev, cin, i, p = [], "data", 3, 0, 0
    for v in cfg:
        if v == "M":
            p += 1
            L.append(dict(kind="pool", name="pool%d" % p, src=prev, win=2, stride=2, pad=0, type=0))
            prev = "pool%d" % p
        else:
            i += 1
            L.append(dict(kind="conv", name="conv%d" % i, src=prev, cin=cin, cout=v, k=3, stride=1, pad=1, relu=True))
            prev, cin = "conv%d" % i, v
    L.append(dict(kind="fc", name="fc6", src=prev, cin=512 * 7 * 7, cout=4096, relu=True, flatten_chw=(512, 7, 7)))
    L.append(dict(kind="fc", name="fc7", src="fc6", cin=4096, cout=4096, relu=True))
    L.append(dict(kind="fc", name="fc8", src="fc7", cin=4096, cout=1000))
    L.append(dict(kind="softmax", name="prob", src="fc8"))
    return L


def conv_macs(spec, hw=224):
    """Algorithmic MACs per image of the conv + fc layers."""
    sizes, total = {"data": hw}, 0
    for l in spec:
        if l["kind"] == "conv":
            o = (sizes[l["src"]] + 2 * l["pad"] - l["k"]) // l["stride"] + 1
            sizes[l["name"]] = o
            total += l["cin"] * l["cout"] * l["k"] ** 2 * o * o
        elif l["kind"] == "pool":
            s = sizes[l["src"]]
            rnd = np.floor if l.get("floor") else np.ceil
            sizes[l["name"]] = int(rnd((s + 2 * l["pad"] - l["win"]) / l["stride"])) + 1
        elif l["kind"] == "eltwise":
            sizes[l["name"]] = sizes[l["a"]]
        elif l["kind"] == "fc":
            total += l["cin"] * l["cout"]
    return total


# --------------------------------------------------------------------------------------------- the reference's own graph
def framework_spec(spec, precision="int8"):
    """The layer list as the REFERENCE'S optimiser and edge rules leave it (checked op for op against the reference's own
    Graph::Optimize + Net::init by tests/test_net_oplist.py and integration/test_net_mi355x.cpp):

    * graph_strategy::apply_stride_up (framework/graph/llvm/optimizer/optimize_strategy.h:41-47,196-287; applied for every
      target, graph.cpp:405): the stride of 1x1 convolutions that all read the same eltwise is pushed UP through it — the
      producing `branch2c` (1x1) hands it on to the 3x3 `branch2b`, which becomes stride 2, and the other eltwise input gets a
      1x1 / stride-s max pooling (floor mode) named `<its Split node>_pool`. Values at the kept pixels are unchanged.
    * INT8 edge dtypes (CalibratorParser::get_dtype, framework/core/net/calibrator_parse.cpp:82-128 with the node dtypes of
      AutoLayoutConfigHelper::auto_config_node_dtype, auto_layout_config.cpp:105-131): conv+relu -> u8, everything else s8 —
      EXCEPT a convolution with 1 or 3 input channels, whose output dtype follows its CONSUMER (conv1 -> pool1: s8).
    * The tail runs 8-bit: an int8 -> fp32 edge would be f32, which SaberEltwise<AK_INT8> cannot write
      (saber_eltwise.cpp:85-95), so pool5 is an INT8 average pooling (s8 -> s8, scale inherited,
      saber_pooling.cpp:571-582) and the fc reads s8 (vender_fc.cpp:254-262 -> PackedMKLInt8Gemm).
    FP32: only the stride-up applies (conv + eltwise stay separate entries here; the FP32 executor fuses them)."""
    out = [dict(l) for l in spec]
    by = {l["name"]: l for l in out}
    readers = {}
    for l in out:
        for key in ("src", "a", "b"):
            if key in l:
                readers.setdefault(l[key], []).append(l)

    def conv1x1(l, stride=None):
        return l["kind"] == "conv" and l["k"] == 1 and (stride is None or l["stride"] == stride)

    def push_up(conv):            # _stride_up: a strided 1x1 conv hands its stride to a conv-like producer
        s = conv["stride"]
        if conv["k"] != 1 or s <= 1:
            return
        src = by.get(conv["src"])
        if src is not None and src["kind"] == "conv" and len(readers[src["name"]]) == 1:
            src["stride"] *= s
            conv["stride"] = 1
            push_up(src)

    inserted = []
    for e in [l for l in out if l["kind"] == "eltwise"]:
        rd = readers.get(e["name"], [])
        if len(rd) < 2 or not all(conv1x1(r) for r in rd) or len({r["stride"] for r in rd}) != 1 or rd[0]["stride"] <= 1:
            continue
        s = rd[0]["stride"]
        for r in rd:
            r["stride"] = 1
        for key in ("a", "b"):    # _stride_up_like_concat: every input of the eltwise
            src = by[e[key]]
            if src["kind"] == "conv" and len(readers[src["name"]]) == 1:
                src["stride"] *= s
                push_up(src)
            else:                 # not a conv (a Split behind the previous block): 1x1 / stride-s max pooling
                pn = src["name"] + "_outsplit_pool"
                pool = dict(kind="pool", name=pn, src=src["name"], win=1, stride=s, pad=0, type=0, floor=True)
                inserted.append((e["name"], pool))
                e[key] = pn
    for before, pool in inserted:
        # the reference schedules it anywhere between its producer and the eltwise; here: right before the block's first conv
        blk = next(i for i, l in enumerate(out) if l["kind"] == "conv" and l.get("eltwise") == before) - 2
        out.insert(max(blk, 0), pool)
        by[pool["name"]] = pool
    if precision == "int8":
        for l in out:
            if l["kind"] == "conv":
                l["odt"] = U8 if l["relu"] else S8
                if l["cin"] in (1, 3):      # calibrator_parse.cpp:104-113: the consumer decides
                    nxt = readers.get(l["name"], [])
                    relu_conv = bool(nxt) and all(r["kind"] == "conv" and r["relu"] for r in nxt)
                    l["odt"] = U8 if relu_conv else S8
            elif l["kind"] == "gpool":
                l["int8"] = True
    return out


def framework_model(model, precision="int8"):
    """`model` with the layer list rewritten by framework_spec (weights are shared, not copied)."""
    m = dict(model)
    m["spec"] = framework_spec(model["spec"], precision)
    m["framework"] = True
    return m


# --------------------------------------------------------------------------------------------- weights
def fold_bn(w, bias, bn_scale, eps, mean, var, scale_w, scale_b):
    """WeightsFusion<float,T>::update_weights (parameter_fusion.cpp:88-131) in numpy float32, same order."""
    f = np.float32
    s = f(1.0) if bn_scale == 0 else f(1.0) / f(bn_scale)
    alpha = f(1.0) / np.sqrt(var.astype(f) * s + f(eps), dtype=f)
    beta = f(-1.0) * (mean.astype(f) * s) * alpha
    alpha = scale_w.astype(f) * alpha
    beta = beta * scale_w.astype(f)
    if scale_b is not None:
        beta = beta + scale_b.astype(f)
    w2 = (w.astype(f) * alpha.reshape(-1, 1, 1, 1)).astype(f)
    b0 = bias.astype(f) if bias is not None else np.zeros(w.shape[0], f)
    return w2, (b0 * alpha + beta).astype(f)


def build_model(name="resnet50", seed=42):
    """Seeded weights: conv ~ N(0, sqrt(2/(C*k*k))), BN gamma U(.5,1.5), beta/mean U(-.1,.1), var U(.5,1.5)."""
    spec = {"resnet50": lambda: resnet_spec(50), "resnet101": lambda: resnet_spec(101), "vgg16": vgg16_spec}[name]()
    params, raw = {}, {}          # raw: the BatchNorm / Scale blobs before folding (what a model file holds)
    for idx, l in enumerate(spec):
        rng = np.random.default_rng(seed + idx)
        if l["kind"] == "conv":
            c, k, ks = l["cin"], l["cout"], l["k"]
            w = (rng.standard_normal((k, c, ks, ks)) * np.sqrt(2.0 / (c * ks * ks))).astype(np.float32)
            if name == "vgg16":
                b = (rng.uniform(-0.1, 0.1, k)).astype(np.float32)
                params[l["name"]] = (w, b)
            else:
                gamma = rng.uniform(0.5, 1.5, k).astype(np.float32)
                beta = rng.uniform(-0.1, 0.1, k).astype(np.float32)
                mean = rng.uniform(-0.1, 0.1, k).astype(np.float32)
                var = rng.uniform(0.5, 1.5, k).astype(np.float32)
                params[l["name"]] = fold_bn(w, None, 1.0, 1e-5, mean, var, gamma, beta)
                raw[l["name"]] = dict(w=w, mean=mean, var=var, gamma=gamma, beta=beta)
        elif l["kind"] == "fc":
            w = (rng.standard_normal((l["cout"], l["cin"])) * np.sqrt(1.0 / l["cin"])).astype(np.float32)
            b = rng.uniform(-0.1, 0.1, l["cout"]).astype(np.float32)
            params[l["name"]] = (w, b)
    return dict(name=name, spec=spec, params=params, raw=raw)


def make_input(batch, seed=1234, hw=224):
    rng = np.random.default_rng(seed + batch)
    return rng.uniform(-1.0, 1.0, (batch, 3, hw, hw)).astype(np.float32)


# --------------------------------------------------------------------------------------------- calibration
def calibrate(model, x):
    """MAXABS per-edge scales from one FP32 pass (torch CPU conv: model preparation, not the hot path)."""
    import torch
    import torch.nn.functional as Fn
    t = {"data": torch.from_numpy(x)}
    scales = {"data": float(np.abs(x).max()) / 127.0}
    with torch.no_grad():
        for l in model["spec"]:
            kd = l["kind"]
            if kd == "conv":
                w, b = model["params"][l["name"]]
                y = Fn.conv2d(t[l["src"]], torch.from_numpy(w), torch.from_numpy(b), l["stride"], l["pad"])
                if l["relu"]:
                    y = torch.relu(y)
            elif kd == "pool":
                y = Fn.max_pool2d(t[l["src"]], l["win"], l["stride"], l["pad"], ceil_mode=not l.get("floor", False))
            elif kd == "eltwise":
                y = torch.relu(t[l["a"]] + t[l["b"]])
            elif kd == "gpool":
                y = t[l["src"]].mean((2, 3), keepdim=True)
            elif kd == "fc":
                w, b = model["params"][l["name"]]
                y = Fn.linear(t[l["src"]].flatten(1), torch.from_numpy(w), torch.from_numpy(b))
                if l.get("relu"):
                    y = torch.relu(y)
            elif kd == "softmax":
                y = torch.softmax(t[l["src"]], 1)
            t[l["name"]] = y
            scales[l["name"]] = max(float(y.abs().max()), 1e-6) / 127.0
    return scales


# --------------------------------------------------------------------------------------------- device nets
def _out_hw(h, k, s, p):
    return (h + 2 * p - k) // s + 1


def build_int8_net(model, scales, batch, hw=224, fuse=True, chain=2, stage=True, stem_pair=True, head_pair=False, shared_device=False,
                   fc_softmax=True, absorb_pool=True, **legacy):
    """ResNet INT8 op list on the device (see the module docstring for the dtype rules): ONE op per reference operator, exactly the list
    `model["spec"]` holds (workloads.framework_spec: what the reference's own optimiser emits) - and, with `fuse`, handed to the C++ host
    side (saber_hip_net_optimize, the product's only executor-level fuser) which finds conv + eltwise, sibling pairs, conv + pooling,
    the pool -> fc quantisation, the conv1x1 chains and their 3x3 heads itself. Bytes of every surviving edge unchanged.

    fuse=False: the reference op list one launch per operator (what bench.py times as `reference_op_list`).
    chain: 0 no conv1x1 chains, 1 = `branch2c + sum + relu` and the next block's 1x1 `branch2a` as one launch (flag 16), 2 (default) = the
    block's 3x3 `branch2b` may lead that launch as well (flag 32; its output edge then stays in LDS: Net.unwritten(name)).
    stage (with chain = 2): runs of 3x3-led C = 256 chains whose blocks feed each other (the res4 stage) may run as ONE persistent launch
    (flag 256) - for a net that has the GPU to itself.
    stem_pair: the fused conv1 + pool1 launch also runs the sibling pair that reads pool1 (flag 512); pool1's edge is then not written.
    head_pair (with chain = 2): res2c's strided-head chain launch also runs the res3a sibling pair (flag 1024; measured no faster, off).
    fc_softmax: the fc and the Softmax that reads it run as one launch (flag 4096).
    shared_device: the net runs beside other nets / streams / processes on its GPU (flag SABER_HIP_NET_SHARED_DEVICE = 2048): no stage
    launch, no cooperating-workgroup chains, no split-K through one XCD's L2 - excluded from the static selection, the autotuner and
    restored selections.
    (The Python fuser of rounds 1 - 5 lives in tests/py_fuser.py: test infrastructure.)"""
    from . import lib as L
    from . import saber as S
    legacy.pop("cxx_optimize", None)      # (rounds 3 - 5 spelling of what is now the only mode)
    chain = 2 if chain is None else chain
    if legacy:
        raise TypeError("build_int8_net: %s belong to the Python fuser, which is test infrastructure now (tests/py_fuser.py)" % sorted(legacy))
    net = S.Net()
    B = batch
    net.add_tensor("data", (B, 3, hw, hw), F32)
    shape = {"data": (hw, 3)}        # name -> (spatial, channels)
    dtype = {"data": F32}
    scales = dict(scales)
    for l in model["spec"]:
        kd, nm = l["kind"], l["name"]
        if kd == "conv":
            hin, cin = shape[l["src"]]
            ho = _out_hw(hin, l["k"], l["stride"], l["pad"])
            w, b = model["params"][nm]
            odt = l.get("odt", U8 if l["relu"] else S8)   # framework_spec: conv1's output dtype follows its consumer
            p = S.ConvParam(w, b, 1, (l["pad"],) * 2, (l["stride"],) * 2, (1, 1), l["relu"])
            shape[nm], dtype[nm] = (ho, l["cout"]), odt
            conv = S.SaberConv2D(True).init((B, cin, hin, hin), p, dtype[l["src"]], odt, scales[l["src"]], scales[nm],
                                            in_layout=L.NCHW if dtype[l["src"]] == F32 else L.NHWC)
            net.add_tensor(nm, (B, ho, ho, l["cout"]), odt)
            net.add_conv(conv, l["src"], nm)
        elif kd == "pool":
            hin, c = shape[l["src"]]
            ho = S.pool_out_dim(hin, l["pad"], l["win"], l["stride"], l.get("floor", False))
            shape[nm], dtype[nm] = (ho, c), dtype[l["src"]]
            scales[nm] = scales[l["src"]]   # SaberPooling<X86,AK_INT8>::init: output scale := input scale
            net.add_tensor(nm, (B, ho, ho, c), dtype[nm])
            net.add_pool_i8(B, hin, hin, c, ho, ho, (l["win"],) * 2, (l["stride"],) * 2, (l["pad"],) * 2, l["type"],
                            dtype[l["src"]], dtype[nm], l["src"], nm)
        elif kd == "eltwise":
            ho, c = shape[l["a"]]
            shape[nm], dtype[nm] = (ho, c), S8
            net.add_tensor(nm, (B, ho, ho, c), S8)
            coeff = 1.0 / scales[nm]
            net.add_eltwise_i8(B * ho * ho * c, scales[l["a"]], scales[l["b"]], coeff, coeff, l["relu"], l["a"], l["b"], nm)
        elif kd == "gpool" and l.get("int8"):
            # INT8 global average pooling (framework_spec): s8 NHWC -> s8 [B,1,1,c], scale inherited
            hin, c = shape[l["src"]]
            shape[nm], dtype[nm] = (1, c), dtype[l["src"]]
            scales[nm] = scales[l["src"]]
            net.add_tensor(nm, (B, 1, 1, c), dtype[nm])
            net.add_pool_i8(B, hin, hin, c, 1, 1, (hin, hin), (hin, hin), (0, 0), 1, dtype[l["src"]], dtype[nm], l["src"], nm)
        elif kd == "gpool":
            # FP32 pooling op fed an s8 NHWC edge: dequantise on entry (saber_pooling.cpp:399-402), then avg
            hin, c = shape[l["src"]]
            net.add_tensor(nm, (B, c, 1, 1), F32)
            net.add_pool_f32_from_i8(B, hin, hin, c, 1, 1, (hin, hin), (hin, hin), (0, 0), 1, dtype[l["src"]],
                                     scales[l["src"]], l["src"], nm)
            shape[nm], dtype[nm] = (1, c), F32
        elif kd == "fc":
            w, b = model["params"][nm]
            fc = S.SaberFc(True).init(B, l["cout"], l["cin"], w, b, dtype.get(l["src"], F32), scales[l["src"]])
            net.add_tensor(nm, (B, l["cout"]), F32)
            net.add_fc(fc, l["src"], nm)
        elif kd == "softmax":
            classes = next(e["cout"] for e in model["spec"] if e["name"] == l["src"])      # (the fc in front of it: 1000 for the BASELINE models)
            net.add_tensor(nm, (B, classes), F32)
            net.add_softmax(B, classes, l["src"], nm)
    net.unfused_ops = net.num_ops()
    if shared_device:
        net.optimize(2048)           # (sticks to the net: every later optimize / autotune / set_choices call honours it)
        stage = False
    net.removed = net.absorbed = net.gpooled = net.stem_paired = net.chained = net.fc_softmaxed = 0
    if fuse:
        net.removed = net.optimize(15)      # conv + eltwise, sibling pairs, conv + pooling, pool -> fc quantisation
        # a stride-up shortcut pooling (framework_spec) read only by a fused eltwise epilogue is folded into that read
        net.absorbed = net.optimize(64) if absorb_pool else 0
        # the last block's conv (+ fused eltwise) also writes the global average pooling of its output (flag 128): pool5's launch goes
        net.gpooled = net.optimize(128) if absorb_pool else 0
        net.stem_paired = net.optimize(512) if stem_pair else 0
        net.chained = net.optimize(16 | (32 if int(chain) >= 2 else 0) | (256 if int(chain) >= 2 and stage else 0) |
                                   (1024 if int(chain) >= 2 and head_pair else 0)) if chain else 0
        # the fc and the Softmax over its output as one launch (flag 4096: the last-arriving workgroup of the fc kernel normalises the rows)
        net.fc_softmaxed = net.optimize(4096) if fc_softmax else 0
    net.finalize()
    return net


def build_fp32_net(model, batch, hw=224, pair_siblings=True, fuse_pool=True, shared_device=False, reproducible=False, fuse_stem=True,
                   fc_softmax=True):
    """FP32 op list: NHWC f32 on the device, conv+eltwise fused in place as the reference's FP32 graph
    does (ConvEltwise writes onto the residual's buffer, conv_elewise_fusion_scheduler.cpp:113-132).
    shared_device: see build_int8_net (here: no split-K through one XCD's L2).
    fuse_stem: conv1 (NCHW image in) + relu + pool1 as one launch where the library has the kernel (ResNet's 7x7 / 2 stem, round 6).
    reproducible: saber_hip_net_optimize flag SABER_HIP_NET_REPRODUCIBLE_FP32 - FP32 ops keep their static kernel selection whatever
    autotune / set_choices say, so that two nets of one model answer bit-identically."""
    from . import lib as L
    from . import saber as S
    net = S.Net()
    B = batch
    net.add_tensor("data", (B, 3, hw, hw), F32)
    spec = model["spec"]
    shape, alias = {}, {}
    fused_away = set()   # pooling ops emitted as part of a SaberConv2DPooling
    done_convs = set()
    produced = []   # (op index, logical edge name the op has just produced): lets a test check every edge right after its op,
                    # before a later in-place residual sum overwrites the buffer
    # the ResNet stem: conv1 reading the NCHW image + relu + pool1 as ONE launch where the library has the fused kernel (round 6,
    # conv_stem_f32.hip: SaberConv2DPooling<AK_FLOAT> on 7x7 / 2, 3 -> 64 + 3x3 / 2 max pooling) - no NHWC copy of the image, no conv1 edge
    l0, l1 = spec[0], (spec[1] if len(spec) > 1 else None)
    if fuse_stem and l0["kind"] == "conv" and l0["src"] == "data" and l1 is not None and l1["kind"] == "pool" and l1["src"] == l0["name"] \
            and l1["type"] == 0 and l0["relu"] and sum(1 for e in spec for k in ("src", "a", "b") if e.get(k) == l0["name"]) == 1:
        w0, b0 = model["params"][l0["name"]]
        p0 = S.ConvParam(w0, b0, 1, (l0["pad"],) * 2, (l0["stride"],) * 2, (1, 1), True)
        cp0 = S.SaberConv2DPooling(int8=False).init((B, 3, hw, hw), p0, l1["type"], (l1["win"],) * 2, (l1["stride"],) * 2, (l1["pad"],) * 2,
                                                     F32, F32, floor_mode=l1.get("floor", False), in_layout=L.NCHW)
        if cp0.fused and cp0.algo().startswith("stem7x7s2_maxpool3x3s2_f32"):
            po = cp0.out_hw[0]
            net.add_tensor(l1["name"], (B, po, po, l0["cout"]), F32)
            shape[l1["name"]] = (po, l0["cout"])
            net.add_conv(cp0.conv, "data", l1["name"])
            net.keep.append(cp0)
            produced.append((net.num_ops() - 1, l1["name"]))
            fused_away.add(l1["name"])
            done_convs.add(l0["name"])
    if not done_convs:
        net.add_tensor("data_nhwc", (B, hw, hw, 4), F32)
        net.add_transpose_in(B, 3, hw, hw, 4, "data", "data_nhwc")
        shape["data_nhwc"] = (hw, 4)
        alias["data"] = "data_nhwc"

    def T(n):
        return alias.get(n, n)
    sib = {}
    consumers = {}
    for e in spec:
        for key in ("src", "a", "b"):
            if key in e:
                consumers[e[key]] = consumers.get(e[key], 0) + 1
    def mark(*names):
        for n_ in names:
            produced.append((net.num_ops() - 1, n_))
    for li, l in enumerate(spec):
        kd, nm = l["kind"], l["name"]
        if kd == "conv" and nm in done_convs:
            continue
        if kd == "conv":
            hin, cin = shape[T(l["src"])]
            ho = _out_hw(hin, l["k"], l["stride"], l["pad"])
            w, b = model["params"][nm]
            if cin != w.shape[1]:   # first layer: channels padded 3 -> 4
                w = np.concatenate([w, np.zeros((w.shape[0], cin - w.shape[1]) + w.shape[2:], np.float32)], 1)
            p = S.ConvParam(w, b, 1, (l["pad"],) * 2, (l["stride"],) * 2, (1, 1), l["relu"])
            if "eltwise" in l:
                # fused: accumulate onto the shortcut tensor, relu of the eltwise
                el = next(e for e in model["spec"] if e["kind"] == "eltwise" and e["name"] == l["eltwise"])
                p.res_mode, p.res_relu = L.RES_SUM_INPLACE, el["relu"]
                conv = S.SaberConv2D(False).init((B, cin, hin, hin), p, F32, F32, in_layout=L.NHWC, out_layout=L.NHWC)
                # identity blocks: the shortcut is the block input, still needed? no - branch2a already consumed it
                net.add_conv(conv, T(l["src"]), T(el["b"]))
                mark(el["name"])
                alias[el["name"]] = T(el["b"])
                shape[T(el["b"])] = (ho, l["cout"])
                continue
            nxt = spec[li + 1] if li + 1 < len(spec) else None
            if fuse_pool and nxt is not None and nxt["kind"] == "pool" and nxt["src"] == nm and consumers.get(nm) == 1 and \
                    nxt["type"] == 0 and l["relu"]:
                # SaberConv2DPooling: conv + relu + max pooling in ONE launch where the library has a fused kernel (VGG16's
                # 2x2 / stride-2 stages); the conv's own output edge then does not exist
                cpool = S.SaberConv2DPooling(int8=False).init((B, cin, hin, hin), p, nxt["type"], (nxt["win"],) * 2,
                                                                (nxt["stride"],) * 2, (nxt["pad"],) * 2, F32, F32)
                if cpool.fused:
                    pn, po = nxt["name"], cpool.out_hw[0]
                    net.add_tensor(pn, (B, po, po, l["cout"]), F32)
                    shape[pn] = (po, l["cout"])
                    net.add_conv(cpool.conv, T(l["src"]), pn)
                    net.keep.append(cpool)
                    mark(pn)
                    fused_away.add(pn)
                    continue
            conv = S.SaberConv2D(False).init((B, cin, hin, hin), p, F32, F32, in_layout=L.NHWC, out_layout=L.NHWC)
            net.add_tensor(nm, (B, ho, ho, l["cout"]), F32)
            shape[nm] = (ho, l["cout"])
            # sibling pair (stage-entry branch1 + branch2a share input and geometry): one launch, two outputs
            nxt = spec[li + 1] if li + 1 < len(spec) else None
            geo = ("src", "k", "stride", "pad")
            if pair_siblings and nxt is not None and nxt["kind"] == "conv" and "eltwise" not in nxt and \
                    all(nxt[g] == l[g] for g in geo) and l["cout"] % 128 == 0 and nxt["cout"] % 16 == 0 and cin % 4 == 0:
                sib[nxt["name"]] = (conv, nm)
                continue
            if nm in sib:
                first, first_nm = sib.pop(nm)
                net.add_conv_pair(S.SaberConvPair(first, conv), T(l["src"]), first_nm, nm)
                mark(first_nm, nm)
                continue
            net.add_conv(conv, T(l["src"]), nm)
            mark(nm)
        elif kd == "pool":
            if nm in fused_away:
                continue
            hin, c = shape[T(l["src"])]
            ho = S.pool_out_dim(hin, l["pad"], l["win"], l["stride"])
            net.add_tensor(nm, (B, ho, ho, c), F32)
            shape[nm] = (ho, c)
            net.add_pool_f32(B, hin, hin, c, ho, ho, (l["win"],) * 2, (l["stride"],) * 2, (l["pad"],) * 2, l["type"],
                             L.NHWC, T(l["src"]), nm)
            mark(nm)
        elif kd == "eltwise":
            pass  # fused into branch2c above
        elif kd == "gpool":
            hin, c = shape[T(l["src"])]
            net.add_tensor(nm, (B, c), F32)
            shape[nm] = (1, c)
            net.add_pool_f32(B, hin, hin, c, 1, 1, (hin, hin), (hin, hin), (0, 0), 1, L.NHWC, T(l["src"]), nm)
            mark(nm)
        elif kd == "fc":
            w, b = model["params"][nm]
            if "flatten_chw" in l:   # NCHW-flattened weights -> our NHWC flatten order
                c, h, ww = l["flatten_chw"]
                w = np.ascontiguousarray(w.reshape(-1, c, h, ww).transpose(0, 2, 3, 1).reshape(w.shape[0], -1))
            fc = S.SaberFc(False).init(B, l["cout"], l["cin"], w, b, F32)
            net.add_tensor(nm, (B, l["cout"]), F32)
            net.add_fc(fc, T(l["src"]), nm)
            if l.get("relu"):
                net.keep.append("relu-after-fc: applied by eltwise(x, x, 0.5, 0.5, relu)")
                net.add_eltwise_f32(B * l["cout"], 0.5, 0.5, True, nm, nm, nm)
            mark(nm)
        elif kd == "softmax":
            classes = next(e["cout"] for e in spec if e["name"] == l["src"])
            net.add_tensor(nm, (B, classes), F32)
            net.add_softmax(B, classes, T(l["src"]), nm)
            mark(nm)
    net.alias = alias
    net.produced = produced
    if shared_device:
        net.optimize(2048)
    if reproducible:
        net.optimize(8192)
    # the fc and the Softmax over its output as one launch where the library's small-batch FP32 fc can normalise its own rows (flag 4096)
    net.fc_softmaxed = net.optimize(4096) if fc_softmax else 0
    net.finalize()
    return net


# --------------------------------------------------------------------------------------------- algorithmic counts
def algorithmic_bytes_int8(model, batch, hw=224):
    """Compulsory HBM bytes of the conv/fc kernels of one batch (each tensor touched once; weights once
    per batch), per SURVEY.md §8(d): INT8 conv in + out (+ residual re-read) elements x 1 byte + weights."""
    sizes = {"data": (hw, 3)}
    act, wts = 0, 0
    for l in model["spec"]:
        if l["kind"] == "conv":
            hin, cin = sizes[l["src"]]
            ho = _out_hw(hin, l["k"], l["stride"], l["pad"])
            sizes[l["name"]] = (ho, l["cout"])
            act += cin * hin * hin + l["cout"] * ho * ho
            if "eltwise" in l:
                act += l["cout"] * ho * ho
            wts += cin * l["cout"] * l["k"] ** 2
        elif l["kind"] == "pool":
            hin, c = sizes[l["src"]]
            rnd = np.floor if l.get("floor") else np.ceil
            sizes[l["name"]] = (int(rnd((hin + 2 * l["pad"] - l["win"]) / l["stride"])) + 1, c)
        elif l["kind"] == "eltwise":
            sizes[l["name"]] = sizes[l["a"]]
        elif l["kind"] == "fc":
            wts += l["cin"] * l["cout"]
    return act * batch + wts
