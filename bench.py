#!/usr/bin/env python
"""bench.py — ResNet50 INT8 inference throughput on MI355X through the C-ABI HIP Saber target.

A "step" is ONE forward pass of the ResNet50 INT8 op list over one batch of synthetic images already resident in HBM.
The list is the one the REFERENCE'S OWN optimiser and edge rules emit for the Caffe ResNet50 graph
(anakin_amd.workloads.framework_spec, proved equal to Graph::Optimize + Net<MI355X>::init by tests/test_net_oplist.py:
76 operators = 53 conv-like + 16 eltwise + 5 pooling + fc + softmax), re-fused by this executor into fewer launches with
bit-identical results on every surviving edge (fused eltwise epilogues, sibling pairs, conv+pooling, conv1x1 chains, the
stride-up shortcut poolings folded into a residual read). The unfused 76-op list is timed beside it
(`reference_op_list`). Protocol mirrors the reference's (README.md:44, benchmark/CNN/run.sh): warm-up, then K timed
iterations; images/s = batch * K / time.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--batch 8]

N > 1: one rank per GPU under torch.distributed.run — started by the driver, or by this script itself when it is called
plainly with --gpus N (it re-executes under torch.distributed.run on 127.0.0.1). The batch is sharded as independent
per-GPU sub-batches (weak scaling, no data-path collective) and the per-rank logits are all-gathered over RCCL each step
(the only exchange the path has).
"""
import argparse
import json
import os
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_I8_PEAK_TOPS = 5000.0   # dense i8 = 2x bf16 dense (~2.5 PF), MI355X_MICROARCH.md MFMA table
MFMA_F32_PEAK_TFLOPS = 157.3
MFMA_BF16_PEAK_TFLOPS = 2500.0   # dense bf16 (no sparsity)
# FP32 convolutions on three bf16 planes per operand issue SIX bf16 products per f32 product: the MFMA roof of those kernels in
# f32-equivalent FLOP/s is the dense bf16 peak / 6
MFMA_BF16X3_PEAK_TFLOPS = MFMA_BF16_PEAK_TFLOPS / 6.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=8, help="images per GPU per step")
    ap.add_argument("--model", default="resnet50", choices=["resnet50", "resnet101", "vgg16"])
    ap.add_argument("--precision", default="int8", choices=["int8", "fp32"])
    ap.add_argument("--model-file", default=None,
                    help="an `.anakin.bin` (the reference's model format: protobuf wire format, read by anakin_amd/anakin_bin.py) holding the --model "
                         "network as original operators with raw BatchNorm / Scale blobs; replaces the seeded synthetic weights (the default, and what "
                         "the driver times: there is no model file on the GPU box). `--write-model-file PATH` writes the synthetic model out first.")
    ap.add_argument("--write-model-file", default=None, help="write the synthetic --model network as an `.anakin.bin` to this path, then use it as --model-file")
    ap.add_argument("--graph", default="framework", choices=["framework", "caffe"],
                    help="INT8: 'framework' = the op list the reference's own optimiser + edge rules emit (workloads.framework_spec: "
                         "stride-up, conv1 -> s8, INT8 tail; what Net<MI355X> runs), 'caffe' = the round-1/2 list (plain Caffe topology)")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--force-graph", action="store_true", help="hipGraph replay without timing it against eager launches (A/B aid)")
    ap.add_argument("--no-fuse", action="store_true", help="keep the 16 eltwise ops separate (reference op list)")
    ap.add_argument("--no-autotune", action="store_true")
    ap.add_argument("--tune-cache", default=os.path.join(ROOT, "profiles", "tune.json"),
                    help="JSON file with the autotuned kernel selection per (config, source hash): applied instead of autotuning when it "
                         "holds an entry for this exact configuration AND these exact sources (profiles/tune.json is committed: the "
                         "driver's run and every profiling pass then launch the same kernels, whichever box they land on); on a "
                         "mismatch the net is autotuned as before and - with --write-tune-cache - the entry is (re)written")
    ap.add_argument("--write-tune-cache", action="store_true", help="store the autotuned selection in --tune-cache")
    ap.add_argument("--retune", action="store_true", help="ignore --tune-cache entries: autotune on this box")
    ap.add_argument("--chain", type=int, default=None,
                    help="INT8 ResNet: 0 no conv1x1 chains, 1 chains, 2 (default) chains that may start with the block's 3x3 conv")
    ap.add_argument("--gather-every", type=int, default=16,
                    help="N > 1: all-gather the per-step logits once per this many steps (each step's logits are kept in a ring)")
    ap.add_argument("--compact-arena", type=int, default=0,
                    help="1: after kernel selection the net's arena is re-laid out with lifetime aliasing (saber_hip_net_compact_arena: what the "
                         "reference's MemoryScheduler does for a Net's edges; ResNet50 INT8 batch 8: 72.3 -> 24.5 MB); 0 (default): every edge keeps "
                         "its own slot - measured 3 % FASTER single-stream (0.2152 vs 0.2216 ms, profiles/r06/compact_ab.txt); the multi-stream "
                         "leg reports both forms")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-b1", action="store_true", help="skip the batch-1 latency leg")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--per-op", action="store_true", help="print the per-op time table to stderr")
    ap.add_argument("--no-stem-pair", action="store_true",
                    help="INT8: conv1 + pool1 and the sibling pair reading pool1 stay two launches (saber_hip_net_optimize flag 512 off)")
    ap.add_argument("--head-pair", action="store_true",
                    help="INT8: res2c's strided-head chain launch also runs the res3a sibling pair (saber_hip_net_optimize flag 1024; measured no faster)")
    ap.add_argument("--no-fc-softmax", action="store_true",
                    help="INT8: the fc and the Softmax over its output stay two launches (saber_hip_net_optimize flag 4096 off)")
    ap.add_argument("--no-stage", action="store_true",
                    help="INT8: do not let runs of res4 block chains run as one persistent stage launch (saber_hip_net_optimize flag 256)")
    ap.add_argument("--timed-only", action="store_true",
                    help="profiling aid: only warm-up + timed region (no latency / per-op / cpu legs), so the last "
                         "dispatches of a rocprofv3 trace are exactly one forward pass")
    return ap.parse_args()


def build_net(W, model, scales, batch, args, stage=True, shared_device=False):
    """shared_device: the net will run BESIDE other nets on this GPU (the multi-stream leg, ranks sharing a device): declared to the
    executor (saber_hip_net_optimize flag SABER_HIP_NET_SHARED_DEVICE), which then never selects a placement-dependent kernel variant"""
    if args.precision == "int8":
        # the list is handed over one op per reference operator; the C++ host side finds the fusions (saber_hip_net_optimize: the only
        # executor-level fuser of the product - the north star's "host side stays C++"); --no-fuse: the reference list as it is
        return W.build_int8_net(model, dict(scales), batch, fuse=not args.no_fuse, chain=args.chain,
                                stage=stage and not args.no_stage and not shared_device, stem_pair=not args.no_stem_pair, head_pair=args.head_pair,
                                shared_device=shared_device, fc_softmax=not args.no_fc_softmax)
    return W.build_fp32_net(model, batch, shared_device=shared_device)


def tune_key(args, batch, L):
    """a cached selection is only valid for the sources and executor options it was tuned on"""
    return "%s_%s_%s_b%d_fuse%d_lanes%d_chain%s_py%d_stage%d_sp%d_hp%d_fs%d_%s" % (args.model, args.precision, args.graph, batch, int(not args.no_fuse),
                                                                       0, args.chain, 0, int(not args.no_stage),
                                                                       int(not args.no_stem_pair), int(args.head_pair),
                                                                       int(not getattr(args, "no_fc_softmax", False)), L.source_sha())


def tune(net, args, batch, L, rank, iters=20, refill=None):
    """the committed selection when it matches this configuration and these sources, the RUNTIME strategy (autotune) otherwise;
    returns how the selection came about ("cache" | "autotune" | "static")"""
    key = tune_key(args, batch, L)
    cache = {}
    if args.tune_cache and os.path.exists(args.tune_cache):
        try:
            cache = json.load(open(args.tune_cache))
        except ValueError:
            cache = {}
    if not args.retune and cache.get(key) and len(cache[key]) == net.num_ops():
        net.set_choices(cache[key])
        return "cache"
    if args.no_autotune:
        return "static"
    net.autotune(iters=iters)   # RUNTIME strategy (BaseFunc::pick_best_runtime), once, outside the timed region
    if refill is not None:
        refill()
    if args.tune_cache and args.write_tune_cache and rank == 0:
        cache = {k: v for k, v in cache.items() if k.endswith("_" + L.source_sha())}      # entries of other sources are dead
        cache[key] = net.choices()
        json.dump(cache, open(args.tune_cache, "w"), indent=0)
    return "autotune"


def timed_steps(net, steps, use_graph, gather=None, flush=None):
    import torch
    for _ in range(steps):
        if use_graph:
            net.replay()
        else:
            net.run()
        if gather is not None:
            gather()
    if flush is not None:
        flush()
    torch.cuda.synchronize()


def pick_launch_mode(net, steps=60, gather=None, flush=None):
    """hipGraph replay against eager launches (the C++ op loop of saber_hip_net_run) of the same op list, timed once before the
    timed region (with the per-step logits gather when there is one); the faster one is used. Both forms are GPU-bound on this
    workload: the eager loop enqueues a batch-8 pass in ~60 us of host time (scripts/enqueue_cost.py; 24 launches x ~2.5 us) against
    ~216 us of GPU time, so the launching thread stays ahead as long as it has a core to itself, and a graph has no launch cost to
    save; what differs is what the command processor does between two dependent kernels (profiles/r05/graph_vs_eager.txt)."""
    import torch
    t = {}
    for mode in ("graph", "eager", "graph", "eager"):
        timed_steps(net, 5, mode == "graph", gather, flush)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        timed_steps(net, steps, mode == "graph", gather, flush)
        t[mode] = min(t.get(mode, 1e9), (time.perf_counter() - t0) * 1e3 / steps)
    use_graph = t["graph"] <= t["eager"]
    return use_graph, {"graph_ms": round(t["graph"], 4), "eager_ms": round(t["eager"], 4)}


def respawn_under_torchrun(args):
    """`python bench.py --gpus N` called plainly: become N ranks (torch.distributed.run, one node, 127.0.0.1)."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.execv(sys.executable, cmd)


def cpu_baseline_fp32(args, model):
    """BASELINE.json configs[0] ("ResNet50 FP32 batch=1 via Net<X86,Precision::FP32> on host CPU") and its VGG16 twin: the FP32 op
    list, batch 1, through the REFERENCE'S OWN x86 objects compiled into oracle/_ref (oracle/net_oracle.RefNetF32), each convolution
    on the implementation saber/funcs/impl/x86/saber_conv.cpp:49-136 selects for it when the xbyak JIT kernels are absent
    (SaberConvWinograd / SaberConv1X1 / SaberIm2colConv), 8 threads, warm-up 10 (README.md:85-86); the same list with im2col forced
    wherever the rule says Winograd is timed beside it. CHECKER / BASELINE code: never part of a timed GPU region."""
    try:
        from oracle import net_oracle as NO
        from oracle import oracle as ORC
        from anakin_amd import workloads as W
        if not ORC.ref_available():
            return {"error": "oracle/_ref not present", "value": None, "unit": "images/s", "cores": 0, "kind": "none",
                    "sample": "the FP32 x86 baseline needs the compiled reference objects (oracle/Makefile ref)"}
        ncpu = os.cpu_count() or 1
        cores = min(8, ncpu)
        xs = W.make_input(1)
        res = {}
        for tag, impl in (("dispatcher_rule", 0), ("im2col_forced", 1)):
            rn = NO.RefNetF32(model, 1, impl=impl)
            NO.ref_set_threads(cores)
            rn.run(xs)
            one = rn.time_ms(1, 2)
            budget_ms = args.cpu_seconds * 1000.0 * (0.6 if impl == 0 else 0.3)
            iters = max(3, min(200, int(budget_ms / max(one, 1e-3)) - 10))
            ms = rn.time_ms(min(10, max(2, iters // 3)), iters)
            res[tag] = dict(ms_per_image=round(ms, 3), images_per_s=round(1000.0 / ms, 3), timed_forwards=iters, conv_impls=rn.impl_counts())
            del rn
        d = res["dispatcher_rule"]
        published = {"resnet50": "README.md:92 quotes 20.62 ms/image for ResNet50 FP32 batch 1 on 8 threads of a Xeon E5-2650 v4 through the "
                                 "JIT AVX2 kernels", "vgg16": "README.md:96 quotes 55.61 ms/image for VGG16 FP32 batch 1 on 8 threads of a Xeon E5-2650 v4"}
        return dict(value=d["images_per_s"], unit="images/s", cores=cores, kind="reference", ms_per_image=d["ms_per_image"],
                    sample="%s FP32 batch 1, 224x224 (BASELINE.json configs[0] for resnet50), %d timed forwards after warm-up, through the "
                           "reference's own x86 Saber objects compiled unmodified into oracle/_ref: per convolution the implementation "
                           "SaberConv2D<X86,AK_FLOAT>::init (saber_conv.cpp:49-136) selects with the xbyak JIT kernels absent - %s; ConvEltwise = "
                           "SaberConv1X1 with beta = 1; fc = Gemm<X86,VENDER_IMPL,float> + bias; pooling restated; MKL/OpenMP threads = %d of %d host "
                           "cores. %s - not buildable here (xbyak)" % (
                               args.model, d["timed_forwards"], ", ".join("%d x %s" % (v, k) for k, v in sorted(d["conv_impls"].items())),
                               cores, ncpu, published.get(args.model, "")),
                    im2col_forced=res["im2col_forced"], dispatcher_rule=d)
    except Exception as e:   # noqa: BLE001 - a broken checker must not cost the measured line; it is reported instead
        return {"error": "%s: %s" % (type(e).__name__, e), "value": None, "unit": "images/s", "cores": 0, "kind": "none",
                "sample": "cpu baseline failed to run on this host"}


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        respawn_under_torchrun(args)          # does not return
    import torch
    import torch.distributed as dist
    from anakin_amd import lib as L
    from anakin_amd import shard
    from anakin_amd import workloads as W

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # BENCH_DIST_BACKEND=gloo + BENCH_SHARE_GPU=1: dry run of the N > 1 control flow on a single GPU (every rank on
        # cuda:0, logits exchanged through gloo); the real runs use RCCL with one GPU per rank
        backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
        dev_index = 0 if os.environ.get("BENCH_SHARE_GPU") == "1" else local_rank
        torch.cuda.set_device(dev_index)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)
    else:
        torch.cuda.set_device(0)
    L.require_device()   # no fallback: the HIP library and a gfx950 device are mandatory
    n_gpus = world
    if n_gpus != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE is %d: launch with torch.distributed.run --nproc-per-node %d "
                         "(or call `python bench.py --gpus %d` plainly and let it spawn the ranks)" % (args.gpus, world, args.gpus, args.gpus))

    # all launches go to one non-default stream (graph capture/replay, event timing)
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)

    B = args.batch
    model = W.build_model(args.model)
    if args.write_model_file:
        from anakin_amd import anakin_bin
        if rank == 0:
            anakin_bin.write_model(model, args.write_model_file, batch=B)
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        args.model_file = args.write_model_file
    if args.model_file:
        from anakin_amd import anakin_bin
        model = dict(anakin_bin.load_model(args.model_file), name=args.model)
    if args.precision == "int8" and args.graph == "framework":
        model = W.framework_model(model, "int8")
    x = W.make_input(B, seed=1234 + rank)
    scales = W.calibrate(model, W.make_input(2)) if args.precision == "int8" else {}
    # (ranks sharing one GPU - the 2-rank test of tests/test_gpu_dist.py - run concurrently on it: no persistent stage launches there,
    # two of them in flight can starve each other of CUs; one rank per GPU: the stage has its GPU to itself)
    shared = os.environ.get("BENCH_SHARE_GPU") == "1"
    net = build_net(W, model, scales, B, args, stage=not shared, shared_device=shared)
    net.tensor("data").copy_(torch.from_numpy(x).cuda())
    net.run()
    torch.cuda.synchronize()
    selection = tune(net, args, B, L, rank, refill=lambda: net.tensor("data").copy_(torch.from_numpy(x).cuda()))
    arena_full = net.arena_bytes()
    if args.compact_arena:
        net.compact()
        net.run()
        torch.cuda.synchronize()
    use_graph = not args.no_graph
    launch_probe = None
    if use_graph:
        net.capture()

    logits = net.tensor("prob")
    gather = gather_flush = None
    gather_host_us = None
    if world > 1 and os.environ.get("BENCH_NO_GATHER") != "1":   # (BENCH_NO_GATHER=1: diagnostic, isolates what the exchange costs)
        # the path's only exchange: the per-rank logits over RCCL. Every step's logits go into a device ring (one small async
        # copy per step); once per --gather-every steps the ring is all-gathered in one asynchronous collective that overlaps
        # the following steps; every gather completes inside the timed region (gather_flush = finish)
        ag = shard.BatchedLogitGather(logits, world, every=args.gather_every)

        def gather():
            ag.step(logits)
        gather_flush = ag.finish
        t_g = time.perf_counter()
        for _ in range(50):
            gather()
        gather_host_us = (time.perf_counter() - t_g) * 1e6 / 50     # host time of one asynchronous gather call
        gather_flush()
        torch.cuda.synchronize()

    if use_graph and not args.force_graph:
        use_graph, launch_probe = pick_launch_mode(net, gather=gather, flush=gather_flush)
        if world > 1:   # every rank uses the same mode (the slowest rank sets the step time anyway)
            flag = torch.tensor([1 if use_graph else 0], device="cuda")
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            use_graph = bool(flag.item())

    # ---------------- warm-up, then the timed region (barrier + synchronize on both sides) ----------
    timed_steps(net, args.warmup, use_graph, gather, gather_flush)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    timed_steps(net, args.steps, use_graph, gather, gather_flush)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # a cooperative launch (the res4 stage, a two-workgroup chain) that could not complete - another process's kernels holding the CUs
    # its workgroups wait for - counts itself in a pinned error word: such a timed region does not count (its steps computed nothing
    # valid for ~20 ms each); the site has fallen back to its single-workgroup launches, time the region once more
    coop_fallback = False
    if hasattr(net, "status"):
        try:
            net.status()
        except L.SaberHipError as e:
            coop_fallback = True
            if world > 1:      # (one rank per GPU: nothing else runs there; re-timing on one rank only would break the ranks' barriers)
                raise RuntimeError("rank %d: a cooperative launch failed inside the timed region: %s" % (rank, e))
            if use_graph:
                net.capture()
            timed_steps(net, args.warmup, use_graph, gather, gather_flush)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            timed_steps(net, args.steps, use_graph, gather, gather_flush)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            net.status()
    dt = shard.max_over_ranks(dt, "cuda")
    ms_per_step = dt * 1000.0 / args.steps
    value = n_gpus * B * args.steps / dt

    # N > 1: the same steps with the logits exchanged PER REQUEST (one all-gather per step, --gather-every 1): what a batch-64 request
    # split 8 ways pays when its answer must be complete before the next request starts; the headline amortises the exchange over
    # --gather-every steps (a request's logits then arrive up to that many steps late). Every rank takes part; reported beside it.
    per_request = None
    if world > 1 and gather is not None and args.gather_every != 1:
        ag1 = shard.BatchedLogitGather(logits, world, every=1)
        n1 = max(20, min(args.steps, 200))
        timed_steps(net, 10, use_graph, lambda: ag1.step(logits), ag1.finish)
        dist.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        timed_steps(net, n1, use_graph, lambda: ag1.step(logits), ag1.finish)
        dist.barrier()
        torch.cuda.synchronize()
        dt1 = shard.max_over_ranks(time.perf_counter() - t1, "cuda")
        per_request = {"every_steps": 1, "steps": n1, "ms_per_step": round(dt1 * 1000.0 / n1, 4),
                       "images_per_s": round(n_gpus * B * n1 / dt1, 1)}

    out = None
    if args.timed_only:
        if rank == 0:
            print(json.dumps({"value": round(value, 1), "unit": "images/s", "ms_per_step": round(ms_per_step, 4),
                              "ops": net.num_ops(), "launches": net.num_launches(), "hip_graph": use_graph, "launch_probe": launch_probe}))
        if world > 1:
            dist.destroy_process_group()
        return
    if rank == 0:
        # ---------------- per-step latency distribution (hipEvents around each replay) -------------
        # (>= 1000 iterations with per-iteration events -> mean / p50 / p99; ~0.25 s at batch 8)
        lat = []
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(1000)]
        for a, b in ev:
            a.record()
            net.replay() if use_graph else net.run()
            b.record()
        torch.cuda.synchronize()
        lat = sorted(a.elapsed_time(b) for a, b in ev)
        p50, p99 = lat[len(lat) // 2], lat[min(len(lat) - 1, int(len(lat) * 0.99))]
        lat_mean = statistics.mean(lat)

        # ---------------- roofline: every kernel function of the pass, and the dominant one ---------------
        # Per-op durations measured LIVE, in the pipeline: one hipEvent after every launch of an eager pass on the launch
        # stream (saber_hip_net_time_pass, 30 passes); their shares are scaled to the step time of the timed region (the
        # events lengthen a pass). Algorithmic bytes / ops per launch come from the executor itself
        # (saber_hip_net_op_work: SURVEY §8d — input + output + residual activations once, weights once; 2 x MACs — summed
        # over the operators a chain / pair launch covers). The rocprofv3 kernel trace of the same command sits in profiles/.
        pass_us = net.time_pass(iters=30)
        op_us = net.time_ops(iters=20)          # for reference: each op repeated back to back (operands warm)
        names = [net.op_name(i) for i in range(net.num_ops())]
        scale_to_step = ms_per_step * 1e3 / max(sum(pass_us), 1e-9)
        peak_ops = MFMA_I8_PEAK_TOPS if args.precision == "int8" else MFMA_F32_PEAK_TFLOPS
        kern = {}
        for i, (nm, t) in enumerate(zip(names, pass_us)):
            if "(in the chain launch)" in nm or "(in the stage launch)" in nm or "(in the stem launch)" in nm or "(in the fc launch)" in nm:
                continue
            by, fl = net.op_work(i)
            k = kern.setdefault(nm, dict(kernel=nm, launches=0, us=0.0, bytes=0.0, flops=0.0))
            k["launches"] += 1
            k["us"] += t * scale_to_step
            k["bytes"] += by
            k["flops"] += fl
        per_kernel = []
        for k in kern.values():
            gbs = k["bytes"] / (k["us"] * 1e-6) / 1e9
            tops = k["flops"] / (k["us"] * 1e-6) / 1e12
            pk = MFMA_BF16X3_PEAK_TFLOPS if "bf16x3" in k["kernel"] else peak_ops     # the matrix pipe this kernel runs on
            per_kernel.append(dict(kernel=k["kernel"], launches=k["launches"], avg_us=round(k["us"] / k["launches"], 3),
                                   total_us=round(k["us"], 2), bytes_per_launch=int(k["bytes"] / k["launches"]),
                                   gops_per_launch=round(k["flops"] / k["launches"] / 1e9, 4), gbs=round(gbs, 1),
                                   hbm_frac=round(gbs / HBM_PEAK_GBS, 4), mfma_peak=round(pk, 1), mfma_frac=round(tops / pk, 4)))
        per_kernel.sort(key=lambda r: -r["total_us"])
        convs = [r for r in per_kernel if r["kernel"].startswith(("conv:", "fc:"))]
        dom = dict(convs[0])                     # the kernel function with the largest share of the step
        # ... re-timed UNDISTURBED: whole eager passes with only two events, around each of its launches in turn (the per-launch markers
        # above stretch a pass, so the scaled shares understate a long kernel among many short ones - 42.5 us for a launch the
        # rocprofv3 trace of the same command has at 54.9; this figure agrees with the trace)
        dom_ops = [i for i, nm in enumerate(names) if nm == dom["kernel"]]
        dom_us = [net.time_op_in_pass(i, iters=30) for i in dom_ops]
        dom["avg_us_pass_share"] = dom["avg_us"]
        dom["avg_us"] = round(sum(dom_us) / len(dom_us), 3)
        dom["gbs"] = round(dom["bytes_per_launch"] / (dom["avg_us"] * 1e-6) / 1e9, 1)
        dom["hbm_frac"] = round(dom["gbs"] / HBM_PEAK_GBS, 4)
        dom["mfma_frac"] = round(dom["gops_per_launch"] * 1e9 / (dom["avg_us"] * 1e-6) / 1e12 / dom["mfma_peak"], 4)
        conv_us_step = sum(r["total_us"] for r in convs)
        alg_bytes = sum(r["bytes_per_launch"] * r["launches"] for r in convs)
        alg_ops = sum(r["gops_per_launch"] * r["launches"] for r in convs) * 1e9
        n_conv = sum(r["launches"] for r in convs)
        t_hbm = dom["bytes_per_launch"] / (HBM_PEAK_GBS * 1e9)
        t_mfma = dom["gops_per_launch"] * 1e9 / (dom["mfma_peak"] * 1e12)
        if t_hbm >= t_mfma:
            roof = dict(bound="hbm", achieved=dom["gbs"], peak=HBM_PEAK_GBS, unit="GB/s", frac=dom["hbm_frac"], traffic=None)
        else:
            roof = dict(bound="mfma", achieved=round(dom["mfma_frac"] * dom["mfma_peak"], 2), peak=dom["mfma_peak"], unit="TFLOP/s",
                        frac=dom["mfma_frac"], traffic=None)
        # HBM traffic of that kernel from the committed PMC passes (profiles/r06/traffic.json: per launch, FETCH_SIZE doubled per
        # the gfx950 correction) - only while the sources it was measured on are unchanged (src_sha): a stale counter is worse than none
        try:
            tr = None
            for rd in ("r06", "r05", "r04", "r03"):      # the newest profile round whose sources still match
                tp = os.path.join(ROOT, "profiles", rd, "traffic.json")
                if os.path.exists(tp):
                    tr = json.load(open(tp))
                    if tr.get("src_sha") == L.source_sha():
                        break
            if tr.get("batch") == B and tr.get("graph") == args.graph and args.precision == "int8" and args.model == "resnet50" \
                    and tr.get("src_sha") == L.source_sha():
                roof["traffic"] = tr.get("per_kernel", {}).get(dom["kernel"])
                roof["traffic_all_conv"] = tr.get("hbm_bytes_per_forward")
                roof["traffic_src_sha"] = tr["src_sha"]
        except (OSError, ValueError, KeyError):
            pass
        all_gbs = alg_bytes / (conv_us_step * 1e-6) / 1e9
        roof.update(kernel=dom["kernel"], launches=dom["launches"], avg_launch_us=dom["avg_us"],
                    algorithmic_bytes_per_launch=dom["bytes_per_launch"], algorithmic_gops_per_launch=dom["gops_per_launch"],
                    avg_launch_us_pass_share=dom["avg_us_pass_share"],
                    how="dominant kernel function of the pass (largest share of an event-per-launch eager pass, saber_hip_net_time_pass); "
                        "avg_launch_us = hipEvents on the launch stream around each of ITS launches only, inside otherwise untimed eager "
                        "forward passes (saber_hip_net_time_op_in_pass, 30 passes per launch) - the figure the rocprofv3 kernel trace of "
                        "the same command shows; avg_launch_us_pass_share = its share of the event-per-launch pass scaled to ms_per_step "
                        "(what per_kernel lists for every function: the markers stretch the pass, long kernels read short)",
                    frac_all_conv=round(all_gbs / HBM_PEAK_GBS, 4), achieved_all_conv_gbs=round(all_gbs, 1),
                    # every conv / fc launch at its own matrix pipe's peak, over the time they take in the step
                    mfma_frac_all_conv=round(sum(r["gops_per_launch"] * r["launches"] * 1e9 / (r["mfma_peak"] * 1e12) for r in convs)
                                             / (conv_us_step * 1e-6), 4),
                    achieved_all_conv_tflops=round(alg_ops / (conv_us_step * 1e-6) / 1e12, 1),
                    conv_fc_launches=n_conv, conv_fc_us_of_step=round(conv_us_step, 1),
                    algorithmic_bytes_per_forward=int(alg_bytes), algorithmic_ops_per_forward=int(alg_ops),
                    back_to_back_op_sum_us=round(sum(op_us), 1), per_kernel=per_kernel)
        if args.per_op:
            for i, (n, t, t2) in enumerate(zip(names, pass_us, op_us)):   # execution order: in-pass (scaled) | back-to-back
                print("%3d %8.2f us %8.2f us  %s" % (i, t * scale_to_step, t2, n), file=sys.stderr)

        # ---------------- batch-1 latency leg (the metric quotes p50 @ batch 1 and 8) ---------------
        b1 = None
        if not args.no_b1 and B != 1:
            net1 = build_net(W, model, scales, 1, args)
            net1.tensor("data").copy_(torch.from_numpy(W.make_input(1)).cuda())
            net1.run()
            sel1 = tune(net1, args, 1, L, rank)
            if args.compact_arena:
                net1.compact()
                net1.tensor("data").copy_(torch.from_numpy(W.make_input(1)).cuda())
                net1.run()
            g1 = not args.no_graph
            if g1:
                net1.capture()
                g1, _ = pick_launch_mode(net1)
            timed_steps(net1, 20, g1)
            ev1 = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(1000)]
            for a, b in ev1:
                a.record()
                net1.replay() if g1 else net1.run()
                b.record()
            torch.cuda.synchronize()
            l1 = sorted(a.elapsed_time(b) for a, b in ev1)
            b1 = dict(p50_ms=round(l1[len(l1) // 2], 4), p99_ms=round(l1[min(len(l1) - 1, int(len(l1) * 0.99))], 4),
                      mean_ms=round(statistics.mean(l1), 4), samples=len(l1),
                      images_per_s=round(1000.0 / statistics.mean(l1), 1), images_per_s_p50=round(1000.0 / l1[len(l1) // 2], 1),
                      launches=net1.num_launches(), selection=sel1)

        # ---------------- serving throughput: several independent batches in flight (extra, NOT `value`) ----------
        # The forward pass is a chain of 35 dependent launches that leaves most CUs idle most of the time; a server with
        # more than one request queue (the reference's Worker runs one Net per thread, framework/core/worker.h) fills them
        # with another batch. Two / three / four op lists with the same kernel selection, each a hipGraph on its own stream.
        multi = None
        if not args.no_b1 and world == 1 and args.precision == "int8":
            try:
                multi = {}
                extra, streams = [], []
                # streams that do not share a hardware queue (saber_hip_serving_streams: the runtime serves every stream of the process from
                # four queues assigned by creation order - rounds 4 / 5 measured streams_3 BELOW streams_2 because the third stream of this
                # process happened to share the first one's queue; profiles/r06/multi_stream_curve.txt)
                from anakin_amd.streams import serving_streams
                picked, distinct_queues = serving_streams(4)
                for i in range(4):
                    st = picked[i]
                    with torch.cuda.stream(st):
                        # every net of this leg runs BESIDE the others: built with SABER_HIP_NET_SHARED_DEVICE (no persistent stage launch,
                        # no cooperating-workgroup chains, no split-K through an XCD's L2 - selected out, not found out by a time-out)
                        ne = build_net(W, model, scales, B, args, shared_device=True)
                        ne.tensor("data").copy_(torch.from_numpy(W.make_input(B, seed=11 + i)).cuda())
                        ne.run()
                        if i == 0:
                            ne.autotune(iters=7)
                            shared_choices = ne.choices()
                        else:
                            ne.set_choices(shared_choices)
                        ne.run()
                        ne.capture()
                    extra.append(ne)
                    streams.append(st)
                torch.cuda.synchronize()
                arena_full = extra[0].arena_bytes()
                # every edge in its own slot first, then the same nets with lifetime-aliased arenas (saber_hip_net_compact_arena - the
                # reference's MemoryScheduler role: three nets' working sets then fit the 256 MB Infinity Cache together)
                for form in ("every_edge", "compact"):
                    if form == "compact":
                        if os.environ.get("BENCH_NO_COMPACT"):
                            break
                        for n_, s_ in zip(extra, streams):
                            with torch.cuda.stream(s_):
                                n_.compact()
                                n_.run()
                                n_.capture()
                        torch.cuda.synchronize()
                    for k in (2, 3, 4):
                        group = list(zip(extra[:k], streams[:k]))

                        def round_():
                            for n_, s_ in group:
                                with torch.cuda.stream(s_):
                                    n_.replay()
                        for _ in range(20):
                            round_()
                        torch.cuda.synchronize()
                        t0 = time.perf_counter()
                        for _ in range(200):
                            round_()
                        torch.cuda.synchronize()
                        dt = (time.perf_counter() - t0) / 200
                        key = "streams_%d" % k if form == "every_edge" else "streams_%d_compact_arena" % k
                        multi[key] = {"images_per_s": round(k * B / dt, 1), "ms_per_round": round(dt * 1e3, 4),
                                      "batches_in_flight": k, "batch": B}
                arena_now = extra[0].arena_bytes()
                multi["coop_fallbacks"] = sum(n_.coop_fallbacks() for n_ in extra)
                multi["launches_per_net"] = extra[0].num_launches()
                multi["distinct_hardware_queues"] = distinct_queues
                multi["arena_mb_per_net"] = round(arena_now / 2**20, 1)
                multi["arena_mb_per_net_every_edge"] = round(arena_full / 2**20, 1)
                multi["note"] = "independent batch-%d forward passes in flight on separate streams; each batch's latency is ms_per_round" % B
            except Exception as e:   # noqa: BLE001 - an optional extra must never cost the headline line
                multi = {"error": "%s: %s" % (type(e).__name__, e)}

        # ---------------- the reference op list, unfused: what an unchanged Net dispatches (76 operators) ---------------
        # (a) the same C++ executor on the list with ONE op per reference operator (no executor-level fusion at all);
        # (b) when the integration binary is present: the reference's own Net<MI355X>::prediction() loop on the same network
        #     (integration/test_net_mi355x.cpp: Graph(AddOp) -> Optimize() -> Net::init -> prediction, hipEvents through
        #     SaberTimer<MI355X>) - the reference's executor is host-bound at these kernel durations (BaseFunc::operator()
        #     re-checks shapes and records an event per output edge), that time is ITS cost, not the kernels'.
        ref_list = None
        if not args.no_b1 and world == 1 and args.precision == "int8" and args.graph == "framework":
            try:
                rn = W.build_int8_net(model, dict(scales), B, fuse=False)
                rn.tensor("data").copy_(torch.from_numpy(x).cuda())
                rn.run()
                if not args.no_autotune:
                    rn.autotune(iters=10)
                rn.capture()
                g_ref, probe_ref = pick_launch_mode(rn, steps=40)
                timed_steps(rn, 10, g_ref)
                t0 = time.perf_counter()
                timed_steps(rn, 200, g_ref)
                ms_ref = (time.perf_counter() - t0) * 1e3 / 200
                ref_list = dict(ops=rn.num_ops(), launches=rn.num_launches(), ms_per_step=round(ms_ref, 4),
                                images_per_s=round(B * 1000.0 / ms_ref, 1), hip_graph=g_ref,
                                what="workloads.framework_spec one op per reference operator (the list Graph::Optimize + "
                                     "Net<MI355X>::init produce, tests/test_net_oplist.py), through saber_hip_net_run")
                del rn
                exe = os.path.join(ROOT, "integration", "_build", "test_net_mi355x.bin")
                if os.path.exists(exe) and args.model == "resnet50":
                    import subprocess
                    import tempfile
                    from integration import net_model as NM
                    with tempfile.TemporaryDirectory() as td:
                        base = W.build_model(args.model)
                        mt, wb = NM.write_model(base, dict(scales), B, td, "int8", calibrator_config=True)
                        x.tofile(os.path.join(td, "input.bin"))
                        r = subprocess.run([exe, mt, wb, os.path.join(td, "input.bin"), td, "200"], capture_output=True,
                                           text=True, timeout=300, cwd=td)      # (the reference's logger writes ./log/)
                        if r.returncode == 0:
                            tt = open(os.path.join(td, "timing.txt")).read().split()
                            pl = open(os.path.join(td, "plan.txt")).read().split("\n")[0].split()
                            ref_list["net_prediction"] = dict(
                                ms_per_step=round(float(tt[tt.index("ms_per_prediction") + 1]), 4), exec_funcs=int(tt[1]),
                                planned=bool(int(tt[tt.index("planned") + 1])),
                                plan_launches=int(pl[pl.index("launches") + 1]), plan_hip_graph=bool(int(pl[pl.index("graph") + 1])),
                                op_loop_ms_per_step=round(float(tt[tt.index("ms_per_prediction_op_loop") + 1]), 4),
                                images_per_s=round(B * 1000.0 / float(tt[tt.index("ms_per_prediction") + 1]), 1),
                                what="the reference's own Net<MI355X, INT8>::prediction() (framework/core/net/net.cpp:417-509) on the "
                                     "graph its optimiser produced, MI355X Saber target underneath: Net::init captured the operator loop "
                                     "once (saber_hip_capture_begin / _end), the executor fused + autotuned it, prediction() replays that "
                                     "plan and syncs the outputs (integration/mi355x/framework/mi355x_net_plan.h); op_loop_ms_per_step = "
                                     "the same Net with the plan switched off (94 executors, one launch per operator)")
                            # the reference's serving shape on the same model file: Worker<MI355X, INT8> (framework/core/net/
                            # worker.h:38-60), 3 pool threads = 3 Nets, each replaying its own plan on its own stream; requests
                            # and answers are HOST tensors (4.8 MB of f32 image per batch-8 request over PCIe - an inclusive rate)
                            def worker_run(mode, threads, requests=300):
                                # (stderr discarded: the reference's Worker logs ~22 INFO lines per request, see integration/test_net_mi355x.cpp)
                                rw = subprocess.run([exe, mt, wb, os.path.join(td, "input.bin"), td, mode, str(threads), str(requests)],
                                                    stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, errors="replace", timeout=600, cwd=td)
                                if rw.returncode != 0:
                                    return {"error": "rc %d" % rw.returncode}
                                wt = open(os.path.join(td, "worker.txt")).read().split()
                                f = {wt[i]: wt[i + 1] for i in range(0, len(wt) - 1, 2)}
                                return dict(threads=threads, requests=int(f["requests"]), mismatches=int(f["mismatches"]),
                                            images_per_s=round(float(f["images_per_s"]), 1), median_ms=float(f["median_ms"]),
                                            max_ms=float(f["max_ms"]), coop_fallbacks=int(f["coop_fallbacks"]))
                            # (every run warms up until ALL pool threads serve: each builds its Net inside ThreadPool::launch, one at a time,
                            # seconds each - rounds 4 / 5 timed right after the first answers = ONE serving thread whatever the pool size,
                            # profiles/r05/worker_ready.txt)
                            ref_list["worker"] = worker_run("worker", 3, 600)
                            ref_list["worker"]["what"] = (
                                "Worker<MI355X, INT8>::sync_prediction, batch-%d requests from PAGEABLE host memory (PCIe-inclusive: 4.8 MB of "
                                "f32 image per request, hipMemcpyAsync on the calling pool thread's own copy stream, mi355x_impl.cpp), "
                                "3 pool threads x (Graph::load + load_calibrator_config + Optimize + Net with its captured plan on its own stream, "
                                "SABER_HIP_NET_SHARED_DEVICE), timed once every pool thread serves; median / max = submit -> answer with at most "
                                "2 x threads requests outstanding" % B)
                            # four pool threads = the hardware's four concurrently served queues: the best Worker shape measured
                            # (50 - 52k images/s; profiles/r06/worker_serving_streams_ab.txt)
                            ref_list["worker_4_threads"] = worker_run("worker", 4, 600)
                            ref_list["worker_6_threads"] = worker_run("worker", 6, 600)
                            # the SAME per-thread Graph + Net<MI355X> + request (host tensor -> input, prediction(), output -> host tensor) from
                            # plain std::threads, without the reference's Worker / ThreadPool shell around it
                            for nt in (1, 3):
                                ref_list["net_threads_%d" % nt] = worker_run("threads", nt, 600)
                            ref_list["net_threads_3"]["what"] = (
                                "3 std::threads x (Graph::load + Optimize + Net<MI355X, INT8>): Tensor::copy_from(host) -> Net::prediction() -> "
                                "Tensor::copy_from(device), batch-%d requests from pageable host memory (PCIe-inclusive), no Worker / ThreadPool" % B)
                            ref_list["worker_pinned_requests"] = worker_run("worker_pinned", 3, 600)
                            ref_list["worker_async_prediction"] = worker_run("worker_async", 3, 600)
            except Exception as e:   # noqa: BLE001 - an optional extra must never cost the headline line
                ref_list = {"error": "%s: %s" % (type(e).__name__, e)}

        # ---------------- Gemm<MI355X, float, float> (saber_hip_gemm_f32), timed once: SURVEY §8 row a-8 ---------------
        gemm = None
        if not args.no_b1 and world == 1:
            try:
                from anakin_amd import saber as S
                gm = gn = gk = 2048
                ga = torch.randn(gm, gk, device="cuda")
                gb = torch.randn(gk, gn, device="cuda")
                gc = torch.empty(gm, gn, device="cuda")
                for _ in range(3):
                    S.gemm(False, False, gm, gn, gk, 1.0, ga, gb, 0.0, gc)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    S.gemm(False, False, gm, gn, gk, 1.0, ga, gb, 0.0, gc)
                e1.record()
                torch.cuda.synchronize()
                g_us = e0.elapsed_time(e1) * 1e3 / 20
                g_tf = 2.0 * gm * gn * gk / (g_us * 1e-6) / 1e12
                gemm = dict(m=gm, n=gn, k=gk, us=round(g_us, 2), tflops=round(g_tf, 2),
                            mfma_f32_frac=round(g_tf / MFMA_F32_PEAK_TFLOPS, 4), bf16x3_frac=round(g_tf / MFMA_BF16X3_PEAK_TFLOPS, 4),
                            what="saber_hip_gemm_f32 (Gemm<MI355X, float, float>::dispatch): f32-equivalent FLOP/s; the kernel runs on three bf16 "
                                 "planes per operand (roof = dense bf16 / 6 = %.0f TF), B re-split on the device every call (included)" % MFMA_BF16X3_PEAK_TFLOPS)
                # VGG16's fc6 as a GEMM (m = 8, k = 25088, n = 4096, weights [n, k] = trans_b): a 411 MB weight stream, HBM-bound
                fm, fk, fn = 8, 25088, 4096
                fa = torch.randn(fm, fk, device="cuda")
                fb = torch.randn(fn, fk, device="cuda")
                fc_ = torch.empty(fm, fn, device="cuda")
                for _ in range(2):
                    S.gemm(False, True, fm, fn, fk, 1.0, fa, fb, 0.0, fc_)
                e0.record()
                for _ in range(10):
                    S.gemm(False, True, fm, fn, fk, 1.0, fa, fb, 0.0, fc_)
                e1.record()
                torch.cuda.synchronize()
                f_us = e0.elapsed_time(e1) * 1e3 / 10
                wbytes = fn * fk * 4.0
                gemm["vgg16_fc6"] = dict(m=fm, n=fn, k=fk, trans_b=True, us=round(f_us, 2), weight_bytes=int(wbytes),
                                         weight_gbs=round(wbytes / (f_us * 1e-6) / 1e9, 1), hbm_frac=round(wbytes / (f_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                                         bound_us=round(wbytes / (HBM_PEAK_GBS * 1e9) * 1e6, 1))
                del fa, fb, fc_
            except Exception as e:   # noqa: BLE001
                gemm = {"error": "%s: %s" % (type(e).__name__, e)}

        # ---------------- CPU baseline on this host, bounded sample, rank 0 only -----------------------------
        # "reference": the ResNet50 INT8 op list through the REFERENCE'S OWN x86 objects compiled into oracle/_ref
        # (GemmX8S8S32XConv + MKL cblas_gemm_s8u8s32, SaberEltwise, PackedMKLInt8Gemm; oracle/net_oracle.RefNet), batch 1,
        # 8 threads = the reference README's protocol ("8 thread parallel", warm-up 10, average of N runs; README.md:85-86).
        # "port": the same list through the plain-C restatement oracle/saber_oracle.c (OpenMP), kept beside it.
        cpu = None
        if not args.no_cpu_baseline and world == 1 and args.precision == "int8":
            try:
                from oracle import net_oracle as NO
                from oracle import oracle as ORC
                ncpu = os.cpu_count() or 1
                xs = W.make_input(1)
                port = {}
                prep = NO.prepare_int8(model)           # weight quantisation is init-time work, not timed
                for cores in sorted({min(8, ncpu), min(ncpu, 32)}):
                    NO.set_threads(cores)
                    NO.run_int8(model, dict(scales), xs, prep=prep)
                    t1 = time.perf_counter()
                    n_img = 0
                    while time.perf_counter() - t1 < args.cpu_seconds * 0.2:
                        NO.run_int8(model, dict(scales), xs, prep=prep)
                        n_img += 1
                    port[cores] = round(n_img / (time.perf_counter() - t1), 3)
                if ORC.ref_available():
                    rn = NO.RefNet(model, dict(scales), 1)
                    rn.run(xs)
                    NO.ref_set_threads(min(8, ncpu))
                    one = rn.time_ms(2, 3)
                    iters = max(5, min(200, int(args.cpu_seconds * 0.5 * 1000.0 / max(one, 1e-3))))
                    ms8 = rn.time_ms(10, iters)
                    # a second point at more threads, bounded: MKL oversubscribes badly on these small GEMMs (256 threads on a
                    # 256-core host: 10 s per image), so at most 32 threads and at most ~3 s of it
                    more = min(ncpu, 32)
                    more_ips = None
                    if more > 8:
                        NO.ref_set_threads(more)
                        one = rn.time_ms(1, 1)
                        if one < 500.0:
                            more_ips = round(1000.0 / rn.time_ms(1, max(2, min(30, int(3000.0 / one)))), 3)
                        NO.ref_set_threads(min(8, ncpu))
                    cpu = dict(value=round(1000.0 / ms8, 3), unit="images/s", cores=min(8, ncpu), kind="reference",
                               ms_per_image=round(ms8, 3),
                               sample="ResNet50 INT8 batch 1, 224x224, unfused reference op list (%s), %d timed forwards "
                                      "after 10 warm-up, through the reference's own x86 Saber "
                                      "objects compiled unmodified into oracle/_ref (GemmX8S8S32XConv + MKL cblas_gemm_s8u8s32, "
                                      "SaberEltwise, PackedMKLInt8Gemm), MKL/OpenMP threads = %d of %d host cores; this is the "
                                      "reference's GEMM path - its JIT-VNNI path needs xbyak and is not buildable here "
                                      "(README.md:92 quotes 3.21 ms/image for it on 8 Xeon-6271 threads)" % (
                                   "the 76 operators the reference's optimiser emits: 53 conv + 16 eltwise + 5 pooling + fc"
                                   if args.graph == "framework" else "Caffe topology: 53 conv + 16 eltwise + pool + gpool + fc",
                                   iters, min(8, ncpu), ncpu),
                               more_threads={"cores": more, "images_per_s": more_ips},
                               port={"kind": "port", "what": "oracle/saber_oracle.c (plain-C restatement, OpenMP)",
                                     "images_per_s_by_cores": port})
                else:
                    cores = max(port)
                    cpu = dict(value=port[cores], unit="images/s", cores=cores, kind="port",
                               sample="ResNet50 INT8 forward (batch 1, 224x224, unfused reference op list) through "
                                      "oracle/saber_oracle.c, OpenMP over %d host threads (oracle/_ref not present)" % cores,
                               port={"images_per_s_by_cores": port})
            except Exception as e:   # noqa: BLE001 - a broken checker must not cost the measured line; it is reported instead
                cpu = {"error": "%s: %s" % (type(e).__name__, e), "value": None, "unit": "images/s", "cores": 0, "kind": "none",
                       "sample": "cpu baseline failed to run on this host"}
        elif not args.no_cpu_baseline and world == 1:
            cpu = cpu_baseline_fp32(args, model)

        out = {
            # BASELINE.json's metric: value = images/s at batch 8 per GPU; the p50 latencies at batch 8 and batch 1 are
            # in latency_ms / batch1
            "metric": "ResNet50 INT8 images/sec + p50 latency @ batch 1/8, %d\u00d7MI355X" % n_gpus
            if args.precision == "int8" and args.model == "resnet50" and B == 8 else
            "%s %s images/sec @ batch %d, %d\u00d7MI355X" % (args.model, args.precision, B, n_gpus),
            "value": round(value, 1), "unit": "images/s", "n_gpus": n_gpus, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": "int8" if args.precision == "int8" else "f32",
            "dtype_detail": "s8/u8 x s8 -> s32 (MFMA i8), f32 requantisation epilogue" if args.precision == "int8"
            else "f32 tensors and accumulation; products on v_mfma_f32_16x16x4_f32, or - kernels named bf16x3 - on v_mfma_f32_16x16x32_bf16 "
                 "with each operand split exactly into three bf16 planes (six products per f32 product; error vs f64 below the f32 MFMA's, "
                 "profiles/r03/bf16_split_probe.txt)",
            "data": "synthetic (seeded uniform images, He-init weights with folded BN, MAXABS scales)",
            "config": {"workload": "%s %s, batch %d per GPU, 224x224: %s" %
                                   (args.model, args.precision, B,
                                    "the op list the reference's own optimiser emits (76 operators), re-fused by the executor"
                                    if (args.precision == "int8" and args.graph == "framework") else "post-fusion op list (Caffe topology)"),
                       "graph": args.graph if args.precision == "int8" else "caffe",
                       "global_batch": B * n_gpus, "ops": net.num_ops(), "launches": net.num_launches(), "hip_graph": use_graph,
                       "arena_mb": round(net.arena_bytes() / 2**20, 1), "arena_mb_every_edge": round(arena_full / 2**20, 1),
                       "launch_probe": launch_probe,
                       "fused_eltwise": not args.no_fuse, "parallelism": "batch-shard x%d" % n_gpus,
                       "dist_backend": dist.get_backend() if world > 1 else None,
                       # ranks of an RCCL communicator that really exists (0 under the gloo dry run of the multi-rank control flow)
                       "rccl_ranks": world if (world > 1 and dist.get_backend() == "nccl") else 0,
                       "kernel_selection": selection, "coop_fallback": coop_fallback,
                       "coop_fallbacks": net.coop_fallbacks() if hasattr(net, "coop_fallbacks") else 0, "shared_device": shared, "fused_by": "saber_hip_net_optimize (C++)" if (args.precision == "int8" and not args.no_fuse) else ("none (reference list)" if args.precision == "int8" else "workloads.build_fp32_net (conv + eltwise in place, sibling pairs, conv + pooling: the reference's own FP32 graph fusions)"),
                       "gather": None if (world == 1 or gather is None) else {"every_steps": args.gather_every, "backend": dist.get_backend(),
                                                          "host_us_per_step": round(gather_host_us, 1),
                                                          "per_request": per_request,
                                                          "what": "every step's logits -> device ring (async copy); one asynchronous "
                                                                  "all-gather of the ring per every_steps steps, all inside the timed region"}},
            "parity_scope": "tests/test_gpu_baseline_configs.py::test_resnet50_int8_exactly_what_the_driver_times: the cached selection "
                            "(profiles/tune.json for this source hash), C++-fused list, stage launch + stem pair on, b1/2/4/8 - every edge "
                            "the executor materialises, every image, bit-exact against the CPU oracle, eager and hipGraph, stage on and off; "
                            "the same file: fresh autotune, Python-fused, ResNet50 INT8 b1/2/4/8, ResNet101 INT8 b8, ResNet50 FP32 b1/2/4/8 and "
                            "VGG16 FP32 b8 within 1e-4 on two criteria; "
                            "tests/test_gpu_net.py: the same through the reference's own Net<MI355X> (every edge at batch 2, every image's "
                            "output at batch 8, plan == operator loop)",
            # value_p50 = batch / p50 of the per-iteration event pairs (1000 samples): what `value` would be if the timed region were
            # long - under the driver's --steps 20 `value` rests on a ~4 ms region and moves by +-1 % between runs
            "value_p50": round(B * n_gpus * 1000.0 / p50, 1),
            "latency_ms": {"batch": B, "p50": round(p50, 4), "p99": round(p99, 4), "mean": round(lat_mean, 4), "samples": len(lat)},
            "batch1": b1,
            "reference_op_list": ref_list,
            "multi_stream": multi,
            "gemm_f32": gemm,
            "roofline": roof,
            "cpu_baseline": cpu,
        }
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
