#!/usr/bin/env python
"""bench.py — ResNet50 INT8 inference throughput on MI355X through the C-ABI HIP Saber target.

A "step" is ONE forward pass of the post-fusion ResNet50 INT8 op list (53 conv-like ops, 2 pools,
16 residual adds fused bit-exactly into the branch2c epilogues, global pool, FC, softmax) over one
batch of synthetic images already resident in HBM. Protocol mirrors the reference's (README.md:44,
benchmark/CNN/run.sh): warm-up, then K timed iterations; images/s = batch * K / time.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--batch 8]

N > 1: launched by torch.distributed.run, one rank per GPU; the batch is sharded as independent
per-GPU sub-batches (weak scaling, no data-path collective) and the per-rank logits are
all-gathered over RCCL each step (the only exchange the path has).
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


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=8, help="images per GPU per step")
    ap.add_argument("--model", default="resnet50", choices=["resnet50", "resnet101", "vgg16"])
    ap.add_argument("--precision", default="int8", choices=["int8", "fp32"])
    ap.add_argument("--graph", default="framework", choices=["framework", "caffe"],
                    help="INT8: 'framework' = the op list the reference's own optimiser + edge rules emit (workloads.framework_spec: "
                         "stride-up, conv1 -> s8, INT8 tail; what Net<MI355X> runs), 'caffe' = the round-1/2 list (plain Caffe topology)")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--no-fuse", action="store_true", help="keep the 16 eltwise ops separate (reference op list)")
    ap.add_argument("--no-autotune", action="store_true")
    ap.add_argument("--tune-cache", default=None,
                    help="JSON file with the autotuned kernel selection per (batch, op): written after autotuning when absent, "
                         "applied instead of autotuning when present (profiling passes then all run the same kernels)")
    ap.add_argument("--lanes", action="store_true",
                    help="run the shortcut projections on a side stream (measured SLOWER under hipGraph: 0.464 vs 0.371 ms)")
    ap.add_argument("--chain", type=int, default=None,
                    help="INT8 ResNet: 0 no conv1x1 chains, 1 chains, 2 (default) chains that may start with the block's 3x3 conv")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-b1", action="store_true", help="skip the batch-1 latency leg")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--per-op", action="store_true", help="print the per-op time table to stderr")
    ap.add_argument("--timed-only", action="store_true",
                    help="profiling aid: only warm-up + timed region (no latency / per-op / cpu legs), so the last "
                         "dispatches of a rocprofv3 trace are exactly one forward pass")
    return ap.parse_args()


def build_net(W, model, scales, batch, args):
    if args.precision == "int8":
        return W.build_int8_net(model, dict(scales), batch, fuse_eltwise=not args.no_fuse, lanes=args.lanes, chain=args.chain)
    return W.build_fp32_net(model, batch)


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
    """hipGraph replay against eager launches (the C++ op loop of saber_hip_net_run) of the same op list, timed once
    before the timed region — with the per-step logits gather when there is one, since the eager loop costs ~275 us of
    host time per step (graph: ~30 us) and leaves little room for anything else on the launching thread; the faster
    one is used. On this host, single GPU, the eager loop keeps the GPU fed and is ~1 % faster than the graph; and
    occasionally (2 of ~30 runs) a process gets a graph whose replay is 25-30 % slower than the sum of its kernels
    while eager per-op times are normal."""
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


def main():
    args = parse()
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

    # all launches go to one non-default stream (graph capture/replay, event timing)
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)

    B = args.batch
    model = W.build_model(args.model)
    if args.precision == "int8" and args.graph == "framework":
        model = W.framework_model(model, "int8")
    x = W.make_input(B, seed=1234 + rank)
    scales = W.calibrate(model, W.make_input(2)) if args.precision == "int8" else {}
    net = build_net(W, model, scales, B, args)
    net.tensor("data").copy_(torch.from_numpy(x).cuda())
    net.run()
    torch.cuda.synchronize()
    cache_key = "%s_%s_b%d" % (args.model, args.precision, B)
    cache = {}
    if args.tune_cache and os.path.exists(args.tune_cache):
        cache = json.load(open(args.tune_cache))
    if cache.get(cache_key) and len(cache[cache_key]) == net.num_ops():
        net.set_choices(cache[cache_key])
    elif not args.no_autotune:
        net.autotune(iters=20)   # RUNTIME strategy (BaseFunc::pick_best_runtime), once, outside the timed region
        net.tensor("data").copy_(torch.from_numpy(x).cuda())
        if args.tune_cache and rank == 0:
            cache[cache_key] = net.choices()
            json.dump(cache, open(args.tune_cache, "w"))
    use_graph = not args.no_graph
    launch_probe = None
    if use_graph:
        net.capture()

    logits = net.tensor("prob")
    gather = gather_flush = None
    if world > 1:
        # the path's only exchange: the per-rank logits over RCCL, double-buffered and asynchronous (step i's
        # all-gather runs on RCCL's stream while step i+1 computes; every gather completes inside the timed region)
        ag = shard.AsyncLogitGather(logits, world)

        def gather():
            ag.step(logits)
        gather_flush = ag.flush

    if use_graph:
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
    dt = shard.max_over_ranks(dt, "cuda")
    ms_per_step = dt * 1000.0 / args.steps
    value = n_gpus * B * args.steps / dt

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
        lat = []
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(100)]
        for a, b in ev:
            a.record()
            net.replay() if use_graph else net.run()
            b.record()
        torch.cuda.synchronize()
        lat = sorted(a.elapsed_time(b) for a, b in ev)
        p50, p99 = lat[len(lat) // 2], lat[min(len(lat) - 1, int(len(lat) * 0.99))]

        # ---------------- roofline of the dominant kernel family (implicit-GEMM conv) ---------------
        op_us = net.time_ops(iters=20)          # hipEvents on the launch stream, per op, eager back-to-back
        names = [net.op_name(i) for i in range(net.num_ops())]
        conv_us = sum(t for t, n in zip(op_us, names) if n.startswith("conv:") or n.startswith("fc:"))
        n_conv = sum(1 for n in names if (n.startswith("conv:") or n.startswith("fc:")) and "(in the chain launch)" not in n)
        es = 1 if args.precision == "int8" else 4
        alg_bytes = W.algorithmic_bytes_int8(model, B) * es
        alg_ops = 2.0 * W.conv_macs(model["spec"]) * B
        # Average launch duration of the conv/fc kernels IN THE TIMED REGION: the step time of the timed region (the number the
        # driver can check against its own clock; it contains every kernel boundary) apportioned to the conv/fc launches by
        # their share of the per-op hipEvent times. The isolated per-op sum itself is kept beside it: it re-runs one op 20
        # times back to back (operands warm in L2 / Infinity Cache, short kernels bounded by the host's launch rate), so it
        # is not the in-pipeline duration. A rocprofv3 kernel trace of the same command sits in profiles/ (traced runs are
        # ~10 % slower: profiles/README.md has the reconciliation).
        conv_share = conv_us / max(sum(op_us), 1e-9)
        conv_region_us = ms_per_step * 1e3 * conv_share
        achieved_gbs = alg_bytes / (conv_region_us * 1e-6) / 1e9
        achieved_tops = alg_ops / (conv_region_us * 1e-6) / 1e12
        peak_ops = MFMA_I8_PEAK_TOPS if args.precision == "int8" else MFMA_F32_PEAK_TFLOPS
        t_hbm, t_mfma = alg_bytes / (HBM_PEAK_GBS * 1e9), alg_ops / (peak_ops * 1e12)
        bound = "hbm" if t_hbm >= t_mfma else "mfma"
        if bound == "hbm":
            roof = dict(bound="hbm", achieved=round(achieved_gbs, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=round(achieved_gbs / HBM_PEAK_GBS, 4), traffic=None)
        else:
            roof = dict(bound="mfma", achieved=round(achieved_tops, 2), peak=peak_ops, unit="TFLOP/s",
                        frac=round(achieved_tops / peak_ops, 4), traffic=None)
        # HBM traffic from the committed PMC passes (profiles/r02/traffic.json: per launch, FETCH_SIZE doubled per the gfx950
        # correction) - only while the sources it was measured on are unchanged (src_sha) and the workload is the same;
        # otherwise null: a stale counter is worse than none.
        try:
            tr = json.load(open(os.path.join(ROOT, "profiles", "r02", "traffic.json")))
            if tr.get("batch") == B and args.precision == "int8" and args.model == "resnet50" and tr.get("src_sha") == L.source_sha():
                roof["traffic"] = tr["hbm_bytes_per_launch"]
                roof["traffic_unit"] = "bytes per launch (algorithmic_bytes_per_launch is the figure to compare with)"
                roof["traffic_src_sha"] = tr["src_sha"]
        except (OSError, ValueError, KeyError):
            pass
        roof.update(kernel="conv_igemm_kernel / conv_igemm_dma_kernel / conv1x1_chain_kernel / conv3x3_img_kernel / conv_stem_pool_kernel (all %d conv/fc launches of one forward)" % n_conv,
                    launches=n_conv, avg_launch_us=round(conv_region_us / n_conv, 3),
                    avg_launch_us_how="ms_per_step of the timed region x (conv/fc share of the per-op hipEvent times) / launches",
                    per_op_event_sum_us=round(conv_us, 1), conv_share_of_step=round(conv_share, 4),
                    algorithmic_bytes_per_launch=int(alg_bytes / n_conv),
                    algorithmic_bytes_per_forward=int(alg_bytes), algorithmic_ops_per_forward=int(alg_ops),
                    mfma_frac=round(achieved_tops / peak_ops, 4), hbm_frac=round(achieved_gbs / HBM_PEAK_GBS, 4),
                    sum_all_op_us=round(sum(op_us), 1))
        if args.per_op:
            for i, (n, t) in enumerate(zip(names, op_us)):   # execution order
                print("%3d %8.2f us  %s" % (i, t, n), file=sys.stderr)

        # ---------------- batch-1 latency leg (the metric quotes p50 @ batch 1 and 8) ---------------
        b1 = None
        if not args.no_b1 and B != 1:
            net1 = build_net(W, model, scales, 1, args)
            net1.tensor("data").copy_(torch.from_numpy(W.make_input(1)).cuda())
            net1.run()
            if not args.no_autotune:
                net1.autotune(iters=20)
            g1 = not args.no_graph
            if g1:
                net1.capture()
                g1, _ = pick_launch_mode(net1)
            timed_steps(net1, 20, g1)
            ev1 = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(200)]
            for a, b in ev1:
                a.record()
                net1.replay() if g1 else net1.run()
                b.record()
            torch.cuda.synchronize()
            l1 = sorted(a.elapsed_time(b) for a, b in ev1)
            b1 = dict(p50_ms=round(l1[len(l1) // 2], 4), mean_ms=round(statistics.mean(l1), 4),
                      images_per_s=round(1000.0 / statistics.mean(l1), 1))

        # ---------------- serving throughput: several independent batches in flight (extra, NOT `value`) ----------
        # The forward pass is a chain of 35 dependent launches that leaves most CUs idle most of the time; a server with
        # more than one request queue (the reference's Worker runs one Net per thread, framework/core/worker.h) fills them
        # with another batch. Two / three op lists with the same kernel selection, each a hipGraph on its own stream.
        multi = None
        if not args.no_b1 and world == 1 and args.precision == "int8":
            try:
                multi = {}
                extra, streams = [], []
                for i in range(2):
                    st = torch.cuda.Stream()
                    with torch.cuda.stream(st):
                        ne = build_net(W, model, scales, B, args)
                        ne.set_choices(net.choices())
                        ne.tensor("data").copy_(torch.from_numpy(W.make_input(B, seed=11 + i)).cuda())
                        ne.run()
                        ne.capture()
                    extra.append(ne)
                    streams.append(st)
                torch.cuda.synchronize()
                net.capture()      # (again: the launch-mode probe may have kept or dropped its graph)
                for k in (2, 3):
                    group = [(net, torch.cuda.current_stream())] + list(zip(extra[:k - 1], streams[:k - 1]))

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
                    multi["streams_%d" % k] = {"images_per_s": round(k * B / dt, 1), "ms_per_round": round(dt * 1e3, 4),
                                               "batches_in_flight": k, "batch": B}
                multi["note"] = "independent batch-%d forward passes in flight on separate streams; each batch's latency is ms_per_round" % B
            except Exception as e:   # noqa: BLE001 - an optional extra must never cost the headline line
                multi = {"error": "%s: %s" % (type(e).__name__, e)}

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
                               sample="ResNet50 INT8 batch 1, 224x224, unfused reference op list (53 conv + 16 eltwise + pool + "
                                      "gpool + fc), %d timed forwards after 10 warm-up, through the reference's own x86 Saber "
                                      "objects compiled unmodified into oracle/_ref (GemmX8S8S32XConv + MKL cblas_gemm_s8u8s32, "
                                      "SaberEltwise, PackedMKLInt8Gemm), MKL/OpenMP threads = %d of %d host cores; this is the "
                                      "reference's GEMM path - its JIT-VNNI path needs xbyak and is not buildable here "
                                      "(README.md:92 quotes 3.21 ms/image for it on 8 Xeon-6271 threads)" % (iters, min(8, ncpu), ncpu),
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
            else "f32 (v_mfma_f32_16x16x4_f32)",
            "data": "synthetic (seeded uniform images, He-init weights with folded BN, MAXABS scales)",
            "config": {"workload": "%s %s post-fusion op list, batch %d per GPU, 224x224" %
                                   (args.model, args.precision, B),
                       "global_batch": B * n_gpus, "ops": net.num_ops(), "launches": net.num_launches(), "hip_graph": use_graph,
                       "launch_probe": launch_probe,
                       "fused_eltwise": not args.no_fuse, "parallelism": "batch-shard x%d" % n_gpus},
            "latency_ms": {"batch": B, "p50": round(p50, 4), "p99": round(p99, 4)},
            "batch1": b1,
            "multi_stream": multi,
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
