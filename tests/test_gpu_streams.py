"""saber_hip_serving_streams (include/saber_hip.h; anakin_amd/csrc/api_streams.hip): the streams a server keeps several Nets in flight on
do not share a hardware queue - the role of the per-Context streams of the reference (saber/core/context.h:38-77; one Net per Worker pool
thread: framework/core/worker.h). Checked against the hardware, not against the library's own probe: spin kernels launched from here."""
import pytest
import torch


def _spin_ms(streams, cycles):
    torch.cuda.synchronize()
    start = torch.cuda.Event(enable_timing=True)
    ends = [torch.cuda.Event(enable_timing=True) for _ in streams]
    start.record(streams[0])
    for s in streams[1:]:
        s.wait_event(start)
    for s, e in zip(streams, ends):
        with torch.cuda.stream(s):
            torch.cuda._sleep(cycles)
            e.record(s)
    torch.cuda.synchronize()
    return max(start.elapsed_time(e) for e in ends)


@pytest.mark.gpu
def test_serving_streams_overlap_pairwise_and_all_together():
    from anakin_amd.streams import serving_streams
    streams, distinct = serving_streams(6)
    assert 1 <= distinct <= 4
    again, d2 = serving_streams(6)
    assert d2 == distinct and [s.cuda_stream for s in again] == [s.cuda_stream for s in streams]      # one set per device
    assert len({s.cuda_stream for s in streams}) == distinct                                           # round-robin over it
    for i in range(distinct, 6):
        assert streams[i].cuda_stream == streams[i - distinct].cuda_stream
    assert distinct >= 2, "the runtime gave every stream the same hardware queue (GPU_MAX_HW_QUEUES=1?)"
    # calibrate the spin to ~0.3 ms, then: any pair, and the whole set, take about ONE spin - not two / not `distinct`
    cyc = 100_000
    ms = min(_spin_ms([streams[0]], cyc) for _ in range(3))
    cyc = int(cyc * 0.3 / ms)
    solo = min(_spin_ms([streams[0]], cyc) for _ in range(5))
    for i in range(distinct):
        for j in range(i + 1, distinct):
            both = min(_spin_ms([streams[i], streams[j]], cyc) for _ in range(3))
            assert both < 1.5 * solo, (i, j, solo, both)
    every = min(_spin_ms(streams[:distinct], cyc) for _ in range(3))
    assert every < 1.6 * solo, (distinct, solo, every)
    # the same stream twice does serialise: what the probe tells apart
    twice = min(_spin_ms([streams[0], streams[distinct]], cyc) for _ in range(3))
    assert twice > 1.6 * solo, (solo, twice)


@pytest.mark.gpu
def test_nets_in_flight_on_serving_streams_answer_with_the_oracles_bits():
    """four shared-device ResNet50 INT8 batch-2 nets, each with its own images, replayed together on serving streams 30 times: every net's
    logits are the oracle's bits for ITS images (passes in flight beside each other do not touch each other's tensors)"""
    import numpy as np
    from anakin_amd import lib as L
    from anakin_amd import workloads as W
    from anakin_amd.streams import serving_streams
    from oracle import net_oracle as NO
    L.require_device()
    model = W.framework_model(W.build_model("resnet50"), "int8")
    scales = W.calibrate(model, W.make_input(2))
    streams, distinct = serving_streams(4)
    nets, xs = [], []
    for i, st in enumerate(streams):
        x = W.make_input(2, seed=31 + i)
        with torch.cuda.stream(st):
            n = W.build_int8_net(model, dict(scales), 2, shared_device=True)
            n.tensor("data").copy_(torch.from_numpy(x).cuda())
            n.run()
            n.capture()
        nets.append(n)
        xs.append(x)
    torch.cuda.synchronize()
    for _ in range(30):
        for n, st in zip(nets, streams):
            with torch.cuda.stream(st):
                n.replay()
    torch.cuda.synchronize()
    for i, (n, x) in enumerate(zip(nets, xs)):
        ref = NO.run_int8(model, dict(scales), x)
        checked = 0
        for nm in n.tensors:
            if nm == "data" or nm not in ref or n.unwritten(nm) or nm == "prob":
                continue
            got = n.tensor(nm).cpu().numpy()
            assert np.array_equal(got, ref[nm].reshape(got.shape)), (i, nm)
            checked += 1
        assert checked > 20 and n.coop_fallbacks() == 0


@pytest.mark.gpu
def test_four_passes_in_flight_soak_every_net_keeps_its_bits():
    """Four shared-device ResNet50 INT8 batch-8 nets (compacted arenas: edges of disjoint lifetimes share memory) replayed together on the four
    serving streams 3 000 rounds: after every pass each net's logits and pool5 are compared on its own stream with its first pass - a pass in
    flight beside three others never sees another net's bytes or a recycled slot too early."""
    from anakin_amd import lib as L
    from anakin_amd import workloads as W
    from anakin_amd.streams import serving_streams
    L.require_device()
    model = W.framework_model(W.build_model("resnet50"), "int8")
    scales = W.calibrate(model, W.make_input(2))
    streams, distinct = serving_streams(4)
    nets, firsts, bads = [], [], []
    for i, st in enumerate(streams):
        with torch.cuda.stream(st):
            n = W.build_int8_net(model, dict(scales), 8, shared_device=True)
            n.tensor("data").copy_(torch.from_numpy(W.make_input(8, seed=71 + i)).cuda())
            n.run()
            n.compact()
            n.run()
            n.capture()
            n.replay()
            firsts.append({nm: n.tensor(nm).clone() for nm in ("fc1000", "pool5")})
            bads.append(torch.zeros((), dtype=torch.int64, device="cuda"))
        nets.append(n)
    torch.cuda.synchronize()
    assert not torch.equal(firsts[0]["fc1000"], firsts[1]["fc1000"])          # different images: different answers
    for _ in range(3000):
        for n, st, f, b in zip(nets, streams, firsts, bads):
            with torch.cuda.stream(st):
                n.replay()
                for nm in f:
                    b += (n.tensor(nm) != f[nm]).any().to(torch.int64)
    torch.cuda.synchronize()
    assert [int(b.item()) for b in bads] == [0, 0, 0, 0]
    assert all(n.coop_fallbacks() == 0 for n in nets)
