"""What the host link gives a Worker request: a batch-8 f32 image tensor (8 x 3 x 224 x 224 x 4 B = 4.8 MB) from pinned and from pageable
host memory to the device, one copy at a time and several streams at once - the PCIe-inclusive bound of Worker<MI355X>::sync_prediction
(requests per second = host-to-device GB/s / 4.8 MB). usage: python scripts/probe/pcie_probe.py   (GPU box)"""
import json
import time

import torch

out = {}
for mb, label in ((4.816896, "request_4.8MB"), (64.0, "64MB")):
    n = int(mb * 1e6) // 4
    dev = torch.empty(n, dtype=torch.float32, device="cuda")
    pin = torch.empty(n, dtype=torch.float32).pin_memory()
    pag = torch.empty(n, dtype=torch.float32)
    pin.fill_(1.0)
    pag.fill_(1.0)
    for name, src in (("pinned", pin), ("pageable", pag)):
        for _ in range(5):
            dev.copy_(src, non_blocking=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 100 if mb < 10 else 20
        for _ in range(reps):
            dev.copy_(src, non_blocking=True)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        out["%s_%s" % (label, name)] = {"ms": round(dt * 1e3, 4), "GBps": round(n * 4 / dt / 1e9, 2)}
    # three streams at once (three Worker threads' copy lanes)
    streams = [torch.cuda.Stream() for _ in range(3)]
    devs = [torch.empty(n, dtype=torch.float32, device="cuda") for _ in range(3)]
    pins = [torch.empty(n, dtype=torch.float32).pin_memory() for _ in range(3)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 60 if mb < 10 else 10
    for _ in range(reps):
        for s, d, p in zip(streams, devs, pins):
            with torch.cuda.stream(s):
                d.copy_(p, non_blocking=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / (reps * 3)
    out["%s_pinned_3_streams" % label] = {"ms_per_copy": round(dt * 1e3, 4), "GBps_aggregate": round(n * 4 / dt / 1e9, 2)}
r = out["request_4.8MB_pinned_3_streams"]["GBps_aggregate"]
out["worker_bound"] = {"requests_per_s": round(r * 1e9 / 4816896, 1), "images_per_s": round(8 * r * 1e9 / 4816896, 1),
                       "what": "batch-8 f32 requests per second the host-to-device link carries (pinned, three streams)"}
print(json.dumps(out))
