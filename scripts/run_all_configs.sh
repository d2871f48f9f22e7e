#!/bin/bash
# All BASELINE.json configurations on one MI355X: ResNet50 INT8 / FP32 at batch 1/2/4/8, VGG16 FP32 b=8,
# ResNet101 INT8 b=8. One JSON line per run into gpurun_out/configs.jsonl
mkdir -p gpurun_out; : > gpurun_out/configs.jsonl
# (INT8 lines: the x86 baseline is in the default bench line; FP32 lines - round 6 - carry cpu_baseline.kind "reference": the FP32 op list, batch 1, through the
# reference's own x86 objects on this box's host, BASELINE.json configs[0])
for b in 1 2 4 8; do
  timeout 500 python bench.py --steps 200 --batch $b --precision int8 --no-cpu-baseline --no-b1 2>/dev/null >> gpurun_out/configs.jsonl
done
for b in 1 2 4 8; do
  timeout 500 python bench.py --steps 200 --batch $b --precision fp32 --cpu-seconds 10 --no-b1 2>/dev/null >> gpurun_out/configs.jsonl
done
timeout 500 python bench.py --steps 100 --model vgg16 --precision fp32 --cpu-seconds 12 --no-b1 2>/dev/null >> gpurun_out/configs.jsonl
timeout 500 python bench.py --steps 200 --model resnet101 --no-cpu-baseline --no-b1 2>/dev/null >> gpurun_out/configs.jsonl
python - <<'PY'
import json
for l in open('gpurun_out/configs.jsonl'):
    d=json.loads(l); r=d["roofline"]
    c=d.get("cpu_baseline") or {}
    print("%-46s %10.1f img/s  %8.4f ms/step  p50 %.4f ms  launches %d  roofline %s %.1f (%.3f)  cpu %s" % (d["config"]["workload"][:46], d["value"], d["ms_per_step"], d["latency_ms"]["p50"], d["config"]["launches"], r["bound"], r["achieved"], r["frac"], ("%.1f img/s (%s, %d cores)" % (c["value"], c["kind"], c["cores"])) if c.get("value") else "-"))
PY
