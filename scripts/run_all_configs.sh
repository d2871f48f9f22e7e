#!/bin/bash
# All BASELINE.json configurations on one MI355X: ResNet50 INT8 / FP32 at batch 1/2/4/8, VGG16 FP32 b=8,
# ResNet101 INT8 b=8. One JSON line per run into gpurun_out/configs.jsonl
mkdir -p gpurun_out; : > gpurun_out/configs.jsonl
for prec in int8 fp32; do for b in 1 2 4 8; do
  timeout 500 python bench.py --steps 200 --batch $b --precision $prec --no-cpu-baseline --no-b1 2>/dev/null >> gpurun_out/configs.jsonl
done; done
timeout 500 python bench.py --steps 100 --model vgg16 --precision fp32 --no-cpu-baseline --no-b1 2>/dev/null >> gpurun_out/configs.jsonl
timeout 500 python bench.py --steps 200 --model resnet101 --no-cpu-baseline --no-b1 2>/dev/null >> gpurun_out/configs.jsonl
python - <<'PY'
import json
for l in open('gpurun_out/configs.jsonl'):
    d=json.loads(l); r=d["roofline"]
    print("%-46s %10.1f img/s  %8.4f ms/step  p50 %.4f ms  roofline %s %.1f (%.3f)" % (d["config"]["workload"][:46], d["value"], d["ms_per_step"], d["latency_ms"]["p50"], r["bound"], r["achieved"], r["frac"]))
PY
