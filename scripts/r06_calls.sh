#!/bin/bash
# Round 6: every GPU call of the round as ONE parameterised runner (round-5 verdict, hygiene: 19 one-off wrappers were tracked).
#   gpurun --timeout N -- 'bash scripts/r06_calls.sh <step> [args]'        outputs under gpurun_out/r06_<step>/
set -u
STEP=${1:?step}; shift || true
O=gpurun_out/r06_$STEP
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
case $STEP in
baseline)   # this round's box: the headline line, the FP32 lines WITH the x86 FP32 baseline (BASELINE configs[0])
  python bench.py --steps 400 --no-cpu-baseline > $O/bench_b8_int8.json 2> $O/bench_b8_int8.err
  python bench.py --precision fp32 --batch 1 --steps 200 --no-b1 --cpu-seconds 30 > $O/fp32_b1.json 2> $O/fp32_b1.err
  python bench.py --precision fp32 --batch 8 --steps 200 --no-b1 --no-cpu-baseline --per-op > $O/fp32_b8.json 2> $O/fp32_b8_per_op.txt
  python bench.py --model vgg16 --precision fp32 --batch 8 --steps 100 --no-b1 --cpu-seconds 20 > $O/vgg16_b8.json 2> $O/vgg16.err
  nproc > $O/host.txt; lscpu | head -20 >> $O/host.txt ;;
*) echo "unknown step $STEP"; exit 2 ;;
esac
