#!/bin/bash
# Profiles `python bench.py` on the GPU box: kernel-trace stats, then FETCH_SIZE and WRITE_SIZE PMC passes
# (separate runs, kernel-trace only, as MI355X_MICROARCH.md prescribes). Outputs under gpurun_out/<tag>/.
set -u
TAG=${1:-r01}
shift
EXTRA="$@"          # e.g. --model vgg16 --precision fp32
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ARGS="--steps 20 --warmup 5 --timed-only $EXTRA"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python bench.py $ARGS > $OUT/bench_trace.log 2>&1
grep -h '"value"' $OUT/bench_trace.log | head -1 > $OUT/bench_under_trace.json
NOPS=$(python -c "import json;print(json.load(open('$OUT/bench_under_trace.json'))['ops'])")   # one launch per op
python scripts/rocprof_summary.py $OUT/trace/t_results.db 70 $((NOPS*20)) > $OUT/kernel_trace_summary.txt
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -o f -- python bench.py $ARGS --no-graph > $OUT/bench_fetch.log 2>&1
python scripts/pmc_traffic.py $OUT/fetch/f_results.db FETCH_SIZE $NOPS > $OUT/fetch_size.json
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -o w -- python bench.py $ARGS --no-graph > $OUT/bench_write.log 2>&1
python scripts/pmc_traffic.py $OUT/write/w_results.db WRITE_SIZE $NOPS > $OUT/write_size.json
rocprofv3 --kernel-trace --pmc MfmaUtil -d $OUT/mfma -o m -- python bench.py $ARGS --no-graph > $OUT/bench_mfma.log 2>&1
python scripts/pmc_kernel_avg.py $OUT/mfma/m_results.db MfmaUtil $NOPS > $OUT/mfma_util.txt
rm -rf $OUT/trace $OUT/fetch $OUT/write $OUT/mfma
head -12 $OUT/kernel_trace_summary.txt; cat $OUT/fetch_size.json $OUT/write_size.json
