#!/bin/bash
# kernel trace + MfmaUtil of another configuration -> gpurun_out/<tag>/ ; usage: profile_model.sh <tag> <bench flags...>
set -u
TAG=$1; shift
O=gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -f $O/tune.json
python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-b1 --per-op --retune --write-tune-cache --tune-cache $O/tune.json "$@" > $O/bench.json 2> $O/per_op.txt
ARGS="--steps 10 --warmup 3 --timed-only --tune-cache $O/tune.json $*"
rocprofv3 --kernel-trace --stats -d $O/trace -o t -- python bench.py $ARGS > $O/bench_trace.log 2>&1
grep -h '"value"' $O/bench_trace.log | head -1 > $O/bench_under_trace.json
N=$(python -c "import json;d=json.load(open('$O/bench_under_trace.json'));print(d.get('launches', d['ops']))")
DB=$(find $O/trace -name '*_results.db' | head -1)
python scripts/rocprof_summary.py $DB 70 $((N*10)) > $O/kernel_trace_summary.txt
python scripts/trace_sequence.py $DB $N 10 > $O/sequence.txt
rocprofv3 --kernel-trace --pmc MfmaUtil -d $O/pmc -o m -- python bench.py $ARGS --no-graph > $O/bench_mfma.log 2>&1
python scripts/pmc_kernel_avg.py $(find $O/pmc -name '*_results.db' | head -1) MfmaUtil $N > $O/mfma_util.txt
rm -rf $O/trace $O/pmc $O/bench_trace.log $O/bench_mfma.log
head -3 $O/sequence.txt; tail -1 $O/mfma_util.txt
