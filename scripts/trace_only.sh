#!/bin/bash
# kernel trace of one tuned configuration -> gpurun_out/<tag>/sequence.txt (+ per-op table); usage: trace_only.sh <tag> [bench flags]
set -u
TAG=${1:-t}; shift
O=gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -f $O/tune.json
python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-b1 --per-op --tune-cache $O/tune.json "$@" > $O/bench.json 2> $O/per_op.txt
rocprofv3 --kernel-trace --stats -d $O/trace -o t -- python bench.py --steps 20 --warmup 5 --timed-only --tune-cache $O/tune.json "$@" > $O/bench_trace.log 2>&1
grep -h '"value"' $O/bench_trace.log | head -1 > $O/bench_under_trace.json
N=$(python -c "import json;d=json.load(open('$O/bench_under_trace.json'));print(d.get('launches', d['ops']))")
DB=$(find $O/trace -name '*_results.db' | head -1)
python scripts/trace_sequence.py $DB $N 20 > $O/sequence.txt
rm -rf $O/trace
cat $O/sequence.txt
