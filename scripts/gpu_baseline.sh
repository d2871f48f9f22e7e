#!/bin/bash
# baseline: per-op table (hipEvents) + in-order kernel trace of one forward at batch 8 and batch 1
set -u
TAG=${1:-base}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py --steps 200 --warmup 10 --per-op --no-cpu-baseline > $OUT/bench.json 2> $OUT/per_op.txt
for B in 8 1; do
  rocprofv3 --kernel-trace -d $OUT/trace$B -o t -- python bench.py --steps 20 --warmup 5 --timed-only --batch $B > $OUT/trace$B.log 2>&1
  NOPS=$(grep -h '"value"' $OUT/trace$B.log | head -1 | python -c "import json,sys;print(json.loads(sys.stdin.readline())['ops'])")
  python scripts/trace_sequence.py $(find $OUT/trace$B -name '*_results.db' | head -1) $NOPS 10 > $OUT/sequence_b$B.txt
  rm -rf $OUT/trace$B
done
cat $OUT/bench.json; head -5 $OUT/sequence_b8.txt
