#!/bin/bash
# per-kernel A/B at IDENTICAL (static, un-autotuned) kernel choices: kernel trace of one forward in ab/old_tree and here
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/abtrace; mkdir -p $OUT
for t in old new; do
  D=$GRAFT_REPO_ROOT; [ $t = old ] && D=$GRAFT_REPO_ROOT/ab/old_tree
  (cd $D && rocprofv3 --kernel-trace -d $OUT/tr_$t -o t -- python bench.py --steps 30 --warmup 5 --timed-only --no-autotune "$@" > $OUT/$t.log 2>&1)
  python scripts/trace_sequence.py $(find $OUT/tr_$t -name '*_results.db' | head -1) 52 20 > $OUT/seq_$t.txt
  rm -rf $OUT/tr_$t
done
