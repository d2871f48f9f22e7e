#!/bin/bash
# Same-box A/B of two source trees: ${AB_TREE:-ab/old_tree} (a built copy of an earlier commit) against the working tree,
# alternating `bench.py --timed-only` runs. Usage (on the GPU box): bash scripts/ab_tree.sh [rounds] [extra bench flags]
R=${1:-3}; shift
mkdir -p gpurun_out/ab
for r in $(seq 1 $R); do
  (cd ${AB_TREE:-ab/old_tree} && timeout 300 python bench.py --steps 400 --timed-only "$@" 2>/dev/null | sed "s/^/old $r /")
  timeout 300 python bench.py --steps 400 --timed-only "$@" 2>/dev/null | sed "s/^/new $r /"
done | tee gpurun_out/ab/ab_tree.log
