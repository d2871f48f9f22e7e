#!/bin/bash
# Where do the idle gaps of the forward pass come from (profiles/r04/sequence_b8.txt: 2 - 5 us in front of launch #5, the res3a pair)?
# The eager pass traced once more, with the gap of every forward kept apart (scripts/trace_sequence.py: median / max / how many of the 20
# forwards have it): a gap present in every forward belongs to that kernel's dispatch, one that is a single stall of the launching
# thread in one forward averages to "a few us in front of launch #k" with a different k in every run. (hipGraph replay under the
# tracer is no control: the tracer instruments every graph node - 48 us of gaps, 0.31 ms per step.)
# usage (on the GPU box): bash scripts/probe/gap_probe.sh <outdir>
set -u
O=${1:-gpurun_out/gap}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ARGS="--steps 20 --warmup 5 --timed-only"
for mode in eager; do
  F=$([ $mode = graph ] && echo --force-graph || echo --no-graph)
  rocprofv3 --kernel-trace --stats -d $O/t_$mode -o t -- python bench.py $ARGS $F > $O/bench_$mode.log 2>&1
  N=$(python -c "import json;print(json.loads([l for l in open('$O/bench_$mode.log') if '\"value\"' in l][0])['launches'])")
  python scripts/trace_sequence.py $(find $O/t_$mode -name '*_results.db' | head -1) $N 20 > $O/sequence_$mode.txt
  rm -rf $O/t_$mode
  echo "== $mode"; head -1 $O/sequence_$mode.txt; awk 'NR>2 && $3+0 > 0.3' $O/sequence_$mode.txt
done
