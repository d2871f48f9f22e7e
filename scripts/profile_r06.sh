#!/bin/bash
# Round-6 profile of `python bench.py` on the GPU box -> gpurun_out/<tag>/ (copy what is to be judged into profiles/<tag>/). scripts/profile_r04.sh plus:
# (scripts/profile_r05.sh without the DRAM-destination pass - answered in round 5: the counter does not separate - and without round 5's probes)
set -u
TAG=${1:-r06}; shift
EXTRA="$@"
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -f $OUT/tune.json
# the untraced run autotunes on this box and WRITES the selection (batch 8 and batch 1); every later pass applies it. Copied to
# profiles/tune.json it is also what the driver's default `python bench.py` applies (keyed by configuration + source hash)
python bench.py --steps 400 --warmup 10 --per-op --retune --write-tune-cache --tune-cache $OUT/tune.json $EXTRA > $OUT/bench.json 2> $OUT/per_op.txt
for B in 2 4; do      # the other two BASELINE batch sizes into the same cache (their bench lines go to configs.jsonl later, applying these)
  python bench.py --batch $B --steps 100 --warmup 10 --no-cpu-baseline --no-b1 --retune --write-tune-cache --tune-cache $OUT/tune.json $EXTRA > $OUT/bench_b$B.json 2>/dev/null
done
ARGS="--steps 20 --warmup 5 --timed-only --tune-cache $OUT/tune.json $EXTRA"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python bench.py $ARGS > $OUT/bench_trace.log 2>&1
grep -h '"value"' $OUT/bench_trace.log | head -1 > $OUT/bench_under_trace.json
NOPS=$(python -c "import json;d=json.load(open('$OUT/bench_under_trace.json'));print(d.get('launches', d['ops']))")
DB=$(find $OUT/trace -name '*_results.db' | head -1)
python scripts/rocprof_summary.py $DB 70 $((NOPS*20)) > $OUT/kernel_trace_summary.txt
python scripts/trace_sequence.py $DB $NOPS 20 > $OUT/sequence_b8.txt
for C in FETCH_SIZE WRITE_SIZE; do
  L=$(echo $C | tr A-Z a-z | sed 's/_size//')
  rocprofv3 --kernel-trace --pmc $C -d $OUT/pmc_$L -o p -- python bench.py $ARGS --no-graph > $OUT/bench_$L.log 2>&1
  python scripts/pmc_per_kernel.py $(find $OUT/pmc_$L -name '*_results.db' | head -1) $C $NOPS > $OUT/${L}_per_kernel.txt
done
rocprofv3 --kernel-trace --pmc MfmaUtil -d $OUT/pmc_mfma -o m -- python bench.py $ARGS --no-graph > $OUT/bench_mfma.log 2>&1
python scripts/pmc_kernel_avg.py $(find $OUT/pmc_mfma -name '*_results.db' | head -1) MfmaUtil $NOPS > $OUT/mfma_util.txt
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD -d $OUT/pmc_inst -o i -- python bench.py $ARGS --no-graph > $OUT/bench_inst.log 2>&1
python scripts/pmc_mix.py $(find $OUT/pmc_inst -name '*_results.db' | head -1) $NOPS > $OUT/inst_mix_per_launch.txt
rm -rf $OUT/pmc_inst
python scripts/make_traffic_json.py $OUT 8 framework
rm -rf $OUT/trace $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_mfma
# batch 1 (half of the headline metric is its p50): the same trace for the batch-1 list, with ITS tuned selection
ARGS1="--batch 1 --steps 20 --warmup 5 --timed-only --tune-cache $OUT/tune.json $EXTRA"
rocprofv3 --kernel-trace --stats -d $OUT/trace1 -o t -- python bench.py $ARGS1 > $OUT/bench_trace_b1.log 2>&1
grep -h '"value"' $OUT/bench_trace_b1.log | head -1 > $OUT/bench_b1_under_trace.json
NOPS1=$(python -c "import json;d=json.load(open('$OUT/bench_b1_under_trace.json'));print(d.get('launches', d['ops']))")
DB1=$(find $OUT/trace1 -name '*_results.db' | head -1)
python scripts/rocprof_summary.py $DB1 70 $((NOPS1*20)) > $OUT/kernel_trace_summary_b1.txt
python scripts/trace_sequence.py $DB1 $NOPS1 20 > $OUT/sequence_b1.txt
rm -rf $OUT/trace1
head -3 $OUT/sequence_b1.txt
head -3 $OUT/sequence_b8.txt; tail -1 $OUT/mfma_util.txt; head -20 $OUT/traffic.json
