# L2 requests (hits, misses) per launch of ResNet50 FP32 batch 8 with the committed selection of profiles/r05_resnet50_fp32/tune.json: what the
# "bytes pulled from the L2s" of DESIGN 4.8 are, measured (separate PMC pass, --kernel-trace only)
mkdir -p gpurun_out/r05q; O=gpurun_out/r05q
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 -L 2>/dev/null | grep -o -E "TCC_HIT[A-Za-z0-9_]*|TCC_MISS[A-Za-z0-9_]*|TCC_REQ[A-Za-z0-9_]*|TCC_READ[A-Za-z0-9_]*|TCP_TCC_READ_REQ[A-Za-z0-9_]*|TCP_TOTAL_CACHE_ACCESSES[A-Za-z0-9_]*" | sort -u > $O/l2_counters_available.txt
ARGS="--precision fp32 --steps 10 --warmup 3 --timed-only --no-graph --tune-cache profiles/r05_resnet50_fp32/tune.json"
python bench.py $ARGS > $O/bench_plain.log 2>&1
N=$(python -c "import json;d=json.loads([l for l in open('$O/bench_plain.log') if '\"value\"' in l][0]);print(d.get('launches', d['ops']))")
echo "launches: $N" > $O/l2_requests_per_kernel.txt
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d $O/pmc_l2 -o p -- python bench.py $ARGS > $O/bench_l2.log 2>&1
python scripts/pmc_two_counters.py $(find $O/pmc_l2 -name '*_results.db' | head -1) TCC_HIT_sum TCC_MISS_sum $N >> $O/l2_requests_per_kernel.txt 2>&1
rocprofv3 --kernel-trace --pmc TCC_REQ_sum TCC_READ_sum -d $O/pmc_l2b -o p -- python bench.py $ARGS > $O/bench_l2b.log 2>&1
python scripts/pmc_two_counters.py $(find $O/pmc_l2b -name '*_results.db' | head -1) TCC_REQ_sum TCC_READ_sum $N > $O/l2_req_read_per_kernel.txt 2>&1
rm -rf $O/pmc_l2 $O/pmc_l2b
cat $O/l2_counters_available.txt | tr '\n' ' '; echo; cat $O/l2_requests_per_kernel.txt | cut -c1-150; tail -3 $O/l2_req_read_per_kernel.txt | cut -c1-200
