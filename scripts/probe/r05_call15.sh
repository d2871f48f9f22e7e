mkdir -p gpurun_out/r05k; O=gpurun_out/r05k
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --maxfail=20 -p no:cacheprovider -k "pointwise" 2>&1 | tail -8) > $O/pytest_pw.txt
(timeout 1500 python -m pytest tests/test_gpu_baseline_configs.py -m gpu -q --maxfail=20 -p no:cacheprovider -k "fp32" 2>&1 | tail -15) > $O/pytest_fp32_nets.txt
for b in 1 2 4 8; do
  timeout 300 python bench.py --steps 200 --precision fp32 --batch $b --no-cpu-baseline --no-b1 > $O/bench_r50_fp32_b$b.json 2> $O/log_b$b.txt
done
tail -4 $O/pytest_pw.txt; tail -6 $O/pytest_fp32_nets.txt
for b in 1 2 4 8; do python -c "
import json; v=json.load(open('$O/bench_r50_fp32_b$b.json')); print('r50 fp32 b$b', v['value'], v['ms_per_step'], v['config']['launches'])"; done
