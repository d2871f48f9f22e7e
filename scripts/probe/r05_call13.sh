mkdir -p gpurun_out/r05i; O=gpurun_out/r05i
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
(timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --maxfail=20 -p no:cacheprovider -k "small_image_3x3 or halo" 2>&1 | tail -25) > $O/pytest.txt
SABER_HIP_AUTOTUNE_LOG=1 timeout 300 python bench.py --steps 200 --precision fp32 --no-cpu-baseline --no-b1 --per-op > $O/bench_r50_fp32.json 2> $O/log_r50_fp32.txt
tail -12 $O/pytest.txt; python -c "
import json; v=json.load(open('$O/bench_r50_fp32.json')); print('r50 fp32', v['value'], v['ms_per_step'], v['config']['launches']); [print('  ', k['kernel'], k['launches'], k['avg_us']) for k in v['roofline']['per_kernel'][:14]]"; grep "img3x3_f32" $O/log_r50_fp32.txt | grep autotune | sort | uniq -c | head -20
