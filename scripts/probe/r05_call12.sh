mkdir -p gpurun_out/r05h; O=gpurun_out/r05h
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
(timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py tests/test_gpu_resnet.py -m gpu -q --maxfail=20 -p no:cacheprovider -k "gemm or fc or vgg or fp32" 2>&1 | tail -8) > $O/pytest2.txt
timeout 300 python bench.py --steps 100 --model vgg16 --precision fp32 --no-cpu-baseline --no-b1 --per-op > $O/bench_vgg16.json 2> $O/per_op_vgg16.txt
SABER_HIP_FC_F32_PACKED=1 timeout 300 python bench.py --steps 100 --model vgg16 --precision fp32 --no-cpu-baseline --no-b1 --per-op > $O/bench_vgg16_packed.json 2> $O/per_op_vgg16_packed.txt
timeout 300 python bench.py --steps 200 --precision fp32 --no-cpu-baseline --no-b1 --per-op > $O/bench_r50_fp32.json 2> $O/per_op_r50_fp32.txt
tail -3 $O/pytest2.txt; for f in vgg16 vgg16_packed r50_fp32; do python -c "
import json; v=json.load(open('$O/bench_$f.json')); print('$f', v['value'], v['ms_per_step']); [print('  ', k['kernel'], k['launches'], k['avg_us'], k['gbs']) for k in v['roofline']['per_kernel'] if k['kernel'].startswith('fc')]"; grep " fc" $O/per_op_$f.txt; done
