mkdir -p gpurun_out/r05c; O=gpurun_out/r05c
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
(timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py tests/test_gpu_resnet.py -m gpu -q --maxfail=20 -p no:cacheprovider -k "pointwise or fc or gemm or fp32 or vgg or softmax" 2>&1 | tail -30) > $O/pytest.txt
python scripts/probe/pcie_probe.py > $O/pcie_probe.txt 2>&1
./scripts/probe/timeline_probe.bin tail > $O/timeline_tail.txt 2>&1
timeout 300 python bench.py --steps 100 --model vgg16 --precision fp32 --no-cpu-baseline --no-b1 --per-op > $O/bench_vgg16.json 2> $O/per_op_vgg16.txt
timeout 300 python bench.py --steps 200 --precision fp32 --no-cpu-baseline --no-b1 --per-op > $O/bench_r50_fp32.json 2> $O/per_op_r50_fp32.txt
SABER_HIP_AUTOTUNE_LOG=1 timeout 300 python bench.py --steps 50 --precision fp32 --no-cpu-baseline --no-b1 2> $O/autotune_log_fp32.txt | head -c 300 > /dev/null
python - <<'PY' > $O/gemm_fc6.txt 2>&1
import torch, sys, os
sys.path.insert(0, os.getcwd())
from anakin_amd import saber as S
fm, fk, fn = 8, 25088, 4096
fa = torch.randn(fm, fk, device="cuda"); fb = torch.randn(fn, fk, device="cuda"); fc_ = torch.empty(fm, fn, device="cuda")
for _ in range(3): S.gemm(False, True, fm, fn, fk, 1.0, fa, fb, 0.0, fc_)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for rep in range(3):
    e0.record()
    for _ in range(10): S.gemm(False, True, fm, fn, fk, 1.0, fa, fb, 0.0, fc_)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    print("fc6 as Gemm (m=8, k=25088, n=4096): %.1f us = %.2f TB/s = %.3f of 8 TB/s" % (us, fn * fk * 4 / us / 1e6, fn * fk * 4 / us / 1e6 / 8))
ref = (fa.double() @ fb.double().T)
print("max rel err vs f64:", float((fc_.double() - ref).abs().max() / ref.abs().max()))
PY
tail -3 $O/pytest.txt; cat $O/pcie_probe.txt; cat $O/gemm_fc6.txt; grep -A8 "fc_i8_small + softmax" $O/timeline_tail.txt | head -40
