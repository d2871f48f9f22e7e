mkdir -p gpurun_out/r05d; O=gpurun_out/r05d
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
(timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py -m gpu -q --maxfail=20 -p no:cacheprovider -k "fc or gemm or vgg or softmax" 2>&1 | tail -8) > $O/pytest.txt
./scripts/probe/timeline_probe.bin tail > $O/timeline_tail.txt 2>&1
python - <<'PY' > $O/worker_env.txt 2>&1
import os, subprocess, sys, tempfile
sys.path.insert(0, os.getcwd())
from anakin_amd import workloads as W
from integration import net_model as NM
exe = os.path.join(os.getcwd(), "integration", "_build", "test_net_mi355x.bin")
model = W.build_model("resnet50"); x = W.make_input(8); scales = W.calibrate(model, W.make_input(2))
td = tempfile.mkdtemp()
mt, wb = NM.write_model(model, dict(scales), 8, td, "int8", calibrator_config=True)
x.tofile(os.path.join(td, "input.bin"))
def run(mode, th, extra):
    env = dict(os.environ, **extra)
    r = subprocess.run([exe, mt, wb, os.path.join(td, "input.bin"), td, mode, str(th), "300"], capture_output=True, text=True, errors="replace", cwd=td, env=env, timeout=300)
    print(extra, mode, "threads", th, "rc", r.returncode, open(os.path.join(td, "worker.txt")).read().strip() if r.returncode == 0 else r.stderr[-400:])
    sys.stdout.flush()
for extra in ({}, {"GPU_MAX_HW_QUEUES": "8"}, {"GPU_MAX_HW_QUEUES": "16"}, {"GPU_MAX_HW_QUEUES": "16", "SABER_MI355X_NET_PLAN_GRAPH": "1"}):
    for mode in ("worker_pinned", "worker"):
        for th in (3, 6):
            run(mode, th, extra)
# Net::prediction() per call: the completion wait
for extra in ({}, {"ROC_ACTIVE_WAIT_TIMEOUT": "100000"}, {"HIP_FORCE_DEV_KERNARG": "1", "ROC_ACTIVE_WAIT_TIMEOUT": "1000000"}):
    env = dict(os.environ, **extra)
    r = subprocess.run([exe, mt, wb, os.path.join(td, "input.bin"), td, "300"], capture_output=True, text=True, errors="replace", cwd=td, env=env, timeout=300)
    print(extra, "prediction rc", r.returncode, open(os.path.join(td, "timing.txt")).read().strip()[:400] if r.returncode == 0 else r.stderr[-400:])
PY
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
timeout 300 python bench.py --steps 100 --model vgg16 --precision fp32 --no-cpu-baseline --no-b1 > $O/bench_vgg16.json 2>/dev/null
rocprofv3 -L 2>/dev/null | grep -o -E "TCC_EA0?_RDREQ[A-Z0-9_]*|TCC_EA0?_WRREQ[A-Z0-9_]*|TCC_[A-Z0-9_]*MALL[A-Z0-9_]*|TCC_[A-Z0-9_]*DRAM[A-Z0-9_]*|TCC_[A-Z0-9_]*IO[A-Z0-9_]*|TCC_[A-Z0-9_]*GMI[A-Z0-9_]*" | sort -u > $O/counters_available.txt
cat $O/pytest.txt | tail -3; cat $O/worker_env.txt; cat $O/gemm_fc6.txt; grep -A8 "fc_i8_small + softmax" $O/timeline_tail.txt | head -10; python -c "
import json; v=json.load(open('$O/bench_vgg16.json')); print('VGG16', v['value'], v['ms_per_step']); [print(k['kernel'], k['launches'], k['avg_us']) for k in v['roofline']['per_kernel'] if k['kernel'].startswith('fc')]"; cat $O/counters_available.txt | tr '\n' ' '
