mkdir -p gpurun_out/r05e; O=gpurun_out/r05e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
(timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py tests/test_gpu_net.py -m gpu -q --maxfail=20 -p no:cacheprovider -k "fc or gemm or vgg or worker" 2>&1 | tail -8) > $O/pytest.txt
python - <<'PY' > $O/worker_breakdown.txt 2>&1
import os, subprocess, sys, tempfile
sys.path.insert(0, os.getcwd())
from anakin_amd import workloads as W
from integration import net_model as NM
exe = os.path.join(os.getcwd(), "integration", "_build", "test_net_mi355x.bin")
model = W.build_model("resnet50"); x = W.make_input(8); scales = W.calibrate(model, W.make_input(2))
td = tempfile.mkdtemp()
mt, wb = NM.write_model(model, dict(scales), 8, td, "int8", calibrator_config=True)
x.tofile(os.path.join(td, "input.bin"))
for mode in ("worker_pinned", "worker"):
    for th in (1, 2, 3, 4):
        r = subprocess.run([exe, mt, wb, os.path.join(td, "input.bin"), td, mode, str(th), "400"], capture_output=True, text=True, errors="replace", cwd=td, timeout=300)
        print(mode, "threads", th, "rc", r.returncode, open(os.path.join(td, "worker.txt")).read().strip() if r.returncode == 0 else r.stderr[-400:])
        for l in r.stdout.splitlines():
            if l.startswith("per request"): print("   ", l)
        sys.stdout.flush()
PY
timeout 300 python bench.py --steps 100 --model vgg16 --precision fp32 --no-cpu-baseline --no-b1 > $O/bench_vgg16.json 2>/dev/null
cat $O/pytest.txt | tail -3; cat $O/worker_breakdown.txt; python -c "
import json; v=json.load(open('$O/bench_vgg16.json')); print('VGG16', v['value'], v['ms_per_step']); [print(k['kernel'], k['launches'], k['avg_us'], k['gbs']) for k in v['roofline']['per_kernel'] if k['kernel'].startswith('fc')]"
