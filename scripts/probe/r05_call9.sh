mkdir -p gpurun_out/r05g; O=gpurun_out/r05g
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python - <<'PY' > $O/worker_vs_threads.txt 2>&1
import os, subprocess, sys, tempfile
sys.path.insert(0, os.getcwd())
from anakin_amd import workloads as W
from integration import net_model as NM
exe = os.path.join(os.getcwd(), "integration", "_build", "test_net_mi355x.bin")
model = W.build_model("resnet50"); x = W.make_input(8); scales = W.calibrate(model, W.make_input(2))
td = tempfile.mkdtemp()
mt, wb = NM.write_model(model, dict(scales), 8, td, "int8", calibrator_config=True)
x.tofile(os.path.join(td, "input.bin"))
for mode in ("threads", "worker"):
    for th in (1, 2, 3, 6):
        r = subprocess.run([exe, mt, wb, os.path.join(td, "input.bin"), td, mode, str(th), "600"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, errors="replace", cwd=td, timeout=300)
        print(mode, "threads", th, "rc", r.returncode, open(os.path.join(td, "worker.txt")).read().strip() if r.returncode == 0 else "")
        for l in r.stdout.splitlines():
            if l.startswith("per request"): print("   ", l)
        sys.stdout.flush()
PY
cat $O/worker_vs_threads.txt
