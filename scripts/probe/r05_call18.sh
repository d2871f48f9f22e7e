# Worker<MI355X, INT8> with the Worker translation unit compiled under the reference logger's release switch (integration/build_quiet_worker.sh)
# against the ordinary build and against plain threads, 1 / 2 / 3 / 6 threads x 600 batch-8 requests
mkdir -p gpurun_out/r05n; O=gpurun_out/r05n
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python - <<'PY' > $O/worker_quiet.txt 2>&1
import os, subprocess, sys, tempfile
sys.path.insert(0, os.getcwd())
from anakin_amd import workloads as W
from integration import net_model as NM
bd = os.path.join(os.getcwd(), "integration", "_build")
model = W.build_model("resnet50"); x = W.make_input(8); scales = W.calibrate(model, W.make_input(2))
td = tempfile.mkdtemp()
mt, wb = NM.write_model(model, dict(scales), 8, td, "int8", calibrator_config=True)
x.tofile(os.path.join(td, "input.bin"))
for exe, mode in (("test_net_mi355x.bin", "threads"), ("test_net_mi355x.bin", "worker"), ("test_net_mi355x_quiet.bin", "worker"), ("test_net_mi355x.bin", "worker_pinned"), ("test_net_mi355x.bin", "worker_async")):
    for th in (1, 2, 3, 6):
        r = subprocess.run([os.path.join(bd, exe), mt, wb, os.path.join(td, "input.bin"), td, mode, str(th), "600"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, errors="replace", cwd=td, timeout=600)
        print(exe, mode, "threads", th, "rc", r.returncode, open(os.path.join(td, "worker.txt")).read().strip() if r.returncode == 0 else "")
        for l in r.stdout.splitlines():
            if l.startswith("per request") or "between a thread" in l or l.startswith("warm-up"): print("   ", l)
        sys.stdout.flush()
PY
cat $O/worker_quiet.txt
