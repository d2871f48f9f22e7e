mkdir -p gpurun_out/r05b; O=gpurun_out/r05b
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests -m gpu -q --maxfail=20 -p no:cacheprovider 2>&1 | tail -40) > $O/pytest.txt
# Worker matrix: launch form x request memory x threads
python - <<'PY' > $O/worker_matrix.txt 2>&1
import os, subprocess, sys, tempfile, time
sys.path.insert(0, os.getcwd())
from anakin_amd import workloads as W
from integration import net_model as NM
exe = os.path.join(os.getcwd(), "integration", "_build", "test_net_mi355x.bin")
model = W.build_model("resnet50"); x = W.make_input(8); scales = W.calibrate(model, W.make_input(2))
td = tempfile.mkdtemp()
mt, wb = NM.write_model(model, dict(scales), 8, td, "int8", calibrator_config=True)
x.tofile(os.path.join(td, "input.bin"))
for graph in ("0", "1"):
    for mode in ("worker", "worker_pinned"):
        for th in (1, 3, 6):
            env = dict(os.environ, SABER_MI355X_NET_PLAN_GRAPH=graph)
            r = subprocess.run([exe, mt, wb, os.path.join(td, "input.bin"), td, mode, str(th), "300"], capture_output=True, text=True, errors="replace", cwd=td, env=env, timeout=300)
            print("graph", graph, mode, "threads", th, "rc", r.returncode, open(os.path.join(td, "worker.txt")).read().strip() if r.returncode == 0 else r.stderr[-500:])
            sys.stdout.flush()
PY
timeout 300 python bench.py --steps 300 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
timeout 300 python bench.py --steps 100 --model vgg16 --precision fp32 --no-cpu-baseline --no-b1 > $O/bench_vgg16.json 2> $O/bench_vgg16.err
tail -3 $O/pytest.txt; cat $O/worker_matrix.txt
