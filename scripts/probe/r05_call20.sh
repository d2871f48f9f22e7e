# Net<MI355X, INT8>::prediction(): does the per-call time depend on how many calls came before the timed ones / how many are timed?
mkdir -p gpurun_out/r05p; O=gpurun_out/r05p
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python - <<'PY' > $O/prediction_warmup.txt 2>&1
import os, subprocess, sys, tempfile
sys.path.insert(0, os.getcwd())
from anakin_amd import workloads as W
from integration import net_model as NM
exe = os.path.join(os.getcwd(), "integration", "_build", "test_net_mi355x.bin")
model = W.build_model("resnet50"); x = W.make_input(8); scales = W.calibrate(model, W.make_input(2))
td = tempfile.mkdtemp()
mt, wb = NM.write_model(model, dict(scales), 8, td, "int8", calibrator_config=True)
x.tofile(os.path.join(td, "input.bin"))
for warm, iters in ((10, 200), (300, 200), (1000, 200), (1000, 2000), (10, 2000)):
    env = dict(os.environ, SABER_TEST_WARMUP=str(warm))
    r = subprocess.run([exe, mt, wb, os.path.join(td, "input.bin"), td, str(iters)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, errors="replace", cwd=td, timeout=600, env=env)
    print("warm-up", warm, "timed", iters, "rc", r.returncode, open(os.path.join(td, "timing.txt")).read().strip() if r.returncode == 0 else r.stdout[-300:])
    sys.stdout.flush()
PY
cat $O/prediction_warmup.txt
