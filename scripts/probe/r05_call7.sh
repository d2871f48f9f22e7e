mkdir -p gpurun_out/r05g; O=gpurun_out/r05g
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
(echo "nproc $(nproc)"; cat /sys/fs/cgroup/cpu.max 2>/dev/null; python -c "import os; print('affinity', len(os.sched_getaffinity(0)), 'cpu_count', os.cpu_count())"; grep -c processor /proc/cpuinfo; uptime) > $O/host_cpus.txt 2>&1
python - <<'PY' > $O/worker_sync_modes.txt 2>&1
import os, subprocess, sys, tempfile
sys.path.insert(0, os.getcwd())
from anakin_amd import workloads as W
from integration import net_model as NM
exe = os.path.join(os.getcwd(), "integration", "_build", "test_net_mi355x.bin")
model = W.build_model("resnet50"); x = W.make_input(8); scales = W.calibrate(model, W.make_input(2))
td = tempfile.mkdtemp()
mt, wb = NM.write_model(model, dict(scales), 8, td, "int8", calibrator_config=True)
x.tofile(os.path.join(td, "input.bin"))
for sync in ("", "blocking", "yield", "spin"):
    for th in (1, 3, 6):
        env = dict(os.environ)
        if sync: env["SABER_TEST_SYNC"] = sync
        r = subprocess.run([exe, mt, wb, os.path.join(td, "input.bin"), td, "worker", str(th), "400"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, errors="replace", cwd=td, timeout=300, env=env)
        print(sync or "default", "threads", th, "rc", r.returncode, open(os.path.join(td, "worker.txt")).read().strip() if r.returncode == 0 else "")
        for l in r.stdout.splitlines():
            if l.startswith("per request"): print("   ", l)
        sys.stdout.flush()
PY
cat $O/host_cpus.txt $O/worker_sync_modes.txt
