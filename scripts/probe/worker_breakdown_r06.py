import os, subprocess, tempfile, sys
sys.path.insert(0, os.getcwd())
from anakin_amd import workloads as W
from integration import net_model as NM
exe = os.path.abspath(os.path.join("integration", "_build", "test_net_mi355x.bin"))
model = W.framework_model(W.build_model("resnet50"), "int8")
scales = W.calibrate(model, W.make_input(2))
with tempfile.TemporaryDirectory() as td:
    base = W.build_model("resnet50")
    mt, wb = NM.write_model(base, dict(scales), 8, td, "int8", calibrator_config=True)
    W.make_input(8).tofile(os.path.join(td, "input.bin"))
    for mode, threads, window in (("worker", 4, 2), ("worker", 4, 4), ("worker", 4, 8), ("worker", 6, 4), ("worker", 8, 4), ("worker", 3, 4), ("worker_pinned", 4, 4), ("threads", 4, 2)):
        r = subprocess.run([exe, mt, wb, os.path.join(td, "input.bin"), td, mode, str(threads), "1200"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, errors="replace", cwd=td, timeout=600,
                           env=dict(os.environ, SABER_TEST_WINDOW=str(window)))
        print("=====", mode, threads, "window %d x threads" % window)
        print("\n".join(l for l in r.stdout.split("\n") if "us" in l or "ok" in l or "per request" in l)[-3000:])
