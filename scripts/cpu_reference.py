"""Times the ResNet50 INT8 op list through the reference's own x86 objects (oracle/_ref; oracle/net_oracle.RefNet)
on THIS host with the reference README's protocol (8 threads, warm-up 10, average of 200 runs, README.md:85-86) and
writes profiles/r02/cpu_reference.json. Build container only (needs oracle/_ref)."""
import json, os, platform, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from anakin_amd import workloads as W
from oracle import net_oracle as NO

model = W.build_model("resnet50")
scales = W.calibrate(model, W.make_input(2))
out = {"what": "ResNet50 INT8 unfused reference op list through oracle/_ref (GemmX8S8S32XConv + MKL cblas_gemm_s8u8s32, "
               "SaberEltwise, PackedMKLInt8Gemm), 224x224, warm-up 10 / 200 timed forwards",
       "host": {"cpu": [l.split(":")[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0],
                "logical_cpus": os.cpu_count()},
       "reference_readme": "3.21 ms/image batch 1, 8 threads, Xeon Gold 6271, JIT-VNNI path (README.md:92)", "runs": []}
for batch in (1, 8):
    rn = NO.RefNet(model, dict(scales), batch)
    rn.run(W.make_input(batch))
    for th in (8, 1) if batch == 1 else (8,):
        NO.ref_set_threads(th)
        iters = 200 if batch == 1 and th == 8 else 30
        ms = rn.time_ms(10, iters)
        out["runs"].append({"batch": batch, "threads": th, "iters": iters, "ms_per_forward": round(ms, 3),
                            "images_per_s": round(batch * 1000.0 / ms, 2)})
        print(out["runs"][-1])
json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r02", "cpu_reference.json"), "w"), indent=1)
