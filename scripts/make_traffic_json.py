"""Combines the FETCH_SIZE and WRITE_SIZE passes of scripts/profile_r02.sh (<dir>/fetch_per_kernel.txt,
write_per_kernel.txt) into <dir>/traffic.json, which bench.py reads for `roofline.traffic` — but only while the kernel
sources it was measured on are unchanged (`src_sha`, anakin_amd.lib.source_sha()). gfx950 correction per
MI355X_MICROARCH.md section HBM: FETCH_SIZE doubled; WRITE_SIZE as reported."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anakin_amd import lib as L  # noqa: E402

d = sys.argv[1]
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 8
f = json.loads(open(os.path.join(d, "fetch_per_kernel.txt")).read().strip().splitlines()[-1])
w = json.loads(open(os.path.join(d, "write_per_kernel.txt")).read().strip().splitlines()[-1])
assert f["conv_dispatches"] == w["conv_dispatches"] and f["dispatches"] == w["dispatches"]
n = f["conv_dispatches"]
hbm = int(f["sum_conv_KB"] * 1024 * 2 + w["sum_conv_KB"] * 1024)
# per kernel function: the dispatches of the last forward pass are the op list's launches in order (per_op.txt, minus the ops
# absorbed into a chain launch), so HBM bytes per launch can be averaged per op name = the key bench.py's per-kernel roofline uses
per_kernel = {}
graph = sys.argv[3] if len(sys.argv) > 3 else "framework"
try:
    names = []
    for line in open(os.path.join(d, "per_op.txt")):
        t = line.split()
        if len(t) >= 5 and t[0].isdigit() and t[2] == "us":
            nm = " ".join(t[5:]) if t[4] == "us" else " ".join(t[3:])
            if "(in the chain launch)" not in nm and "(in the stage launch)" not in nm and "(in the stem launch)" not in nm \
                    and "(in the fc launch)" not in nm:
                names.append(nm)
    fb, wb = f["per_dispatch_bytes_corrected"], w["per_dispatch_bytes_corrected"]
    if len(names) == len(fb) == len(wb):
        acc = {}
        for nm, a, b in zip(names, fb, wb):
            e = acc.setdefault(nm, [0, 0.0])
            e[0] += 1
            e[1] += a + b
        per_kernel = {k: int(v[1] / v[0]) for k, v in acc.items()}
except OSError:
    pass
out = {
    "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) on `python bench.py --steps 20 "
              "--warmup 5 --timed-only --no-graph --tune-cache <the selection of the untraced run>`, last forward pass, "
              "conv/fc kernels only (scripts/profile_r02.sh)",
    "fetch_size_KB_raw": f["sum_conv_KB"], "write_size_KB": w["sum_conv_KB"], "launches": n,
    "correction": "gfx950 FETCH_SIZE counts 128-B requests as 64 B for wide (16 B/lane) reads: doubled "
                  "(MI355X_MICROARCH.md section HBM); WRITE_SIZE taken as reported",
    "hbm_bytes_per_forward": hbm, "hbm_bytes_per_launch": hbm // n, "batch": batch,
    "src_sha": L.source_sha(),
    "graph": graph,
    "per_kernel": per_kernel,      # HBM bytes per launch (2 x FETCH_SIZE + WRITE_SIZE), averaged per kernel function of the pass
}
json.dump(out, open(os.path.join(d, "traffic.json"), "w"), indent=1)
print(json.dumps(out))
