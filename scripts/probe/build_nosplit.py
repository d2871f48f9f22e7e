"""Probe build of the C-ABI library with -DSABER_PROBE_NOSPLIT (conv_igemm_impl.h: split3_pair costs one instruction; results are wrong on purpose):
anakin_amd/build_probe/libsaber_mi355x_nosplit.so, loaded with SABER_MI355X_LIB=... for TIMING only (scripts/r06_calls.sh nosplit). Round-5 verdict item 2(i):
what would FP32 edges handed over as bf16 planes gain at most?"""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from anakin_amd import build as B  # noqa: E402

out = os.path.join(B.HERE, "build_probe")
os.makedirs(out, exist_ok=True)
hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
objs, procs = [], []
for src in B.SOURCES:
    obj = os.path.join(out, src.replace(".hip", ".o"))
    objs.append(obj)
    procs.append((src, subprocess.Popen([hipcc] + B.FLAGS + ["-DSABER_PROBE_NOSPLIT", "-c", os.path.join(B.CSRC, src), "-o", obj])))
    if len(procs) >= 8:
        for s, p in procs:
            assert p.wait() == 0, s
        procs = []
for s, p in procs:
    assert p.wait() == 0, s
lib = os.path.join(out, "libsaber_mi355x_nosplit.so")
subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-o", lib] + objs)
print(lib)
