"""Builds anakin_amd/libsaber_mi355x.so (the C-ABI HIP library) in-tree for gfx950.

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting .so travels
to the GPU box with the repo snapshot (it is git-ignored, not gpurun-ignored)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libsaber_mi355x.so")
SOURCES = ["conv_igemm.hip", "elementwise.hip", "fc_small.hip", "conv1x1_chain.hip", "conv_chain_coop.hip", "conv_stage_coop.hip", "stage_xcd.hip", "conv3x3_b3h.hip", "conv1x1_pw.hip", "conv1x1_pwk.hip", "conv_stem_f32.hip", "fc_f32_splitk.hip"] + \
    ["api_%s.hip" % n for n in ("conv", "autotune", "ops", "chain", "stage", "net", "net_optimize", "net_autotune", "capture", "gemm", "streams")] + \
    ["igemm_m%d_e%d.hip" % me for me in [(0, 0), (0, 1), (0, 2), (0, 3), (0, 4), (1, 0), (1, 1), (1, 2), (1, 3), (2, 3), (3, 3)]] + \
    ["igemm_dma_m%d_e%d.hip" % me for me in [(0, 0), (0, 1), (0, 2), (0, 3), (0, 4), (2, 3)]] + \
    ["halo_e%d.hip" % e for e in range(4)] + ["img_e%d.hip" % e for e in (0, 1, 3)] + ["stem_e%d.hip" % e for e in range(4)] + ["stem_pool.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-Wno-unused-result", "-mllvm", "-amdgpu-mfma-vgpr-form"]


def _stale(obj, deps):
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    import glob
    headers = sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [os.path.join(HERE, "..", "include", "saber_hip.h")]
    objs = []
    procs = []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(obj)
        deps = [h for h in headers if src.startswith("api_") or os.path.basename(h) != "api_internal.h"]
        if force or _stale(obj, [sp] + deps):
            cmd = [hipcc] + FLAGS + ["-c", sp, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed on " + src)
    if force or procs or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB


def build_cpp_tests(verbose=False):
    """tests/cpp/*.cpp -> tests/cpp/*.bin, linked against the HIP library and the oracle library (the test's checker)."""
    import glob
    root = os.path.dirname(HERE)
    build()
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "oracle"), "oracle"])
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    outs = []
    for src in sorted(glob.glob(os.path.join(root, "tests", "cpp", "*.cpp"))):
        out = src[:-4] + ".bin"
        deps = [src, os.path.join(root, "include", "saber_mi355x.hpp"), os.path.join(root, "include", "saber_mi355x_impl.h"), os.path.join(root, "include", "saber_hip.h"), LIB]
        if _stale(out, deps):
            cmd = [hipcc, "--offload-arch=gfx950", "-O2", "-std=c++17", "-I" + os.path.join(root, "include"), src, "-o", out,
                   "-L" + HERE, "-lsaber_mi355x", "-L" + os.path.join(root, "oracle"), "-lsaber_oracle",
                   "-Wl,-rpath,$ORIGIN/../../anakin_amd", "-Wl,-rpath,$ORIGIN/../../oracle", "-Wno-unused-result", "-mllvm", "-amdgpu-mfma-vgpr-form"]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
        outs.append(out)
    return outs


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
