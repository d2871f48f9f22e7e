#!/bin/bash
# The HOST side of the C-ABI library under AddressSanitizer (CPU only: the GPU pool has no sanitizers). The api_*.hip translation units - op-list
# construction, saber_hip_net_optimize's rewrites, capture, finalize, the arena's lifetime aliasing, selection bookkeeping - are rebuilt with
# -fsanitize=address -fno-gpu-sanitize (device code untouched), linked with the product's other objects into /tmp/saber_asan/libsaber_mi355x.so, and the
# reference's own framework (integration/_build/test_net_mi355x.bin) runs Graph::load -> Optimize -> Net<MI355X>::init with the captured plan on the
# malloc-backed mock HIP runtime against it: ResNet50 INT8 (set calls and calibrator files), ResNet101 INT8, ResNet50 FP32, VGG16 FP32, the Worker with
# three pool threads, the eight-device mode. The string / memcmp INTERCEPTORS are off: with them the first report is the reference's own use-after-free in
# GraphBase::remove_byio (framework/graph/graph_base.inl:218-240) - its freed node names compared through memcmp - and the run ends there;
# the instrumented loads and stores of the library and the memcpy / memset checks stay on. Usage: bash scripts/host_asan.sh [outfile]
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-/tmp/saber_asan/report.txt}
SAN=${2:-address}      # address | undefined (UndefinedBehaviorSanitizer: host code only, the device compilation ignores the option)
D=/tmp/saber_${SAN}
mkdir -p $D/obj
if [ $SAN = address ]; then
  RT=$(find /opt/rocm/lib/llvm -name "libclang_rt.asan-x86_64.so" | head -1)
  SF="-fsanitize=address -fno-gpu-sanitize -shared-libsan"
else
  RT=$(find /opt/rocm/lib/llvm -name "libclang_rt.ubsan_standalone-x86_64.so" | head -1)
  SF="-fsanitize=undefined -fno-sanitize-recover=undefined -shared-libsan -Wno-option-ignored"
fi
FLAGS="--offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-result -mllvm -amdgpu-mfma-vgpr-form $SF"
pids=()
for f in $ROOT/anakin_amd/csrc/api_*.hip; do
  o=$D/obj/$(basename ${f%.hip}).o
  if [ ! -f $o ] || [ $f -nt $o ] || [ $ROOT/anakin_amd/csrc/api_internal.h -nt $o ]; then /opt/rocm/bin/hipcc $FLAGS -c $f -o $o & pids+=($!); fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait $p; done
OBJS=$(ls $ROOT/anakin_amd/build/*.o | grep -v "/api_")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared $SF -o $D/libsaber_mi355x.so $D/obj/api_*.o $OBJS || exit 1
cd $ROOT
: > $OUT
echo "# host side of libsaber_mi355x.so (api_*.hip) under -fsanitize=$SAN on the mock HIP runtime; $(date -u +%F)" >> $OUT
SABER_ASAN_LIBDIR=$D SABER_ASAN_RT=$RT python - >> $OUT 2>&1 <<'PY'
import os, subprocess, sys, tempfile
sys.path.insert(0, os.getcwd())
from anakin_amd import workloads as W
from integration import net_model as NM
BIN = os.path.abspath("integration/_build/test_net_mi355x.bin")
MOCK = os.path.abspath("integration/_build/libmock_hip.so")
env = dict(os.environ, LD_PRELOAD=os.environ["SABER_ASAN_RT"] + " " + MOCK, LD_LIBRARY_PATH=os.environ["SABER_ASAN_LIBDIR"] + ":" + os.environ.get("LD_LIBRARY_PATH", ""),
           UBSAN_OPTIONS="halt_on_error=1:exitcode=98:print_stacktrace=1", ASAN_OPTIONS="detect_leaks=0:exitcode=97:abort_on_error=0:detect_odr_violation=0:replace_str=0:intercept_memcmp=0:intercept_strlen=0:intercept_strchr=0:intercept_strstr=0:intercept_strcmp=0", SABER_MI355X_NET_PLAN_TUNE="0")
bad = 0
def run(tag, name, precision, batch, mode, cal=False, rename=None, extra_env=None):
    global bad
    with tempfile.TemporaryDirectory() as d:
        model = W.build_model(name)
        x = W.make_input(batch)
        scales = W.calibrate(model, x[:2]) if precision == "int8" else {}
        mt, wb = NM.write_model(model, scales, batch, d, precision, calibrator_config=cal, rename=rename)
        x.tofile(os.path.join(d, "input.bin"))
        e = dict(env); e.update(extra_env or {})
        r = subprocess.run([BIN, mt, wb, os.path.join(d, "input.bin"), d] + mode, env=e, capture_output=True, text=True, errors="replace", cwd=d, timeout=1800)
        loaded = "saber_asan" in open("/proc/self/maps").read() if False else None
        san = any(k in r.stderr or k in r.stdout for k in ("AddressSanitizer", "runtime error:"))
        print("%-44s rc %d %s" % (tag, r.returncode, "SANITIZER REPORT" if san else "clean"), flush=True)
        if san or r.returncode != 0:
            bad += 1
            print((r.stderr[-4000:]))
run("resnet50 int8 b1 dry (SetOpPrec / SetVarScale)", "resnet50", "int8", 1, ["dry"])
run("resnet50 int8 b8 dry (calibrator files)", "resnet50", "int8", 8, ["dry"], cal=True)
run("resnet101 int8 b2 dry (short names)", "resnet101", "int8", 2, ["dry"], rename=NM.short_names)
run("resnet50 fp32 b2 dry", "resnet50", "fp32", 2, ["dry"])
run("vgg16 fp32 b1 dry", "vgg16", "fp32", 1, ["dry"])
run("resnet50 int8 b8 worker, 3 pool threads", "resnet50", "int8", 8, ["worker", "3", "24"], cal=True)
run("resnet50 int8 b2 devices 8", "resnet50", "int8", 2, ["devices", "8"], cal=True, extra_env={"MOCK_HIP_DEVICES": "8"})
print("runs with a report or a failure: %d" % bad)
sys.exit(1 if bad else 0)
PY
rc=$?
# proof that the instrumented library is the one that ran
LD_LIBRARY_PATH=$D ldd $ROOT/integration/_build/test_net_mi355x.bin | grep saber_mi355x >> $OUT
nm -D $D/libsaber_mi355x.so | grep -c "__asan\|__ubsan" | sed 's/^/sanitizer symbols referenced by the library: /' >> $OUT
tail -12 $OUT
exit $rc
