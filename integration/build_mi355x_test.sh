#!/bin/bash
# Builds, under integration/_build/ (git-ignored; travels to the GPU box with the repo snapshot):
#   anakin/                       the PATCHED COPY of the reference's saber/ + utils/ + framework/ trees with the MI355X target
#                                 (integration/apply_mi355x_target.py; nothing of it is stored in this repository)
#   test_saber_conv_mi355x.bin    Conv / ConvEltwise<MI355X> under the reference's BaseFunc (integration/test_saber_conv_mi355x.cpp)
#   test_net_mi355x.bin           Graph<MI355X> (AddOp) -> Optimize() -> Net<MI355X>::init / prediction: the reference's own
#                                 framework compiled against the target (integration/test_net_mi355x.cpp)
#   libmock_hip.so                malloc-backed HIP stand-in for CPU dry runs of Net::init (integration/mock_hip/)
# Needs /root/reference (build container only). g++: the reference's headers rely on gcc's lazy template checking; the HIP
# runtime API is plain C. The x86 Saber impl headers that need xbyak / mkl-dnn / MKL (absent here) are guarded out by
# pre-defining their include guards: X86 is only the HOST side of these tests (Tensor<X86>, PBlock host mirrors), its
# operator implementations are never instantiated.
set -e
HERE=$(cd "$(dirname "$0")" && pwd)
ROOT=$(dirname "$HERE")
REF=${REF:-/root/reference}
B=$HERE/_build
[ -d "$REF/saber" ] || { echo "integration: $REF not present - MI355X target build skipped"; exit 0; }
OUT1=$B/test_saber_conv_mi355x.bin
OUT2=$B/test_net_mi355x.bin
MOCK=$B/libmock_hip.so
if [ -f "$OUT1" ] && [ -f "$OUT2" ] && [ -f "$MOCK" ] \
   && [ -z "$(find "$HERE" -maxdepth 3 -newer "$OUT2" \( -name '*.h' -o -name '*.cpp' -o -name '*.py' -o -name '*.sh' \) | grep -v _build)" ] \
   && [ ! "$ROOT/anakin_amd/libsaber_mi355x.so" -nt "$OUT2" ] && [ ! "$ROOT/include/saber_hip.h" -nt "$OUT2" ]; then
  echo "integration: $B up to date"; exit 0
fi
python "$HERE/apply_mi355x_target.py" "$REF" "$B/anakin" > /dev/null
A=$B/anakin
INC="-I/opt/rocm/include -I$HERE/mi355x -I$A -I$A/utils -I$A/utils/logger -I$A/saber -I$A/saber/core -I$A/saber/funcs -I$A/framework -I$ROOT/include -I$HERE"
GUARDS="-DANAKIN_SABER_FUNCS_IMPL_X86_VENDER_CONV_H -DANAKIN_SABER_FUNCS_IMPL_X86_SABER_POOLING_H -DANAKIN_SABER_FUNCS_IMPL_X86_SABER_VENDER_FC_H"
CXX="g++ -std=c++14 -O2 -fopenmp -w -D__HIP_PLATFORM_AMD__ $GUARDS -include immintrin.h -include math.h"
mkdir -p $B/obj $B/obj_fw
CORE="$A/saber/core/impl/mi355x/mi355x_impl.cpp $A/saber/core/impl/x86/x86_impl.cpp $A/saber/core/impl/x86/x86_device.cpp $A/saber/core/tensor_op.cpp"
F=$A/framework
FW="$F/graph/graph.cpp $F/graph/node.cpp $F/graph/llvm/scheduler.cpp $F/graph/llvm/virtual_graph.cpp
    $F/graph/llvm/fusion/fusion_op_register.cpp $F/graph/llvm/fusion/graph_pattern.cpp
    $F/graph/llvm/optimizer/conv_elewise_fusion_scheduler.cpp $F/graph/llvm/optimizer/memory_scheduler.cpp
    $F/graph/llvm/optimizer/parall_scheduler.cpp $F/core/functor.cpp $F/core/singleton.cpp $F/core/net/net.cpp
    $F/core/net/operator_func.cpp $F/core/net/calibrator_parse.cpp $F/core/net/calibrator_factory.cpp
    $F/core/net/auto_layout_config.cpp $F/core/net/worker.cpp $F/core/net/entropy_calibrator.cpp $F/core/net/batch_stream.cpp $F/core/operator/operator.cpp $F/core/operator/operator_attr.cpp
    $F/core/operator/operator_help.cpp $F/model_parser/parser/parser.cpp $F/model_parser/parser/model_io.cpp $F/utils/parameter_fusion.cpp $F/utils/data_common.cpp"
OPS=$(python - <<PY
import sys
sys.path.insert(0, "$HERE")
import apply_mi355x_target as a
print(" ".join("$F/operators/%s.cpp" % o for o in a.OPERATORS))
PY
)
# an object is rebuilt when its source changed (the apply script keeps the mtime of files whose content did not change)
# or when any header did; at most $(nproc) compilers at a time
NEWHDR=""
[ -f $B/obj_fw/.stamp ] && NEWHDR=$(find $A $HERE $ROOT/include -name '*.h' -newer $B/obj_fw/.stamp -not -path '*/_build/obj*' | head -1)
[ -f $B/obj_fw/.stamp ] || NEWHDR=all
compile() {  # src obj
  if [ -n "$NEWHDR" ] || [ ! -f "$2" ] || [ "$1" -nt "$2" ]; then $CXX $INC -c "$1" -o "$2" || { echo "integration: FAILED $1"; exit 1; }; fi
}
pids=()
for f in $CORE; do compile $f $B/obj/$(basename ${f%.cpp}).o & pids+=($!); done
compile $HERE/test_saber_conv_mi355x.cpp $B/obj/test_saber_conv_mi355x.o & pids+=($!)
for f in $FW; do compile $f $B/obj_fw/$(basename ${f%.cpp}).o & pids+=($!); done
for f in $OPS; do compile $f $B/obj_fw/op_$(basename ${f%.cpp}).o & pids+=($!); done
compile $HERE/test_net_mi355x.cpp $B/obj_fw/test_net_mi355x.o & pids+=($!)
for p in "${pids[@]}"; do wait $p; done
touch $B/obj_fw/.stamp
LIBS="-L$ROOT/anakin_amd -lsaber_mi355x -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,\$ORIGIN/../../anakin_amd -Wl,-rpath,/opt/rocm/lib"
g++ -fopenmp -o $OUT1 $B/obj/*.o $LIBS -L$ROOT/oracle -lsaber_oracle -Wl,-rpath,'$ORIGIN/../../oracle'
COREOBJ=$(ls $B/obj/*.o | grep -v test_saber_conv)
g++ -fopenmp -o $OUT2 $B/obj_fw/*.o $COREOBJ $LIBS
g++ -O1 -shared -fPIC -w -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include $HERE/mock_hip/mock_hip.cpp -o $MOCK
echo "integration: built $OUT1 $OUT2 $MOCK"
