#!/bin/bash
# Builds integration/_build/test_saber_conv_mi355x.bin: the patched copy of the reference's Saber library with the MI355X
# target (integration/apply_mi355x_target.py) + integration/test_saber_conv_mi355x.cpp, linked to the C-ABI HIP library
# and the oracle library. Needs /root/reference (build container only); the binary travels to the GPU box.
set -e
HERE=$(cd "$(dirname "$0")" && pwd)
ROOT=$(dirname "$HERE")
REF=${REF:-/root/reference}
B=$HERE/_build
[ -d "$REF/saber" ] || { echo "integration: $REF not present - MI355X target build skipped"; exit 0; }
OUT=$B/test_saber_conv_mi355x.bin
if [ -f "$OUT" ] && [ -z "$(find "$HERE" -maxdepth 3 -newer "$OUT" \( -name '*.h' -o -name '*.cpp' -o -name '*.py' -o -name '*.sh' \) | grep -v _build)" ] \
   && [ ! "$ROOT/anakin_amd/libsaber_mi355x.so" -nt "$OUT" ] && [ ! "$ROOT/include/saber_hip.h" -nt "$OUT" ]; then
  echo "integration: $OUT up to date"; exit 0
fi
python "$HERE/apply_mi355x_target.py" "$REF" "$B/anakin" > /dev/null
A=$B/anakin
INC="-I/opt/rocm/include -I$HERE/mi355x -I$A -I$A/utils -I$A/utils/logger -I$A/saber -I$A/saber/core -I$A/saber/funcs -I$ROOT/include -I$HERE"
# g++ (the reference's headers rely on gcc's lazy template checking); the HIP runtime API is plain C.
# The x86 MKL-DNN vender conv header (needs mkldnn.hpp, absent here) is guarded out: the X86 target is only the HOST side
# of this test (Tensor<X86>), its conv implementations are never instantiated.
CXX="g++ -std=c++14 -O2 -fopenmp -w -D__HIP_PLATFORM_AMD__ -DANAKIN_SABER_FUNCS_IMPL_X86_VENDER_CONV_H -include immintrin.h -include math.h"
mkdir -p $B/obj
pids=()
for f in $A/saber/core/impl/mi355x/mi355x_impl.cpp $A/saber/core/impl/x86/x86_impl.cpp $A/saber/core/impl/x86/x86_device.cpp \
         $A/saber/core/tensor_op.cpp $HERE/test_saber_conv_mi355x.cpp; do
  $CXX $INC -c $f -o $B/obj/$(basename ${f%.cpp}).o & pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
g++ -fopenmp -o $OUT $B/obj/*.o -L$ROOT/anakin_amd -lsaber_mi355x -L$ROOT/oracle -lsaber_oracle -L/opt/rocm/lib -lamdhip64 \
  -Wl,-rpath,'$ORIGIN/../../anakin_amd' -Wl,-rpath,'$ORIGIN/../../oracle' -Wl,-rpath,/opt/rocm/lib
echo "integration: built $OUT"
