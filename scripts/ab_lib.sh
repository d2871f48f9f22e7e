#!/bin/bash
# Same-box A/B of two builds of the HIP library: ab/lib_old.so vs ab/lib_new.so (SABER_MI355X_LIB override),
# alternating `bench.py --timed-only` runs. Usage (on the GPU box): bash scripts/ab_lib.sh [rounds] [extra bench flags]
R=${1:-3}; shift
mkdir -p gpurun_out/ab
for r in $(seq 1 $R); do
  for v in old new; do
    SABER_MI355X_LIB=$PWD/ab/lib_$v.so timeout 200 python bench.py --steps 300 --timed-only "$@" 2>/dev/null | sed "s/^/$v $r /"
  done
done | tee gpurun_out/ab/ab.log
