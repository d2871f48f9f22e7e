#!/bin/bash
# A second test_net binary whose Worker translation unit is compiled with the reference logger's OWN release switch (utils/logger/logger.h:19
# `#define LOGGER_SHUTDOWN 0` -> 1: LOG(...) compiles to nothing): Worker::sync_prediction logs 22 INFO lines per request (worker.cpp:101-105,
# 143-147: the first ten input and output floats, the thread id), each one formatted, written to stderr and flushed. Only worker.cpp's object
# differs; everything else is the objects of build_mi355x_test.sh. -> integration/_build/test_net_mi355x_quiet.bin
set -e
HERE=$(cd "$(dirname "$0")" && pwd)
ROOT=$(dirname "$HERE")
B=$HERE/_build
A=$B/anakin
[ -f $B/test_net_mi355x.bin ] || bash $HERE/build_mi355x_test.sh
[ -d $A ] || { echo "integration: no patched copy (build container only)"; exit 0; }
Q=$B/quiet_inc
mkdir -p $Q/utils/logger $B/obj_quiet
for f in logger.h logger_core.h log_utils.h; do cp $A/utils/logger/$f $Q/utils/logger/$f; done
sed -i 's/^#define LOGGER_SHUTDOWN 0/#define LOGGER_SHUTDOWN 1/' $Q/utils/logger/logger.h
grep -q "define LOGGER_SHUTDOWN 1" $Q/utils/logger/logger.h
INC="-I$Q -I$Q/utils -I$Q/utils/logger -I/opt/rocm/include -I$HERE/mi355x -I$A -I$A/utils -I$A/utils/logger -I$A/saber -I$A/saber/core -I$A/saber/funcs -I$A/framework -I$ROOT/include -I$HERE"
GUARDS="-DANAKIN_SABER_FUNCS_IMPL_X86_VENDER_CONV_H -DANAKIN_SABER_FUNCS_IMPL_X86_SABER_POOLING_H -DANAKIN_SABER_FUNCS_IMPL_X86_SABER_VENDER_FC_H"
g++ -std=c++14 -O2 -fopenmp -w -D__HIP_PLATFORM_AMD__ $GUARDS -include immintrin.h -include math.h $INC -c $A/framework/core/net/worker.cpp -o $B/obj_quiet/worker.o
OBJS=$(ls $B/obj_fw/*.o | grep -v "/worker.o")
COREOBJ=$(ls $B/obj/*.o | grep -v test_saber_conv)
LIBS="-L$ROOT/anakin_amd -lsaber_mi355x -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,\$ORIGIN/../../anakin_amd -Wl,-rpath,/opt/rocm/lib"
g++ -fopenmp -o $B/test_net_mi355x_quiet.bin $OBJS $B/obj_quiet/worker.o $COREOBJ $LIBS
echo "integration: built $B/test_net_mi355x_quiet.bin"
