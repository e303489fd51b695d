#!/bin/bash
# Builds libnsk with the phase trace of the one-launch Poisson passes compiled in (into tools/dbg) -- run here,
# where hipcc is; then on the GPU box:  NSK_LIB=$PWD/tools/dbg/libnsk_smalltrace.so python tools/trace_small.py
set -e
cd "$(dirname "$0")/../navierstokes_amd/csrc"
make -s -j8
mkdir -p ../../tools/dbg
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-value"
/opt/rocm/bin/hipcc $F -DNSK_SMALL_TRACE -c ns_poisson_small.hip -o /tmp/ns_poisson_small_trace.o
OBJS=$(ls *.o | grep -v ns_poisson_small.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/dbg/libnsk_smalltrace.so /tmp/ns_poisson_small_trace.o $OBJS -ldl
echo built tools/dbg/libnsk_smalltrace.so
