#!/bin/bash
# Builds libnsk with the per-task clock trace of the multi-pass Poisson kernel compiled in (into tools/dbg) --
# run here, where hipcc is; then on the GPU box:  NSK_LIB=$PWD/tools/dbg/libnsk_ptrace.so python tools/trace_poisson.py --k 6
set -e
cd "$(dirname "$0")/../navierstokes_amd/csrc"
make -s -j8
mkdir -p ../../tools/dbg
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-value -mllvm -amdgpu-sched-strategy=max-ilp"
K=${1:-6}
/opt/rocm/bin/hipcc $F -DNSK_K=$K -DNSK_POISSON_TRACE=4096 -c ns_poisson_inst.hip -o /tmp/ns_poisson_trace_k$K.o
OBJS=$(ls *.o | grep -v ns_poisson_k$K.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/dbg/libnsk_ptrace.so /tmp/ns_poisson_trace_k$K.o $OBJS -ldl
echo built tools/dbg/libnsk_ptrace.so
