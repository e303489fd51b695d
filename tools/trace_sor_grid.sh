#!/bin/bash
# Builds libnsk with the phase trace of the tiled SOR compiled in (into tools/dbg, next to nothing else) --
# run here, where hipcc is; then on the GPU box:  NSK_LIB=$PWD/tools/dbg/libnsk_sortrace.so python tools/trace_sor_grid.py
set -e
cd "$(dirname "$0")/../navierstokes_amd/csrc"
make -s -j8
mkdir -p ../../tools/dbg
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-value"
/opt/rocm/bin/hipcc $F -DNSK_SOR_GRID_TRACE -c bubble_sor_grid.hip -o /tmp/bubble_sor_grid_trace.o
OBJS=$(ls *.o | grep -v bubble_sor_grid.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/dbg/libnsk_sortrace.so /tmp/bubble_sor_grid_trace.o $OBJS -ldl
echo built tools/dbg/libnsk_sortrace.so
