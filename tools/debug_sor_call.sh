#!/bin/bash
# Inlined against called form of sor_pipe inside the merged code3 step: both debug builds print every
# judged sweep (mode, step counter, sweep, word index, bit pattern of max|old-new|, LDS address).
# Needs tools/dbg/libnsk_dbg_inl.so and libnsk_dbg_noinl.so (bubble_kernels.hip built with
# -DNSK_SOR_DEBUG and, for the second, -DNSK_SOR_NOINLINE, linked with the other objects of libnsk).
set -u
mkdir -p gpurun_out
NSK_LIB=$PWD/tools/dbg/libnsk_dbg_inl.so timeout -k 10 120 python -u tools/debug_sor_call.py 2 > gpurun_out/sor_dbg_inl.log 2>&1
echo "inlined build: exit $?"
NSK_LIB=$PWD/tools/dbg/libnsk_dbg_noinl.so timeout -k 10 120 python -u tools/debug_sor_call.py 2 > gpurun_out/sor_dbg_noinl.log 2>&1
echo "called build: exit $?"
grep STEP gpurun_out/sor_dbg_inl.log gpurun_out/sor_dbg_noinl.log
