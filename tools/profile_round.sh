#!/bin/bash
# One GPU call's worth of profiling for the round's committed summaries (run from the repo root
# on the GPU box): kernel trace of the bench command, then one PMC pass per counter group.
#   bash tools/profile_round.sh r02 6
set -e
tag=$1; k=$2
export TMPDIR=/tmp
out=gpurun_out/${tag}_prof
mkdir -p $out
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $out/kt -o kt -- python3 bench.py --steps 1 --warmup 1 --time-steps 0 --no-cpu > $out/kt.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $out/fetch -o fetch -- python3 tools/prof_poisson.py --k $k > $out/fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d $out/write -o write -- python3 tools/prof_poisson.py --k $k > $out/write.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES -d $out/sq -o sq -- python3 tools/prof_poisson.py --k $k > $out/sq.log 2>&1
find $out -name "*.db" | sort
