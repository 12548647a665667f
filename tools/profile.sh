#!/bin/bash
# rocprofv3 passes behind profiles/: kernel trace, then HBM traffic counters in separate
# --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit one pass: MI355X_MICROARCH.md "rocprofv3 PMC slots").
# Usage (on the GPU box, from the repo root): tools/profile.sh <tag>
set -u
TAG=${1:-r1}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $OLDPWD/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-kernel-timing --no-pipeline --settle-s 0"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT -o trace -- $BENCH > $OUT/trace.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT -o fetch -- $BENCH > $OUT/fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT -o write -- $BENCH > $OUT/write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE -d $OUT -o mfma -- $BENCH > $OUT/mfma.log 2>&1
# the default bench command (4 steps in flight) under the kernel trace, for the duration cross-check
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT -o default -- python $OLDPWD/bench.py --no-cpu-baseline --no-extras > $OUT/default.log 2>&1
ls -la $OUT
