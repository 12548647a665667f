#!/bin/bash
# Shader clock while a kernel runs: GRBM_GUI_ACTIVE (GPU cycles, busy) / kernel duration from the same trace.
set -u
OUT=$PWD/gpurun_out/clk
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $OLDPWD/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-pipeline --inflight 1 --settle-s 0"
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d $OUT -o clk -- $BENCH > $OUT/clk.log 2>&1
python $OLDPWD/tools/pmc_dump.py $OUT/clk_results.db > $OUT/summary.txt 2>&1
rm -f $OUT/*.db
