#!/bin/bash
# SQ stall/issue counters for the recognition kernels (two --pmc passes, 8 SQ slots each).
# Usage (GPU box, repo root): tools/pmc_sq.sh <tag>
set -u
TAG=${1:-sq}
OUT=$PWD/gpurun_out/sq_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $OLDPWD/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras --no-kernel-timing --no-pipeline --inflight 1 --settle-s 0"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT -d $OUT -o p1 -- $BENCH > $OUT/p1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM SQ_INSTS_SALU -d $OUT -o p2 -- $BENCH > $OUT/p2.log 2>&1
ls $OUT
python $OLDPWD/tools/pmc_dump.py $OUT/p1_results.db $OUT/p2_results.db > $OUT/summary.txt 2>&1
rm -f $OUT/*.db
