#!/bin/bash
# SQ counters of the recognition kernels of ONE 16-page request, exact or relaxed numerics (two --pmc passes).
# Usage (GPU box, repo root): tools/pmc_relaxed.sh <exact|relaxed> <tag>
set -u
MODE=${1:-relaxed}
TAG=${2:-$MODE}
ROOT=$PWD
OUT=$ROOT/gpurun_out/sq_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/tools/relaxed_report.py --once $MODE"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT -d $OUT -o p1 -- $CMD > $OUT/p1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM SQ_INSTS_SALU -d $OUT -o p2 -- $CMD > $OUT/p2.log 2>&1
python $ROOT/tools/pmc_dump.py $OUT/p1_results.db $OUT/p2_results.db --match conv3x3 > $OUT/summary.txt 2>&1
rm -f $OUT/*.db
head -120 $OUT/summary.txt
