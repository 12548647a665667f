#!/bin/bash
# rows-per-workgroup sweep of the row-streaming workgroup kernels (option det_rows = 8 / 14 / 20 / 32; 1 = rule; 0 = tiled)
mkdir -p gpurun_out/drows
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for S in 1 8 14 20 32 0; do
  OCRS_DET_ROWS=$S timeout 120 python tools/det_bench.py 40 2>&1 | tail -1 | sed "s/^/det_rows=$S /"
  OCRS_DET_ROWS=$S timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/drows/s$S -o det -- python tools/det_bench.py 20 > gpurun_out/drows/s$S.log 2>&1
  python tools/rocprof_summary.py gpurun_out/drows/s$S/det_results.db gpurun_out/drows/s$S.txt > /dev/null
done
