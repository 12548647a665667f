#!/bin/bash
# rows-per-wave sweep of the streaming detection blocks (option det_stream = 8 / 14 / 32; 0 = LDS-tiled blocks)
mkdir -p gpurun_out/dstream
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for S in 0 8 14 32; do
  OCRS_DET_STREAM=$S timeout 120 python tools/det_bench.py 40 2>&1 | tail -1 | sed "s/^/S=$S /"
  OCRS_DET_STREAM=$S timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/dstream/s$S -o det -- python tools/det_bench.py 20 > gpurun_out/dstream/s$S.log 2>&1
done
