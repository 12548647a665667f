#!/bin/bash
# detection-only loop: rate + per-kernel times (rocprofv3), one configuration.   gpurun -- 'bash tools/det_prof_once.sh <tag>'
T=${1:-x}; mkdir -p gpurun_out/dprof
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for i in 1 2; do timeout 120 python tools/det_bench.py 40 2>&1 | tail -1; done
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/dprof/$T -o det -- python tools/det_bench.py 20 > gpurun_out/dprof/$T.log 2>&1
python tools/rocprof_summary.py gpurun_out/dprof/$T/det_results.db gpurun_out/dprof/$T.txt > /dev/null; head -12 gpurun_out/dprof/$T.txt | cut -c1-150
