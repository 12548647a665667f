#!/bin/bash
# A/B of the detection blocks: row-streaming kernels (det_stream 1) vs LDS-tiled blocks (0): parity tests, detection-only
# rate, per-kernel times.   gpurun -- 'bash tools/det_stream_ab.sh'
mkdir -p gpurun_out/dstream
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "detection_model_run_bit_exact or detect_text_pixels_and_words or edge_pages or detect_words_batch" 2>&1 | tail -5
for m in 1 0 1 0; do OCRS_DET_STREAM=$m timeout 120 python tools/det_bench.py 40 2>&1 | tail -1 | sed "s/^/det_stream=$m /"; done
for m in 1 0; do
  OCRS_DET_STREAM=$m timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/dstream/prof$m -o det -- python tools/det_bench.py 20 > gpurun_out/dstream/prof$m.log 2>&1
  f=$(ls gpurun_out/dstream/prof$m/*kernel_stats.csv 2>/dev/null | head -1)
  echo "== det_stream=$m kernel stats ($f)"; [ -n "$f" ] && head -14 "$f" | cut -c1-200
done
