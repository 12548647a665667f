#!/bin/bash
# default bench with option det_stream = 1 (rule: 32 rows per wave for the decoder block at 16 pages) / 8 / 14 (shorter-lived waves)
export TMPDIR=/tmp
O=gpurun_out/dstream_ab3; mkdir -p $O
for m in 1 8 14 1 8 14; do
  OCRS_DET_STREAM=$m timeout 300 python bench.py --no-cpu-baseline --no-extras > $O/bench_$m.json 2> $O/err.txt
  python - $O/bench_$m.json $m <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
print("det_stream=%s: %.1f pages/s, %.2f ms/step, conv live %.3f" % (sys.argv[2], d["value"], d["ms_per_step"], d["roofline"]["frac"]))
PY
done
