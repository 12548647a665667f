#!/bin/bash
# default bench with option det_rows = 8 / 14 / 0 / 1 / 8 / 0 (rows per workgroup forced small: shorter-lived workgroups beside the conv stacks)
export TMPDIR=/tmp
O=gpurun_out/drows_ab; mkdir -p $O
for m in 8 14 0 1 8 0; do
  OCRS_DET_ROWS=$m timeout 300 python bench.py --no-cpu-baseline --no-extras > $O/bench2_$m.json 2> $O/err.txt
  python - $O/bench2_$m.json $m <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
print("det_rows=%s: %.1f pages/s, %.2f ms/step, conv live %.3f" % (sys.argv[2], d["value"], d["ms_per_step"], d["roofline"]["frac"]))
PY
done
