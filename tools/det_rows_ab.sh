#!/bin/bash
# ABAB of option det_rows (1 vs 0): detection-only loop and the default bench
export TMPDIR=/tmp
O=gpurun_out/drows_ab; mkdir -p $O
for m in 1 0 1 0; do OCRS_DET_ROWS=$m timeout 120 python tools/det_bench.py 60 2>&1 | tail -1 | sed "s/^/det_rows=$m /"; done
for m in 1 0 1 0; do
  OCRS_DET_ROWS=$m timeout 300 python bench.py --no-cpu-baseline > $O/bench_$m.json 2> $O/err.txt
  python - $O/bench_$m.json $m <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
e=d.get("extras",{}); sp=e.get("single_page_api",{})
print("det_rows=%s: %.1f pages/s | det-only %s / %s | one page alone %s ms, 12 threads %s | det roofline frac %s, %s ms per 8 pages" % (
  sys.argv[2], d["value"], e.get("detection_only_pages_per_s_one_request_at_a_time"), e.get("detection_only_pages_per_s"),
  sp.get("one_page_alone_ms"), sp.get("pages_per_s"), d.get("roofline_detection",{}).get("frac"), d.get("roofline_detection",{}).get("ms_per_8_pages")))
PY
done
