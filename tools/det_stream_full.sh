#!/bin/bash
# Full GPU suite + default bench ABAB for option det_stream (1 vs 0)
export TMPDIR=/tmp
O=gpurun_out/dstream_full; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/test_gpu_all.log 2>&1; echo "tests rc=$?"; tail -3 $O/test_gpu_all.log
for m in 1 0 1 0; do
  OCRS_DET_STREAM=$m timeout 300 python bench.py --no-cpu-baseline > $O/bench_ds${m}_$RANDOM.json 2> $O/err.txt
  f=$(ls -t $O/bench_ds${m}_*.json | head -1)
  python - "$f" $m <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
e=d.get("extras",{}); sp=e.get("single_page_api",{})
print("det_stream=%s: %.1f pages/s | det-only %s / %s | one page alone %s ms, 12 threads %s | det roofline frac %s, %s ms per 8 pages" % (
  sys.argv[2], d["value"], e.get("detection_only_pages_per_s_one_request_at_a_time"), e.get("detection_only_pages_per_s"),
  sp.get("one_page_alone_ms"), sp.get("pages_per_s"), d.get("roofline_detection",{}).get("frac"), d.get("roofline_detection",{}).get("ms_per_8_pages")))
PY
done
