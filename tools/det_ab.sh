#!/bin/bash
# A/B of a detection option on the GPU box (the runs behind DESIGN.md §6.2.2 / §6.2.3 / §7 and profiles/r4_det_stream_experiments.txt):
#   tools/det_ab.sh <ENV_NAME> <mode> <value> [<value> ...]      e.g.  tools/det_ab.sh OCRS_DET_ROWS loop 1 0 1 0
# modes:  loop  = detection-only loop (tools/det_bench.py: 8 pages per request, one request at a time)
#         prof  = the same under rocprofv3 --kernel-trace --stats (per-kernel times into gpurun_out/det_ab/<value>.txt)
#         bench = the default bench (16-page requests, 5 in flight) with extras
# ENV_NAME: OCRS_DET_STREAM (0 | 1 | 8 | 14 | 32), OCRS_DET_ROWS (0 | 1 | 8 | 14 | 20 | 32), OCRS_CCL_QUAD (0 | 1), ...
export TMPDIR=/tmp
NAME=$1; MODE=$2; shift 2
O=gpurun_out/det_ab; mkdir -p $O
for v in "$@"; do
  case $MODE in
  loop) env $NAME=$v timeout 120 python tools/det_bench.py 60 2>&1 | tail -1 | sed "s/^/$NAME=$v /";;
  prof)
    (cd /tmp && env $NAME=$v timeout 300 rocprofv3 --kernel-trace --stats -d $OLDPWD/$O/p$v -o det -- python $OLDPWD/tools/det_bench.py 20 > $OLDPWD/$O/p$v.log 2>&1)
    python tools/rocprof_summary.py $O/p$v/det_results.db $O/$v.txt > /dev/null; echo "== $NAME=$v"; head -16 $O/$v.txt | cut -c1-150;;
  bench)
    env $NAME=$v timeout 300 python bench.py --no-cpu-baseline > $O/bench_$v.json 2> $O/err.txt
    python - $O/bench_$v.json "$NAME=$v" <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
e=d.get("extras",{}); sp=e.get("single_page_api",{}); r=d.get("roofline_detection",{})
print("%s: %.1f pages/s | det-only %s / %s | one page alone %s ms, 12 threads %s | detection CNN %s ms per 8 pages, frac %s" % (
  sys.argv[2], d["value"], e.get("detection_only_pages_per_s_one_request_at_a_time"), e.get("detection_only_pages_per_s"),
  sp.get("one_page_alone_ms"), sp.get("pages_per_s"), r.get("ms_per_8_pages"), r.get("frac")))
PY
    ;;
  esac
done
