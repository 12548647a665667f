#!/bin/bash
# Round-2 GPU session A: validate the persistent GRU kernel + bench-scale parity, A/B the two GRU modes.
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r2a
mkdir -p $OUT
echo "== new bench-scale tests" | tee $OUT/summary.txt
timeout 900 python -m pytest tests/test_gpu_bench_scale.py -x -q > $OUT/test_bench_scale.log 2>&1; echo "rc=$?" | tee -a $OUT/summary.txt
tail -5 $OUT/test_bench_scale.log | tee -a $OUT/summary.txt
echo "== bench A/B (16 pages/step, 6 in flight)" | tee -a $OUT/summary.txt
for mode in 0 1; do
  OCRS_GRU_MODE=$mode timeout 300 python bench.py --no-cpu-baseline --no-extras > $OUT/bench_default_gru$mode.json 2> $OUT/bench_default_gru$mode.err; echo "mode $mode rc=$?" | tee -a $OUT/summary.txt
  python - <<PY | tee -a $OUT/summary.txt
import json
try:
    d = json.loads(open("$OUT/bench_default_gru$mode.json").read().strip().splitlines()[-1])
    print("gru_mode=$mode pages/s", d["value"], "ms/step", d["ms_per_step"], "roof", d.get("roofline", {}).get("kernel"), d.get("roofline", {}).get("frac"))
    print("  kernels", {k: v for k, v in list(d.get("kernels_ms_per_step", {}).items())[:6]})
except Exception as e:
    print("parse failed", e)
PY
done
echo "== latency, 1 page, strictly serial" | tee -a $OUT/summary.txt
for mode in 0 1; do
  OCRS_GRU_MODE=$mode timeout 300 python bench.py --pages 1 --inflight 1 --no-pipeline --steps 20 --warmup 5 --settle-s 0 --no-cpu-baseline --no-extras --profile-hint > $OUT/bench_lat1_gru$mode.json 2> $OUT/bench_lat1_gru$mode.err; echo "mode $mode rc=$?" | tee -a $OUT/summary.txt
  python - <<PY | tee -a $OUT/summary.txt
import json
try:
    d = json.loads(open("$OUT/bench_lat1_gru$mode.json").read().strip().splitlines()[-1])
    print("gru_mode=$mode 1 page: ms/step", d["ms_per_step"], "stages", d.get("stages_ms_per_step"))
except Exception as e:
    print("parse failed", e)
PY
done
for mode in 0 1; do
  OCRS_GRU_MODE=$mode timeout 300 python bench.py --pages 16 --inflight 1 --no-pipeline --steps 6 --warmup 2 --settle-s 0 --no-cpu-baseline --no-extras --profile-hint > $OUT/bench_serial16_gru$mode.json 2> $OUT/bench_serial16_gru$mode.err; echo "serial16 mode $mode rc=$?" | tee -a $OUT/summary.txt
  grep -E "gemm_gru_hidden|stage rec_gru" $OUT/bench_serial16_gru$mode.err | tee -a $OUT/summary.txt
done
echo "== full GPU suite" | tee -a $OUT/summary.txt
timeout 1500 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_bench_scale.py > $OUT/test_gpu_all.log 2>&1; echo "rc=$?" | tee -a $OUT/summary.txt
tail -5 $OUT/test_gpu_all.log | tee -a $OUT/summary.txt
echo "== rocprof kernel trace of the default bench (persistent)" | tee -a $OUT/summary.txt
ROOT=$PWD
cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/prof -o default -- python $ROOT/bench.py --steps 18 --warmup 12 --no-cpu-baseline --no-extras > $ROOT/$OUT/prof_bench.json 2> $ROOT/$OUT/prof_bench.err; echo "rc=$?" | tee -a $ROOT/$OUT/summary.txt
cd $ROOT
db=$(find $OUT/prof -name "*.db" | head -1)
[ -n "$db" ] && python tools/rocprof_summary.py "$db" $OUT/r2a_default_bench_kernel_stats.txt && head -30 $OUT/r2a_default_bench_kernel_stats.txt | cut -c1-220 | tee -a $OUT/summary.txt
find $OUT/prof -size +30M -delete
echo done | tee -a $OUT/summary.txt
