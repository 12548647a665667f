#!/bin/bash
# One gpurun call = one session: tools/gpu_session.sh <tag> <section>...   (outputs under gpurun_out/<tag>/)
# Sections: tests_new tests_r3 tests_all ab lat serial16 det detprof detpmc full multi group stream stream10k conc single prof pmc relaxedprof soak
set -u
export TMPDIR=/tmp
TAG=$1; shift
ROOT=$PWD
OUT=gpurun_out/$TAG
mkdir -p $OUT
S=$OUT/summary.txt
: > $S
say() { echo "$@" | tee -a $S; }
jsum() {  # file label
python - "$1" "$2" <<'PY' | tee -a $S
import json, sys
f, label = sys.argv[1], sys.argv[2]
try:
    d = json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][-1])
    rl = d.get("rooflines", {})
    if "request_latency_ms" in d and d["request_latency_ms"]: print("   request latency:", d["request_latency_ms"])
    print("%s: %.1f pages/s, %.2f ms/step, host cores %.2f | %s" % (label, d["value"], d["ms_per_step"], d.get("host_cpu_cores_busy_per_gpu", 0),
          ", ".join("%s %.3f (%.2f ms x %.1f)" % (k.replace("gemm_", "").replace("_mfma", ""), v["frac"], v["avg_launch_ms"], v["launches_per_step"]) for k, v in rl.items())))
    st = d.get("stages_ms_per_step")
    if st: print("   stages:", {k: round(v, 2) for k, v in st.items()})
    for k in ("roofline_detection", "value_incl_h2d", "cpu_baseline"):
        if k in d: print("   %s: %s" % (k, json.dumps(d[k])[:400]))
    if "extras" in d: print("   extras:", json.dumps({k: v for k, v in d["extras"].items() if "config" not in k})[:400])
except Exception as e:
    print(label, "parse failed:", e)
PY
}
for sec in "$@"; do
case $sec in
tests_new)
  say "== bench-scale tests"; timeout 900 python -m pytest tests/test_gpu_bench_scale.py -x -q > $OUT/test_bench_scale.log 2>&1; say "rc=$?"; tail -4 $OUT/test_bench_scale.log | tee -a $S;;
tests_r3)
  say "== round-3 tests (group, coalescing, capacity fall-backs, recurrence fall-backs)"; timeout 1200 python -m pytest tests/test_gpu_r3.py -x -q > $OUT/test_r3.log 2>&1; say "rc=$?"; tail -15 $OUT/test_r3.log | tee -a $S;;
tests_all)
  say "== full GPU suite"; timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/test_gpu_all.log 2>&1; say "rc=$?"; tail -4 $OUT/test_gpu_all.log | tee -a $S;;
ab)
  say "== default bench, GRU persistent (0) vs per-step (1)"
  for mode in 0 1 0 1; do
    OCRS_GRU_MODE=$mode timeout 300 python bench.py --no-cpu-baseline --no-extras > $OUT/bench_default_gru$mode.json 2> $OUT/bench_default_gru$mode.err; jsum $OUT/bench_default_gru$mode.json "gru_mode=$mode"
  done;;
lat)
  say "== latency: 1 page per request, strictly serial stages"
  for mode in 0 1; do
    OCRS_GRU_MODE=$mode timeout 300 python bench.py --pages 1 --inflight 1 --no-pipeline --steps 20 --warmup 5 --settle-s 0 --no-cpu-baseline --no-extras > $OUT/bench_lat1_gru$mode.json 2> $OUT/bench_lat1_gru$mode.err; jsum $OUT/bench_lat1_gru$mode.json "1 page serial gru_mode=$mode"
  done;;
serial16)
  say "== 16 pages per request, strictly serial (isolated kernel times)"
  for mode in 0 1; do
    OCRS_GRU_MODE=$mode timeout 300 python bench.py --pages 16 --inflight 1 --no-pipeline --steps 6 --warmup 2 --settle-s 0 --no-cpu-baseline --no-extras --profile-hint > $OUT/bench_serial16_gru$mode.json 2> $OUT/bench_serial16_gru$mode.err
    jsum $OUT/bench_serial16_gru$mode.json "16 pages serial gru_mode=$mode"; grep -E "^gemm_|^dwconv|^conv|^pool|^other|^logsoft" $OUT/bench_serial16_gru$mode.err | tee -a $S
  done;;
det)
  say "== detection stack: fused LDS-tiled DoubleConv blocks (1) vs unfused kernels (0); extras legs only"
  for f in 1 0; do
    OCRS_DET_FUSE=$f timeout 400 python bench.py --steps 6 --warmup 3 --settle-s 0 --no-cpu-baseline > $OUT/bench_det$f.json 2> $OUT/bench_det$f.err; jsum $OUT/bench_det$f.json "det_fuse=$f"
  done;;
detprof)
  say "== rocprofv3 kernel trace of the detection-only loop (tools/det_bench.py), fused vs unfused"
  for f in 1 0; do
    (cd /tmp && OCRS_DET_FUSE=$f timeout 300 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/detprof -o det$f -- python $ROOT/tools/det_bench.py 20 > $ROOT/$OUT/detprof$f.log 2>&1); tail -1 $OUT/detprof$f.log | tee -a $S
    db=$(find $OUT/detprof -name "det${f}*.db" | head -1)
    [ -n "$db" ] && python tools/rocprof_summary.py "$db" $OUT/${TAG}_det_fuse${f}_kernel_stats.txt > /dev/null && head -34 $OUT/${TAG}_det_fuse${f}_kernel_stats.txt | cut -c1-200 | tee -a $S
  done
  find $OUT/detprof -size +30M -delete;;
full)
  say "== default bench with every leg (extras, cpu baseline)"; timeout 900 python bench.py > $OUT/bench_full.json 2> $OUT/bench_full.err; say "rc=$?"; jsum $OUT/bench_full.json "default";;
multi)
  say "== 2 ranks on this one GPU (gloo, ranks share the device): exercises the self-spawn + gather path"
  OCRS_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --spawn-ranks --steps 8 --warmup 4 --pages 8 --settle-s 0 --no-cpu-baseline --no-extras > $OUT/bench_2rank.json 2> $OUT/bench_2rank.err; say "rc=$?"; jsum $OUT/bench_2rank.json "2 ranks/1 GPU"; tail -3 $OUT/bench_2rank.err | cut -c1-300 | tee -a $S
  say "== 2 ranks, RCCL backend on one GPU (may be refused by RCCL: informational)"
  timeout 300 python bench.py --gpus 2 --spawn-ranks --steps 4 --warmup 2 --pages 8 --settle-s 0 --no-cpu-baseline --no-extras > $OUT/bench_2rank_rccl.json 2> $OUT/bench_2rank_rccl.err; say "rc=$?"; jsum $OUT/bench_2rank_rccl.json "2 ranks RCCL"; tail -2 $OUT/bench_2rank_rccl.err | cut -c1-300 | tee -a $S;;
group)
  say "== engine group in ONE process: 2 members on this one GPU (devices 0,0; host gather), 8 pages per member per step"
  timeout 600 python bench.py --gpus 2 --devices 0,0 --steps 12 --warmup 6 --pages 8 --settle-s 0 > $OUT/bench_group00.json 2> $OUT/bench_group00.err; say "rc=$?"; jsum $OUT/bench_group00.json "group [0,0]"; tail -2 $OUT/bench_group00.err | cut -c1-300 | tee -a $S
  say "== engine group of one member with the RCCL gather (ncclCommInitAll on one device)"
  timeout 600 python bench.py --devices 0 --gather ${GROUP_GATHER:-rccl-final} --steps 12 --warmup 6 --settle-s 0 > $OUT/bench_group0_rccl.json 2> $OUT/bench_group0_rccl.err; say "rc=$?"; jsum $OUT/bench_group0_rccl.json "group [0] rccl"; tail -2 $OUT/bench_group0_rccl.err | cut -c1-300 | tee -a $S
  python - $OUT/bench_group0_rccl.json <<'PY' | tee -a $S
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1]); print("   ", d["config"]["parallelism"][:300])
except Exception as e: print("parse failed", e)
PY
  ;;
single)
  say "== the reference's call pattern: one page per call from N host threads; coalescing variants (env | inflight)"
  while IFS='|' read -r envs infl; do
    [ -z "$infl" ] && continue
    tag=$(echo "$envs$infl" | tr -c 'A-Za-z0-9' '_')
    eval "$envs"; timeout 300 python bench.py ${BENCH_EXTRA:-} --pages 1 --inflight $infl --steps 360 --warmup 36 --settle-s 1 --no-cpu-baseline --no-extras > $OUT/bench_single_$tag.json 2> $OUT/bench_single_$tag.err; rc=$?
    say "[$envs] inflight=$infl rc=$rc"; jsum $OUT/bench_single_$tag.json "[$envs] inflight=$infl"
    [ $rc -ne 0 ] && tail -2 $OUT/bench_single_$tag.err | cut -c1-300 | tee -a $S
    python - $OUT/bench_single_$tag.json <<'PY' | tee -a $S
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1]); print("    ", d["config"].get("coalesce", "")[-100:])
except Exception as e: print("parse failed", e)
PY
  done <<EOF_SINGLE
${SINGLE_CASES:-BENCH_EXTRA=--coalesce=2|12
BENCH_EXTRA=--coalesce=-1|12
BENCH_EXTRA=--coalesce=2|24}
EOF_SINGLE
  ;;
stream10k)
  say "== configs[4] as stated: 10 000 distinct pages on 1 GPU (31 GB resident)"; timeout 1500 python bench.py --stream-pages 10000 --warmup 4 --no-cpu-baseline --no-extras > $OUT/bench_stream10k.json 2> $OUT/bench_stream10k.err; say "rc=$?"; jsum $OUT/bench_stream10k.json "stream 10k"; tail -2 $OUT/bench_stream10k.err | cut -c1-300 | tee -a $S;;
detpmc)
  say "== PMC passes of the detection-only loop (FETCH_SIZE, WRITE_SIZE): bytes per 8-page request over every kernel"
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $ROOT/$OUT/detpmc -o fetch -- python $ROOT/tools/det_bench.py 10 > $ROOT/$OUT/detpmc_fetch.log 2>&1); say "fetch rc=$?"
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $ROOT/$OUT/detpmc -o write -- python $ROOT/tools/det_bench.py 10 > $ROOT/$OUT/detpmc_write.log 2>&1); say "write rc=$?"
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE -d $ROOT/$OUT/detpmc -o mfma -- python $ROOT/tools/det_bench.py 10 > $ROOT/$OUT/detpmc_mfma.log 2>&1); say "mfma rc=$?"
  python tools/pmc_summary.py $OUT/detpmc $OUT/${TAG}_det_pmc_hbm.txt $OUT/${TAG}_det_pmc.json 13 > /dev/null 2>$OUT/detpmc_summary.err; head -30 $OUT/${TAG}_det_pmc_hbm.txt | cut -c1-200 | tee -a $S
  find $OUT/detpmc -size +30M -delete;;
stream)
  say "== configs[4] stream mode on 1 GPU: 512 distinct pages"; timeout 900 python bench.py --stream-pages 512 --warmup 4 --no-cpu-baseline --no-extras > $OUT/bench_stream512.json 2> $OUT/bench_stream512.err; say "rc=$?"; jsum $OUT/bench_stream512.json "stream 512";;
prof)
  say "== rocprofv3 kernel trace, default bench"
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/prof -o default -- python $ROOT/bench.py --steps 18 --warmup 12 --no-cpu-baseline --no-extras > $ROOT/$OUT/prof_bench.json 2> $ROOT/$OUT/prof_bench.err); say "rc=$?"
  db=$(find $OUT/prof -name "default*.db" | head -1)
  [ -n "$db" ] && python tools/rocprof_summary.py "$db" $OUT/${TAG}_default_bench_kernel_stats.txt > /dev/null && head -16 $OUT/${TAG}_default_bench_kernel_stats.txt | cut -c1-200 | tee -a $S
  jsum $OUT/prof_bench.json "under rocprof"
  say "== rocprofv3 kernel trace, serial 16 pages (isolated kernels)"
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/prof -o serial -- python $ROOT/bench.py --steps 3 --warmup 1 --settle-s 0 --inflight 1 --no-pipeline --no-cpu-baseline --no-extras --no-kernel-timing > /dev/null 2> $ROOT/$OUT/prof_serial.err); say "rc=$?"
  db=$(find $OUT/prof -name "serial*.db" | head -1)
  [ -n "$db" ] && python tools/rocprof_summary.py "$db" $OUT/${TAG}_serial_kernel_stats.txt > /dev/null && head -24 $OUT/${TAG}_serial_kernel_stats.txt | cut -c1-200 | tee -a $S
  find $OUT/prof -size +30M -delete;;
pmc)
  say "== PMC passes (separate runs: FETCH_SIZE, WRITE_SIZE, MFMA busy), serial 16 pages"
  BENCH="python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-kernel-timing --no-pipeline --inflight 1 --settle-s 0"
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $ROOT/$OUT/pmc -o fetch -- $BENCH > $ROOT/$OUT/pmc_fetch.log 2>&1); say "fetch rc=$?"
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $ROOT/$OUT/pmc -o write -- $BENCH > $ROOT/$OUT/pmc_write.log 2>&1); say "write rc=$?"
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE -d $ROOT/$OUT/pmc -o mfma -- $BENCH > $ROOT/$OUT/pmc_mfma.log 2>&1); say "mfma rc=$?"
  python tools/pmc_summary.py $OUT/pmc $OUT/${TAG}_pmc_hbm.txt $OUT/${TAG}_pmc.json > /dev/null 2>$OUT/pmc_summary.err; head -40 $OUT/${TAG}_pmc_hbm.txt | cut -c1-220 | tee -a $S
  find $OUT/pmc -size +30M -delete;;
group8)
  say "== eight members on this one GPU (devices 0 x 8, 2 pages per member per step = the 16-page step), final gather through the librccl test double, against the single engine on the same box"
  STUB=$(python -c "import sys; sys.path.insert(0, 'tests'); import stub_util; print(stub_util.rccl_stub_path())")
  for rep in 1 2; do
    timeout 300 python bench.py --steps 36 --warmup 12 --no-cpu-baseline --no-extras > $OUT/bench_single_$rep.json 2> $OUT/bench_single_$rep.err; jsum $OUT/bench_single_$rep.json "single engine"
    OCRS_RCCL_LIB=$STUB timeout 300 python bench.py --gpus 8 --devices 0,0,0,0,0,0,0,0 --pages 2 --gather rccl-final --steps 36 --warmup 12 --no-cpu-baseline --no-extras > $OUT/bench_group8_$rep.json 2> $OUT/bench_group8_$rep.err; jsum $OUT/bench_group8_$rep.json "group 0x8"
  done
  python - $OUT/bench_group8_2.json <<'PY' | tee -a $S
import json, sys
d = json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
print("   members:", json.dumps(d.get("members"))[:1200]); print("   final_gather:", d.get("final_gather"))
PY
  ;;
relaxedprof)
  say "== rocprofv3 kernel trace, serial 16 pages, numerics relaxed and reduced (isolated kernels of the split matrix path)"
  for m in relaxed reduced; do
    (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/prof -o serial_$m -- python $ROOT/bench.py --numerics $m --steps 3 --warmup 1 --settle-s 0 --inflight 1 --no-pipeline --no-cpu-baseline --no-extras --no-kernel-timing > /dev/null 2> $ROOT/$OUT/prof_serial_$m.err); say "$m rc=$?"
    db=$(find $OUT/prof -name "serial_${m}*.db" | head -1)
    [ -n "$db" ] && python tools/rocprof_summary.py "$db" $OUT/${TAG}_${m}_serial_kernel_stats.txt > /dev/null && head -14 $OUT/${TAG}_${m}_serial_kernel_stats.txt | cut -c1-200 | tee -a $S
  done
  find $OUT/prof -size +30M -delete;;
conc)
  say "== small requests in flight: 1 and 2 pages per request, 6 and 12 in flight (gate-per-wave GRU kernel under concurrency)"
  for a in "--pages 1 --inflight 6" "--pages 1 --inflight 12" "--pages 2 --inflight 6"; do
    timeout 300 python bench.py $a --steps 120 --warmup 20 --settle-s 1 --no-cpu-baseline --no-extras > $OUT/bench_conc.json 2> $OUT/bench_conc.err; rc=$?
    say "$a rc=$rc"; jsum $OUT/bench_conc.json "$a"; [ $rc -ne 0 ] && tail -2 $OUT/bench_conc.err | cut -c1-300 | tee -a $S
  done;;
soak)
  say "== pages of varied sizes from six threads against the sequential run, memory caps in force: exact, relaxed, reduced"
  for m in exact relaxed reduced; do
    timeout 300 python tools/soak_varied.py ${SOAK_SECONDS:-20} 6 --numerics $m > $OUT/soak_$m.txt 2>&1; say "$m rc=$?"; tail -1 $OUT/soak_$m.txt | tee -a $S
  done;;
*) say "unknown section $sec";;
esac
done
say done
