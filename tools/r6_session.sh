#!/bin/bash
# Round-6 GPU sessions: tools/r6_session.sh <tag> <section>...   (outputs under gpurun_out/<tag>/)
set -u
export TMPDIR=/tmp
TAG=$1; shift
OUT=gpurun_out/$TAG
ROOT=$PWD
mkdir -p $OUT
S=$OUT/summary.txt
: > $S
say() { echo "$@" | tee -a $S; }
bsum() {  # file label
python - "$1" "$2" <<'PY' | tee -a $S
import json, sys
f, label = sys.argv[1], sys.argv[2]
try:
    d = json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][-1])
    rl = d.get("rooflines", {})
    print("%s: %.1f pages/s (%.2f ms/step) incl fill/drain %s | %s" % (label, d["value"], d["ms_per_step"], d.get("value_incl_fill_drain"),
          ", ".join("%s %.3f/%.3f (%.2f ms x %.1f)" % (k.replace("gemm_", "").replace("_mfma", ""), v["frac"], v.get("frac_alone", 0), v["avg_launch_ms"], v["launches_per_step"]) for k, v in rl.items())))
    ex = d.get("extras", {})
    if "numerics" in ex: print("   numerics:", {k: (v.get("pages_per_s"), v.get("speedup")) for k, v in ex["numerics"].items() if isinstance(v, dict)})
    for k in ("value_resident", "value_from_pageable_host_pixels"):
        if k in d: print("   %s: %s" % (k, d[k]))
    if "roofline_detection" in d: print("   detection:", json.dumps(d["roofline_detection"])[:300])
    if "single_page_api" in ex: print("   single_page_api:", json.dumps(ex["single_page_api"])[:500])
    if "detection_only" in ex: print("   detection_only:", json.dumps(ex["detection_only"])[:300])
except Exception as e:
    print(label, "parse failed:", e)
PY
}
for sec in "$@"; do
case $sec in
tests)
  say "== full GPU suite"; timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/test_gpu_all.log 2>&1; say "rc=$?"; tail -6 $OUT/test_gpu_all.log | tee -a $S;;
tests_r6)
  say "== round-6 GPU tests, one by one (a hang costs one time-out, not the session)"
  for t in test_detection_batch_of_mixed_page_sizes test_group_replay test_creating_and_destroying test_relaxed_modes_canary test_exact_mode_canary; do
    timeout 240 python -m pytest tests/test_gpu_r6.py -x -q -k $t > $OUT/test_r6_$t.log 2>&1; say "$t rc=$?"; tail -3 $OUT/test_r6_$t.log | tee -a $S
  done;;
tests_old)
  say "== GPU suite without the round-6 file"; timeout 900 python -m pytest tests -m gpu -x -q --ignore=tests/test_gpu_r6.py > $OUT/test_gpu_old.log 2>&1; say "rc=$?"; tail -6 $OUT/test_gpu_old.log | tee -a $S;;
bench20)
  say "== the driver's command: --steps 20 --warmup 5"
  timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_form.json 2> $OUT/bench_driver_form.err; say "rc=$?"; bsum $OUT/bench_driver_form.json "driver form";;
bench72)
  say "== default bench (72 steps)"
  timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; say "rc=$?"; bsum $OUT/bench_default.json "default";;
benchq)
  say "== quick bench, no extras"
  timeout 600 python bench.py --no-extras --no-cpu-baseline > $OUT/bench_quick.json 2> $OUT/bench_quick.err; say "rc=$?"; bsum $OUT/bench_quick.json "quick";;
canary)
  say "== hazard canary (tools/hazard_canary.py): every stage, every element, six threads"
  for cfg in "relaxed none 12" "reduced none 12" "relaxed auto 8" "exact none 10"; do
    set -- $cfg
    timeout 150 python tools/hazard_canary.py --numerics $1 --isolation $2 --seconds $3 > $OUT/canary_$1_$2.json 2> $OUT/canary_$1_$2.err; rc=$?
    python - $OUT/canary_$1_$2.json "$cfg rc=$rc" <<'PY' | tee -a $S
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
    print(sys.argv[2], d["isolation"], "| mismatching checks", d["mismatching_checks"], {k: (v["checks"], v["bad_checks"], v["bad_elements"]) for k, v in d["classes"].items()},
          "| crop cols mod 64 nonzero:", [i for i, c in enumerate(d["crop_mismatch_columns_mod_64"] or []) if c], "| errors", d["errors"][:2])
except Exception as e:
    print(sys.argv[2], "parse failed", e)
PY
  done;;
repro)
  say "== stand-alone reproducer (tools/hazard_repro.hip): the real gemm_split_kernel beside twin launches of the real crop_lines_kernel"
  for a in "--aggressor none --seconds 4" "--aggressor split3 --seconds 12" "--aggressor split2 --seconds 12" "--aggressor exact --seconds 8" \
           "--aggressor split3 --seconds 12 --victim-streams 4" "--aggressor split3 --seconds 12 --lines 8 --victim-streams 4" \
           "--aggressor split3 --seconds 12 --split-cus 128 --victim-streams 4"; do
    timeout 120 tools/_build/hazard_repro $a 2>> $OUT/repro.err | tee -a $S
  done
  say "-- victim at s_setprio 3"
  timeout 120 tools/_build/hazard_repro.prio3 --aggressor split3 --seconds 12 --victim-streams 4 2>> $OUT/repro.err | tee -a $S;;
replay)
  say "== host-side pre-flight: 8 members on device 0, GPU shares replayed"
  for inf in 5 8; do
  timeout 600 python bench.py --devices 0,0,0,0,0,0,0,0 --replay --steps 40 --warmup 10 --inflight $inf --no-extras --no-cpu-baseline > $OUT/bench_replay8_inflight$inf.json 2> $OUT/bench_replay8.err; say "inflight $inf rc=$?"
  python - $OUT/bench_replay8_inflight$inf.json <<'PY' | tee -a $S
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
    print("replay x8: %.1f pages/s, %.2f ms/step, host cores busy %.2f of %s logical | replay %s" % (d["value"], d["ms_per_step"], d["host_cpu_cores_busy_per_gpu"], d.get("host_logical_cpus"), json.dumps(d["replay"]["share_seconds"])))
    print("   members:", [(m["member"], m["pages"], m["host_thread_cpu_s"], m["busy_wall_s"]) for m in d["members"]])
    print("   latency:", d["request_latency_ms"])
    print("   final gather:", d["final_gather"])
except Exception as e:
    print("parse failed:", e)
PY
  done
  tail -5 $OUT/bench_replay8.err | tee -a $S;;
repro3)
  say "== reproducer: exact-mode aggressor, long run, four victim streams (the claim that exact mode is immune, stand-alone)"
  timeout 200 tools/_build/hazard_repro --aggressor exact --seconds 60 --victim-streams 4 2>> $OUT/repro.err | tee -a $S
  timeout 100 tools/_build/hazard_repro.ACC_AGPR --aggressor split2 --seconds 10 --victim-streams 4 2>> $OUT/repro.err | tee -a $S
  timeout 100 tools/_build/hazard_repro.ACC_AGPR --aggressor split3 --seconds 10 --victim-streams 4 --analyse 40 2> $OUT/repro_agpr_analyse.txt | tee -a $S; head -12 $OUT/repro_agpr_analyse.txt | tee -a $S;;
soak)
  say "== varied-size one-page soak, six threads (mixed sizes now share detection batches)"
  timeout 300 python tools/soak_varied.py 20 6 > $OUT/soak_varied_exact.txt 2>&1; say "rc=$?"; tail -6 $OUT/soak_varied_exact.txt | tee -a $S;;
repro2)
  say "== reproducer, round two: where the wrong words come from; aggressor probes"
  timeout 120 tools/_build/hazard_repro --aggressor split3 --seconds 8 --analyse 3 2> $OUT/repro_analyse.txt | tee -a $S; head -45 $OUT/repro_analyse.txt | tee -a $S
  timeout 120 tools/_build/hazard_repro --aggressor split2 --seconds 6 --victim-streams 2 2>> $OUT/repro.err | tee -a $S
  say "-- accumulators in AGPRs"; timeout 120 tools/_build/hazard_repro.ACC_AGPR --aggressor split3 --seconds 8 --victim-streams 2 2>> $OUT/repro.err | tee -a $S
  say "-- the same registers through v_mfma_f32_16x16x32_bf16"; timeout 120 tools/_build/hazard_repro.MFMA16 --aggressor split3 --seconds 8 --victim-streams 2 2>> $OUT/repro.err | tee -a $S
  say "-- split-cus 64 / 192 / 224"
  for c in 64 192 224; do timeout 120 tools/_build/hazard_repro --aggressor split3 --seconds 6 --victim-streams 2 --split-cus $c 2>> $OUT/repro.err | tee -a $S; done;;
canary_agpr)
  say "== the product's split kernels with their accumulators in AGPRs (variant library), isolation NONE: what is left of the hazard"
  for n in relaxed reduced; do
    OCRS_AMD_LIB=$PWD/ocrs_amd/libocrs_amd.agpr.so timeout 150 python tools/hazard_canary.py --numerics $n --isolation none --seconds 12 > $OUT/canary_agpr_$n.json 2> $OUT/canary_agpr_$n.err; rc=$?
    python - $OUT/canary_agpr_$n.json "agpr $n none rc=$rc" <<'PY' | tee -a $S
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
    print(sys.argv[2], "| mismatching checks", d["mismatching_checks"], {k: (v["checks"], v["bad_checks"], v["bad_elements"]) for k, v in d["classes"].items()})
except Exception as e:
    print(sys.argv[2], "parse failed", e)
PY
  done;;
benchab)
  say "== the driver's form with and without per-launch HIP events in the timed window (ABAB)"
  for i in 1 2; do
    timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $OUT/bench_ab_timed$i.json 2> $OUT/bench_ab_timed$i.err; bsum $OUT/bench_ab_timed$i.json "events on  $i"
    timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-kernel-timing > $OUT/bench_ab_plain$i.json 2> $OUT/bench_ab_plain$i.err; bsum $OUT/bench_ab_plain$i.json "events off $i"
  done;;
logits)
  say "== relaxed engine, serial regime, log-probs only: what differs from the quiet run under load, and with which kernels"
  for opt in ${LOGITS_OPTS:-"" "gru_mode=1" "conv12_fuse=0" "gru_gates=0" "conv_flat=0"}; do
    o=""; [ -n "$opt" ] && o="--option $opt"
    timeout 200 python tools/hazard_canary.py --numerics ${LOGITS_MODE:-relaxed} --isolation auto --seconds ${LOGITS_SECONDS:-25} --classes logits $o > $OUT/logits_${opt:-default}.json 2> $OUT/logits_${opt:-default}.err; rc=$?
    python - $OUT/logits_${opt:-default}.json "relaxed auto logits ${opt:-default} rc=$rc" <<'PY' | tee -a $S
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
    print(sys.argv[2], {k: (v["checks"], v["bad_checks"], v["bad_elements"]) for k, v in d["classes"].items()}, "| lines per page", d["lines"], "| examples", json.dumps(d["examples"][:4]))
except Exception as e:
    print(sys.argv[2], "parse failed", e)
PY
  done;;
mubuf)
  say "== split::pipeline with MUBUF-form LDS-DMA (variant library): log-prob canary, then relaxed / reduced throughput ABAB against the stock library"
  OCRS_AMD_LIB=$PWD/ocrs_amd/libocrs_amd.mubuf.so timeout 200 python tools/hazard_canary.py --numerics relaxed --isolation auto --seconds 30 --classes logits,crop > $OUT/mubuf_canary.json 2> $OUT/mubuf_canary.err
  python - $OUT/mubuf_canary.json <<'PY' | tee -a $S
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
    print("mubuf canary relaxed auto", {k: (v["checks"], v["bad_checks"], v["bad_elements"]) for k, v in d["classes"].items()})
except Exception as e:
    print("parse failed", e)
PY
  for n in relaxed reduced; do for rep in 1 2; do
    timeout 300 python bench.py --numerics $n --steps 36 --warmup 12 --no-extras --no-cpu-baseline > $OUT/bench_${n}_stock$rep.json 2> $OUT/bench_${n}_stock$rep.err; bsum $OUT/bench_${n}_stock$rep.json "$n stock $rep"
    OCRS_AMD_LIB=$PWD/ocrs_amd/libocrs_amd.mubuf.so timeout 300 python bench.py --numerics $n --steps 36 --warmup 12 --no-extras --no-cpu-baseline > $OUT/bench_${n}_mubuf$rep.json 2> $OUT/bench_${n}_mubuf$rep.err; bsum $OUT/bench_${n}_mubuf$rep.json "$n mubuf $rep"
  done; done;;
wideab)
  say "== row-wise float4 epilogues (stock) vs the dword epilogues (variant libraries), ABAB, quick bench"
  for i in 1 2; do
    for v in stock narrow narrowconv; do
      lib=""; [ $v != stock ] && lib=$PWD/ocrs_amd/libocrs_amd.$v.so
      [ -n "$lib" ] && [ ! -f "$lib" ] && continue
      OCRS_AMD_LIB=$lib timeout 400 python bench.py --steps 30 --warmup 5 --no-extras --no-cpu-baseline > $OUT/bench_wide_$v$i.json 2> $OUT/bench_wide_$v$i.err; bsum $OUT/bench_wide_$v$i.json "$v $i"
    done
  done;;
splitab)
  say "== gemm_split row-wise epilogue (stock) vs the dword epilogue (variant), relaxed and reduced, ABAB"
  for i in 1 2; do
    for n in relaxed reduced; do
      for v in stock narrowsplit; do
        lib=""; [ $v != stock ] && lib=$PWD/ocrs_amd/libocrs_amd.$v.so
        OCRS_AMD_LIB=$lib timeout 400 python bench.py --numerics $n --steps 30 --warmup 5 --no-extras --no-cpu-baseline > $OUT/bench_split_${n}_$v$i.json 2> $OUT/bench_split_${n}_$v$i.err; bsum $OUT/bench_split_${n}_$v$i.json "$n $v $i"
      done
    done
  done;;
instab)
  say "== per-instance kernel durations, one request at a time, under rocprofv3 --kernel-trace: stock vs variant libraries ($INST_LIBS), ABAB"
  for i in 1 2; do
    for v in stock $INST_LIBS; do
      lib=""; [ $v != stock ] && lib=$ROOT/ocrs_amd/libocrs_amd.$v.so
      (cd /tmp && OCRS_AMD_LIB=$lib timeout 400 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/inst_$v$i -o serial -- python $ROOT/bench.py --steps 4 --warmup 1 --settle-s 0 --inflight 1 --no-pipeline --no-cpu-baseline --no-extras --no-kernel-timing ${INST_ARGS:-} > /dev/null 2> $ROOT/$OUT/inst_$v$i.err)
      db=$(find $OUT/inst_$v$i -name "serial*.db" | head -1)
      [ -n "$db" ] && python tools/rocprof_summary.py "$db" $OUT/inst_$v$i.txt > /dev/null && say "-- $v $i" && grep -E "conv3x3_ragged|conv12_fused|gemm_tiled_kernelILi128ELb0ELb1|gemm_split" $OUT/inst_$v$i.txt | cut -c1-140 | tee -a $S
      rm -rf $OUT/inst_$v$i
    done
  done;;
epiab)
  say "== end to end: row-wise epilogues (stock) vs the dword epilogues of rounds 1-5 (variant library), ABAB x 3, 40 steps"
  for i in 1 2 3; do
    for v in stock oldepi; do
      lib=""; [ $v != stock ] && lib=$PWD/ocrs_amd/libocrs_amd.$v.so
      OCRS_AMD_LIB=$lib timeout 400 python bench.py --steps 40 --warmup 5 --no-extras --no-cpu-baseline > $OUT/bench_epi_$v$i.json 2> $OUT/bench_epi_$v$i.err; bsum $OUT/bench_epi_$v$i.json "$v $i"
    done
  done;;
final)
  say "== the suite as the driver runs it, smoke, long canaries of the numerics modes"
  timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/test_gpu_all.log 2>&1; say "pytest -m gpu rc=$?"; tail -3 $OUT/test_gpu_all.log | tee -a $S
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; say "smoke rc=$?"; tail -1 $OUT/smoke.log | tee -a $S
  for n in relaxed reduced; do
    timeout 200 python tools/hazard_canary.py --numerics $n --isolation auto --seconds 40 > $OUT/canary_long_$n.json 2> $OUT/canary_long_$n.err; rc=$?
    python - $OUT/canary_long_$n.json "$n auto 40 s rc=$rc" <<'PY' | tee -a $S
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
    print(sys.argv[2], d["isolation"], "| mismatching checks", d["mismatching_checks"], {k: (v["checks"], v["bad_checks"], v["bad_elements"]) for k, v in d["classes"].items()}, "| examples", json.dumps(d["examples"][:3])[:400])
except Exception as e:
    print(sys.argv[2], "parse failed", e)
PY
  done;;
*) say "unknown section $sec";;
esac
done
