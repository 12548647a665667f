#!/bin/bash
# gru_waves 16 (gate-per-wave teams, round 4) vs 4 (round 2's kernel): serial (kernel alone) and the default bench, ABAB.
cd $GRAFT_REPO_ROOT
OUT=${1:-gpurun_out/ab_gru}
mkdir -p $OUT
for w in 16 4 16 4; do
  OCRS_GRU_WAVES=$w ${GRU_ENV:-} timeout 300 python bench.py --pages 16 --inflight 1 --no-pipeline --steps 6 --warmup 2 --settle-s 0 --no-cpu-baseline --no-extras --profile-hint 2>&1 >/dev/null | grep -E "gemm_gru_hidden" | sed "s/^/serial waves=$w /"
done
for w in 16 4 16 4; do
  OCRS_GRU_WAVES=$w ${GRU_ENV:-} timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 36 --warmup 12 > $OUT/def_$w.json 2>/dev/null
  python - $OUT/def_$w.json $w <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().splitlines()[-1])
r = d["rooflines"]
print("default waves=%s: %.1f pages/s | %s" % (sys.argv[2], d["value"], ", ".join("%s %.3f live (%.2f ms) %.3f alone" % (k.replace("gemm_", "").replace("_mfma", ""), v["frac"], v["avg_launch_ms"], v.get("frac_alone", 0)) for k, v in r.items())))
PY
done
