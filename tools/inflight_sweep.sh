#!/bin/bash
# Requests in flight vs throughput, request latency and device memory: python bench.py --inflight N for N = 2..7.
cd $GRAFT_REPO_ROOT
OUT=${1:-gpurun_out/inflight}
mkdir -p $OUT
for n in ${SWEEP:-2 3 4 5 6 7}; do
  timeout 300 python bench.py --inflight $n --steps $((12 * n)) --warmup $((4 * n)) --no-cpu-baseline --no-extras > $OUT/inflight_$n.json 2>/dev/null
  python - $OUT/inflight_$n.json $n <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().splitlines()[-1])
l = d["request_latency_ms"]
print("inflight=%s: %.1f pages/s, request latency p50 %.0f ms p99 %.0f ms, device memory %s GB, conv live %.3f" % (
    sys.argv[2], d["value"], l["p50"], l["p99"], d.get("device_memory_in_use_gb"), d["rooflines"]["gemm_conv3x3_mfma"]["frac"]))
PY
done
