#!/bin/bash
# ABAB of the default bench between the product library and a variant build (python -m ocrs_amd.build --variant <tag> ...):
# tools/ab_lib.sh <tag> [reps] [extra bench args]
cd $GRAFT_REPO_ROOT
TAG=$1; REPS=${2:-2}; shift; shift
for rep in $(seq $REPS); do
for v in product $TAG; do
  lib=""; [ $v != product ] && lib=$PWD/ocrs_amd/libocrs_amd.$v.so
  OCRS_AMD_LIB=$lib timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 48 --warmup 12 "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['rooflines']
print('$v: %.1f pages/s | %s | p50 %s ms' % (d['value'], ', '.join('%s %.3f live (%.2f ms)' % (k.replace('gemm_', '').replace('_mfma', ''), x['frac'], x['avg_launch_ms']) for k, x in r.items()), d['request_latency_ms']['p50']))"
done
done
