#!/bin/bash
# Round 5, verdict item 1: what the recurrence's gate arithmetic costs.  Four builds of kernels_gru.hip
# (python -m ocrs_amd.build --variant gateN kernels_gru.hip OCRS_GATE_MATH=N): canon = the numeric spec (degree-7 exp +
# IEEE divide), gate1 = hardware v_exp_f32 / v_rcp_f32, gate2 = a cheaper CPU-reproducible candidate (degree-6 exp,
# division-free reciprocal), gate3 = no transcendental at all.  1..3 give wrong bits: timing only.
cd $GRAFT_REPO_ROOT
OUT=${1:-gpurun_out/ab_gate}
mkdir -p $OUT
for rep in 1 2; do
for v in canon gate1 gate2 gate3; do
  lib=""; [ $v != canon ] && lib=$PWD/ocrs_amd/libocrs_amd.$v.so
  OCRS_AMD_LIB=$lib timeout 300 python bench.py --pages 16 --inflight 1 --no-pipeline --steps 6 --warmup 2 --settle-s 0 --no-cpu-baseline --no-extras --profile-hint 2>&1 >/dev/null | grep -E "gemm_gru_hidden" | sed "s/^/serial16 $v /"
  OCRS_AMD_LIB=$lib timeout 300 python bench.py --pages 1 --inflight 1 --no-pipeline --steps 20 --warmup 5 --settle-s 0 --no-cpu-baseline --no-extras --profile-hint 2>$OUT/lat_$v.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['rooflines']
print('1page $v: %.2f ms/page | gru_hidden %.3f ms/launch' % (d['ms_per_step'], r['gemm_gru_hidden_mfma']['avg_launch_ms']))"
done
done
for v in canon gate1 canon gate1; do
  lib=""; [ $v != canon ] && lib=$PWD/ocrs_amd/libocrs_amd.$v.so
  OCRS_AMD_LIB=$lib timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 36 --warmup 12 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['rooflines']
print('default $v: %.1f pages/s | %s' % (d['value'], ', '.join('%s %.3f live (%.2f ms)' % (k.replace('gemm_', '').replace('_mfma', ''), x['frac'], x['avg_launch_ms']) for k, x in r.items())))"
done
