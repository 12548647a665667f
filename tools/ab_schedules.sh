#!/bin/bash
# The scheduling experiments of DESIGN.md §10 (round 3), ABAB on one box: default | projections on the conv-stack stream |
# lean background recurrence | three conv blocks per CU.  Output: one line per run.
run() { env "$@" timeout 300 python bench.py --steps 48 --warmup 24 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%-46s %7.1f pages/s  %6.2f ms/step  ' % (' '.join(sys.argv[1:]), d['value'], d['ms_per_step']) + '  '.join('%s %.3f (%.2f ms x %.0f)' % (k.replace('gemm_','').replace('_mfma',''), v['frac'], v['avg_launch_ms'], v['launches_per_step']) for k,v in d['rooflines'].items()) + '  p50 %.0f ms' % d['request_latency_ms']['p50'])" "$@"; }
for rep in 1 2; do
run OCRS_DEFAULT=1
run OCRS_GX_HEAVY=1
run OCRS_GRU_BACKGROUND=1 OCRS_GRU_BG_LAZY=8 OCRS_GRU_BG_PRIO=0
run OCRS_CONV_OCCUPANCY=3
done
