#!/bin/bash
# ABAB of one environment switch on one box: tools/ab_env.sh VAR A B [reps]   (one line per run)
VAR=$1; A=$2; B=$3; REPS=${4:-2}
run() { env "$@" timeout 300 python bench.py --steps ${AB_STEPS:-48} --warmup ${AB_WARMUP:-24} --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%-30s %7.1f pages/s  %6.2f ms/step  ' % (' '.join(sys.argv[1:]), d['value'], d['ms_per_step']) + '  '.join('%s %.3f (%.2f ms x %.0f)' % (k.replace('gemm_','').replace('_mfma',''), v['frac'], v['avg_launch_ms'], v['launches_per_step']) for k,v in d['rooflines'].items()) + '  text_match %s' % d.get('text_match'))" "$@"; }
for rep in $(seq $REPS); do run $VAR=$A; run $VAR=$B; done
