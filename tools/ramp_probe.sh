#!/bin/bash
# Fixed cost of starting and draining the pipeline: the default bench at several step counts; T(K) = K * X + R.
# usage: tools/ramp_probe.sh [ENV=VAL ...]
for k in 12 24 48 96; do
  env "$@" timeout 300 python bench.py --steps $k --warmup 12 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%s steps %3d  %7.1f pages/s  %6.2f ms/step  total %.1f ms' % (' '.join(sys.argv[1:]), d['steps'], d['value'], d['ms_per_step'], d['ms_per_step']*d['steps']))" "$@"
done
