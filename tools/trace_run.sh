#!/bin/bash
# One kernel trace of the pipelined default bench (48 steps) under the given environment, then the detection-chain and
# conv-stream analyses.  usage: tools/trace_run.sh <out_dir> <name> [ENV=VAL ...]
export TMPDIR=/tmp
OUT=$1; NAME=$2; shift 2
R=$PWD; mkdir -p $OUT
(cd /tmp && env "$@" timeout 400 rocprofv3 --kernel-trace -d $R/$OUT/prof -o $NAME -- python $R/bench.py --steps 48 --warmup 12 --settle-s 0 --no-cpu-baseline --no-extras --no-kernel-timing > $R/$OUT/$NAME.json 2> $R/$OUT/$NAME.err)
db=$(find $OUT/prof -name "${NAME}_results.db" | head -1)
echo "== $NAME ($*): $(python -c "import json,sys; d=json.loads([l for l in open('$OUT/$NAME.json') if l.startswith('{')][-1]); print('%.1f pages/s under rocprof' % d['value'])")"
(cd tools && python det_chains.py ../$db | head -6 && python heavy_timeline.py ../$db 14 58 | grep "mean period")
