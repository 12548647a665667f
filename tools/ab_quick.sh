# quick A/B of the default bench (persistent GRU) + serial16 GRU time; usage: tools/ab_quick.sh <tag>
set -u
export TMPDIR=/tmp
OUT=gpurun_out/$1; mkdir -p $OUT
for i in 1 2; do
  timeout 300 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['rooflines']; print('default:', d['value'], d['ms_per_step'], {k: (v['frac'], v['avg_launch_ms']) for k, v in r.items()})" | tee -a $OUT/summary.txt
done
timeout 300 python bench.py --pages 16 --inflight 1 --no-pipeline --steps 6 --warmup 2 --settle-s 0 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['rooflines']; print('serial16:', d['value'], d['ms_per_step'], {k: (v['frac'], v['avg_launch_ms']) for k, v in r.items()})" | tee -a $OUT/summary.txt
timeout 300 python bench.py --pages 1 --inflight 1 --no-pipeline --steps 20 --warmup 5 --settle-s 0 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['rooflines']; print('1page:', d['value'], d['ms_per_step'], {k: (v['frac'], v['avg_launch_ms']) for k, v in r.items()})" | tee -a $OUT/summary.txt
