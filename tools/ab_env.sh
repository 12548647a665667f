# A/B of the default bench under environment variants; usage: tools/ab_env.sh <tag> "VAR=val VAR2=val" "..." ...
set -u
export TMPDIR=/tmp
OUT=gpurun_out/$1; shift; mkdir -p $OUT
for rep in 1 2; do
for v in "$@"; do
  env $v timeout 300 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['rooflines']; print('%-40s' % '$v', d['value'], d['ms_per_step'], {k.replace('gemm_','').replace('_mfma',''): (v['frac'], v['avg_launch_ms']) for k, v in r.items()})" | tee -a $OUT/summary.txt
done; done
