# A/B of bench.py argument variants; usage: tools/ab_args.sh <tag> "args" "args" ...
set -u
export TMPDIR=/tmp
OUT=gpurun_out/$1; shift; mkdir -p $OUT
for rep in 1 2; do
for v in "$@"; do
  timeout 400 python bench.py --no-cpu-baseline --no-extras $v 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['rooflines']; print('%-34s' % '$v', d['value'], d['ms_per_step'], 'cores', d['host_cpu_cores_busy_per_gpu'], {k.replace('gemm_','').replace('_mfma',''): (v['frac'], v['avg_launch_ms']) for k, v in r.items()})" | tee -a $OUT/summary.txt
done; done
