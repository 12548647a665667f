set -u
export TMPDIR=/tmp
OUT=gpurun_out/r2l; mkdir -p $OUT
for v in "OCRS_POOL_TRACE=1" "OCRS_POOL_TRACE=1 OCRS_POOL_EXACT=1" "OCRS_POOL_TRACE=1 OCRS_POOL_CAP_GB=250" "OCRS_POOL_TRACE=1 OCRS_POOL_EXACT=1 OCRS_POOL_CAP_GB=250"; do
  env $v timeout 300 python bench.py --no-cpu-baseline --no-extras > $OUT/b.json 2> $OUT/b.err
  python -c "import sys,json; d=json.loads(open('$OUT/b.json').read().strip().splitlines()[-1]); print('%-60s' % '$v', d['value'], d['ms_per_step'])" | tee -a $OUT/summary.txt
  echo "   hipMalloc calls: $(grep -c 'hipMalloc' $OUT/b.err)  last at: $(grep 'hipMalloc' $OUT/b.err | tail -1 | cut -c1-60)" | tee -a $OUT/summary.txt
done
