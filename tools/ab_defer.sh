set -u
export TMPDIR=/tmp
OUT=gpurun_out/r2o; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_bench_scale.py -x -q 2>&1 | tail -2 | tee -a $OUT/summary.txt
for rep in 1 2; do
for v in "OCRS_GRU_DEFER=1" "OCRS_GRU_DEFER=0"; do
  env $v timeout 300 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['rooflines']; print('%-20s default ' % '$v', d['value'], d['ms_per_step'], {k.replace('gemm_','').replace('_mfma',''): (v['frac'], v['avg_launch_ms']) for k, v in r.items()})" | tee -a $OUT/summary.txt
done; done
for v in "OCRS_GRU_DEFER=1" "OCRS_GRU_DEFER=0"; do
  env $v timeout 300 python bench.py --pages 16 --inflight 1 --no-pipeline --steps 6 --warmup 2 --settle-s 0 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['rooflines']; print('%-20s serial16' % '$v', d['value'], d['ms_per_step'], {k.replace('gemm_','').replace('_mfma',''): (v['frac'], v['avg_launch_ms']) for k, v in r.items()})" | tee -a $OUT/summary.txt
done
