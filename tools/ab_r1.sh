set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r2g
for i in 1 2; do
  (cd r1_snapshot && timeout 300 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('r1 code      :', d['value'], d['ms_per_step'], d['roofline']['frac'])") | tee -a gpurun_out/r2g/summary.txt
  OCRS_GRU_MODE=1 timeout 300 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('r2 step mode :', d['value'], d['ms_per_step'])" | tee -a gpurun_out/r2g/summary.txt
  OCRS_GRU_MODE=0 timeout 300 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('r2 persistent:', d['value'], d['ms_per_step'])" | tee -a gpurun_out/r2g/summary.txt
done
