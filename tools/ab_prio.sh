for i in 1 2; do
for v in "high 6" "high 8" "high 12"; do
  set -- $v
  OCRS_REQ_PRIO=$1 timeout 200 python bench.py --no-cpu-baseline --no-extras --inflight $2 --steps 32 > /tmp/o.json 2>/dev/null
  python -c "
import json; d=json.load(open('/tmp/o.json')); print('req=$1 inflight=$2', d['value'], d['ms_per_step'], d['roofline']['achieved'])"
done; done
