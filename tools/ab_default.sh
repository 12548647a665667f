for v in "8 6 24" "16 4 12" "16 6 12" "32 3 6" "32 4 8"; do
set -- $v
timeout 300 python bench.py --no-cpu-baseline --no-extras --pages $1 --inflight $2 --steps $3 --warmup $2 > /tmp/o.json 2>/tmp/o.err || tail -3 /tmp/o.err
python -c "
import json; d=json.load(open('/tmp/o.json')); print('pages=$1 inflight=$2', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'])"
done
