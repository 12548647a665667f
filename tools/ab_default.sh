# default bench repeated (no CPU baseline / extras)
for i in 1 2 3 4; do
timeout 200 python bench.py --no-cpu-baseline --no-extras > /tmp/o.json 2>/dev/null
python -c "
import json; d=json.load(open('/tmp/o.json')); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['pipeline']['sustained_tflops'], d['host_cpu_cores_busy_per_gpu'])"
done
