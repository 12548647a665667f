#!/bin/bash
# default bench with the dominant conv kernel at 3 vs 4 blocks per CU (3 leaves room for the GRU steps of other requests)
cd $GRAFT_REPO_ROOT
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fvisibility=hidden"
for v in 3 4 3 4; do
  hipcc $FLAGS -DOCRS_CONV_WAVES=$v -c ocrs_amd/csrc/kernels_rec.hip -o ocrs_amd/_build/kernels_rec.o || exit 1
  hipcc --offload-arch=gfx950 -shared -fPIC -o ocrs_amd/libocrs_amd.so ocrs_amd/_build/*.o -lpthread || exit 1
  timeout 200 python bench.py --no-cpu-baseline --no-extras > /tmp/o.json 2>/dev/null
  python -c "
import json; d=json.load(open('/tmp/o.json')); print('conv blocks/CU=$v', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['pipeline']['sustained_tflops'])"
done
