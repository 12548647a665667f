#!/bin/bash
# A/B the dominant conv kernel's build-time knobs on the GPU box (same process setup, same data).
cd $GRAFT_REPO_ROOT
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fvisibility=hidden"
for v in "16 3" "16 4" "16 3" "16 4"; do
  set -- $v
  hipcc $FLAGS -DOCRS_CONV_BK=$1 -DOCRS_CONV_WAVES=$2 -c ocrs_amd/csrc/kernels_rec.hip -o ocrs_amd/_build/kernels_rec.o || exit 1
  hipcc --offload-arch=gfx950 -shared -fPIC -o ocrs_amd/libocrs_amd.so ocrs_amd/_build/*.o -lpthread || exit 1
  echo "== BK=$1 waves=$2"
  timeout 200 python bench.py --pages 8 --steps 8 --warmup 3 --inflight 1 --no-cpu-baseline --no-extras --profile-hint --no-pipeline 2>&1 >/dev/null | grep -E "gemm_conv3x3" 
done
