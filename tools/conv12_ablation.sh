#!/bin/bash
# Where does the fused conv1 + conv2 kernel spend its time?  Builds it with parts removed (results are wrong on purpose)
# and times the conv class alone.  OCRS_F12_ABL bits: 1 no conv1 stage, 2 no weight loads, 4 an eighth of the MFMAs.
cd $GRAFT_REPO_ROOT
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fvisibility=hidden"
for v in ${ABL_SET:-0 1 2 4 5 7}; do
  hipcc $FLAGS -DOCRS_F12_ABL=$v -c ocrs_amd/csrc/kernels_rec.hip -o ocrs_amd/_build/kernels_rec.o || exit 1
  hipcc --offload-arch=gfx950 -shared -fPIC -o ocrs_amd/libocrs_amd.so ocrs_amd/_build/*.o -L/opt/rocm/lib -Wl,-rpath,/opt/rocm/lib -lrccl -lpthread || exit 1
  echo "== ablation $v: $(timeout 200 python bench.py --pages 16 --steps 4 --warmup 2 --inflight 1 --no-cpu-baseline --no-extras --profile-hint --no-pipeline --settle-s 0 2>&1 >/dev/null | grep -E '^gemm_conv3x3' | cut -c1-100)"
done
