#!/bin/bash
# Where does the dominant conv kernel lose its MFMA issue slots?  Builds it with parts removed (results
# are wrong on purpose) and times the conv class in isolation.  OCRS_ABL bits: 1 barriers, 2 global loads,
# 4 LDS writes.  (Mask 8 — every A load redirected to one L2-resident window — existed for the r1 study in
# DESIGN.md §10 and went away with the buffer-load rewrite of the load path.)
cd $GRAFT_REPO_ROOT
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fvisibility=hidden"
for v in ${ABL_SET:-0 1 2 4 6 7 0}; do
  hipcc $FLAGS -DOCRS_ABL=$v -c ocrs_amd/csrc/kernels_rec.hip -o ocrs_amd/_build/kernels_rec.o || exit 1
  hipcc --offload-arch=gfx950 -shared -fPIC -o ocrs_amd/libocrs_amd.so ocrs_amd/_build/*.o -lpthread || exit 1
  echo "== ablation mask $v"
  timeout 200 python bench.py --pages ${ABL_PAGES:-16} --steps 6 --warmup 2 --inflight 1 --no-cpu-baseline --no-extras --profile-hint --no-pipeline 2>&1 >/dev/null | grep -E "gemm_conv3x3"
done
