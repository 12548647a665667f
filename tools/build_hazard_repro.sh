#!/bin/bash
# Builds tools/_build/hazard_repro[.prioN] from tools/hazard_repro.hip and the PRODUCT's own kernel sources (same flags as
# ocrs_amd/build.py).  hipcc cross-compiles for gfx950 without a GPU; the binaries travel to the GPU box with the tree.
#   tools/build_hazard_repro.sh            # the kernels as shipped
#   tools/build_hazard_repro.sh 3          # ... and a variant whose victim kernel starts with s_setprio 3
set -e
cd "$(dirname "$0")/.."
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function -Wno-pass-failed"
C=ocrs_amd/csrc
mkdir -p tools/_build
build() {   # $1 = output suffix, $2 = extra define for kernels_lines.hip
  for f in kernels_nn kernels_rec; do
    [ tools/_build/$f.o -nt $C/$f.hip ] || /opt/rocm/bin/hipcc $FLAGS -c $C/$f.hip -o tools/_build/$f.o &
  done
  /opt/rocm/bin/hipcc $FLAGS $2 -c $C/kernels_lines.hip -o tools/_build/kernels_lines$1.o &
  /opt/rocm/bin/hipcc $FLAGS -x hip -c $C/common.cpp -o tools/_build/common.o &
  /opt/rocm/bin/hipcc $FLAGS -c tools/hazard_repro.hip -o tools/_build/hazard_repro.o &
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -o tools/_build/hazard_repro$1 tools/_build/hazard_repro.o tools/_build/kernels_lines$1.o \
      tools/_build/kernels_nn.o tools/_build/kernels_rec.o tools/_build/common.o -ldl -lpthread
  echo built tools/_build/hazard_repro$1
}
build "" ""
if [ -n "$1" ]; then build ".prio$1" "-DOCRS_CROP_SETPRIO=$1"; fi
# aggressor probes: gemm_split_kernel with its accumulators in AGPRs / driven through the 16x16x32 instruction
for v in ACC_AGPR MFMA16 ALL_AGPR; do
  /opt/rocm/bin/hipcc $FLAGS -DOCRS_PROBE_$v -c $C/kernels_nn.hip -o tools/_build/kernels_nn.$v.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -o tools/_build/hazard_repro.$v tools/_build/hazard_repro.o tools/_build/kernels_lines.o \
      tools/_build/kernels_nn.$v.o tools/_build/kernels_rec.o tools/_build/common.o -ldl -lpthread
  echo built tools/_build/hazard_repro.$v
done
