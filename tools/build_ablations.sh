#!/bin/bash
# Timing-only variants of the coarse kNN kernels (outputs are wrong): which part of a tile costs what.
# Each variant library = the product objects with knn_coarse.o swapped for tools/ablation/knn_coarse_ablation.hip
# compiled with one policy.  Use with PM_LIB_PATH=<variant .so>.  NO_EPI applies to every coarse kernel;
# NO_STAGE / NO_BARRIER / NO_LDSREAD to the f32 kernel.
set -e
cd "$(dirname "$0")/.."
python -m points_matching_amd.build > /dev/null
B=points_matching_amd/build
mkdir -p $B/abl
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -ffinite-math-only -Iinclude -Ipoints_matching_amd/csrc -I/opt/rocm/include"
OBJS=$(ls $B/*.o | grep -v '/knn_coarse\.o$')
for v in BASE "NO_EPI" "NO_STAGE" "NO_BARRIER" "NO_LDSREAD" "NO_EPI NO_STAGE" "NO_EPI NO_STAGE NO_BARRIER" "NO_EPI NO_STAGE NO_BARRIER NO_LDSREAD"; do
  name=$(echo "$v" | tr -d '_' | tr ' ' '_')
  defs=""
  if [ "$v" != BASE ]; then for w in $v; do defs="$defs -DABL_$w=1"; done; fi
  /opt/rocm/bin/hipcc $F $defs -x hip -c tools/ablation/knn_coarse_ablation.hip -o /tmp/abl_coarse.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $B/abl/libpm_$name.so /tmp/abl_coarse.o $OBJS -ldl
  echo built $name
done
