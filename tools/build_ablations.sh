#!/bin/bash
# Timing-only variants of the coarse kNN kernels (outputs are wrong): which part of a tile costs what.
# PM_ABL_NOEPI drops the in-chain selection (both the f32 kernel and the 288/256-byte-row kernel);
# NOSTAGE / NOBARRIER / NOLDSREAD apply to the f32 kernel.  Use with PM_LIB_PATH=<variant .so>.
set -e
cd "$(dirname "$0")/.."
python -m points_matching_amd.build > /dev/null
mkdir -p points_matching_amd/build/abl
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -ffinite-math-only -Iinclude -Ipoints_matching_amd/csrc"
B=points_matching_amd/build
for v in BASE NOEPI NOSTAGE NOBARRIER NOLDSREAD "NOEPI -DPM_ABL_NOSTAGE" "NOEPI -DPM_ABL_NOSTAGE -DPM_ABL_NOBARRIER" "NOEPI -DPM_ABL_NOSTAGE -DPM_ABL_NOBARRIER -DPM_ABL_NOLDSREAD"; do
  name=$(echo "$v" | sed 's/ -DPM_ABL_/_/g')
  /opt/rocm/bin/hipcc $F -DPM_ABL_$v -x hip -c points_matching_amd/csrc/knn_coarse.hip -o /tmp/abl_coarse.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $B/abl/libpm_$name.so /tmp/abl_coarse.o $B/pm_capi.o $B/knn_l2.o $B/knn_hamming.o $B/ransac.o $B/filter_gather.o $B/pair_batch.o $B/lmeds.o
  echo built $name
done
