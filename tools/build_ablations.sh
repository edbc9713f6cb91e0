#!/bin/bash
# Timing-only variants of the coarse kNN kernel (outputs are wrong): which part of a tile costs what.
set -e
cd "$(dirname "$0")/.."
mkdir -p points_matching_amd/build/abl
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Iinclude -Ipoints_matching_amd/csrc"
for v in BASE NOEPI NOSTAGE NOBARRIER NOLDSREAD "NOEPI -DPM_ABL_NOSTAGE" "NOEPI -DPM_ABL_NOSTAGE -DPM_ABL_NOBARRIER" "NOEPI -DPM_ABL_NOSTAGE -DPM_ABL_NOBARRIER -DPM_ABL_NOLDSREAD"; do
  name=$(echo "$v" | sed 's/ -DPM_ABL_/_/g')
  /opt/rocm/bin/hipcc $F -DPM_ABL_$v -x hip -c points_matching_amd/csrc/knn_l2.hip -o /tmp/abl_knn.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o points_matching_amd/build/abl/libpm_$name.so /tmp/abl_knn.o points_matching_amd/build/pm_capi.o points_matching_amd/build/knn_hamming.o points_matching_amd/build/ransac.o points_matching_amd/build/filter_gather.o
  echo built $name
done
