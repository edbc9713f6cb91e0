#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=gpurun_out/r3o; mkdir -p $O
for lib in knnstamps knnstamps_NOEPI; do
for o in "12=6" "12=5"; do
  echo "=== $lib $o" | tee -a $O/stamps.log
  PM_LIB_PATH=points_matching_amd/build/abl/libpm_$lib.so timeout -k 10 120 python tools/prof_knn_stamps.py 32768 32768 $o 2>&1 | grep -v amdgpu.ids | tee -a $O/stamps.log
done; done
