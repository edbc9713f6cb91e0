#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=gpurun_out/r3q; mkdir -p $O
for lib in knnstamps knnstamps_NOLDSREAD knnstamps_NOEPI_NOLDSREAD; do
  echo "=== $lib 12=6" | tee -a $O/stamps.log
  PM_LIB_PATH=points_matching_amd/build/abl/libpm_$lib.so timeout -k 10 120 python tools/prof_knn_stamps.py 32768 32768 12=6 2>&1 | grep -v amdgpu.ids | grep "options\|block\|sweep" | tee -a $O/stamps.log
done
