#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=gpurun_out/r3d; mkdir -p $O
for v in "" _NOEPI _NOLDSREAD _NOEPI_NOLDSREAD _NOBARRIER _NOEPI_NOLDSREAD_NOBARRIER_NOSTAGE; do
  for o in "14=8" "14=8 4=3"; do
    echo "=== variant '$v' opts: $o" >> $O/abl.log
    PM_LIB_PATH=points_matching_amd/build/abl/libpm_knnstamps$v.so timeout -k 10 120 python tools/prof_knn_stamps.py 8192 8192 $o 2>&1 | grep -v amdgpu.ids | grep "hipEvent\|span\|clock\|tile [0-4] \|issued" >> $O/abl.log
  done
done
cat $O/abl.log
