#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=gpurun_out/r3ae; mkdir -p $O
for o in '12=2' '12=6'; do echo "== options $o"; PM_LIB_PATH=points_matching_amd/build/abl/libpm_knnstamps.so timeout -k 10 120 python tools/prof_knn_stamps.py 32768 32768 $o 2>&1 | grep -v amdgpu.ids; done > $O/knn_stamps_32k.txt 2>&1
cat $O/knn_stamps_32k.txt | tail -40
