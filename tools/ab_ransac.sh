#!/bin/bash
# A/B of two builds of the one-launch RANSAC kernel (the product vs points_matching_amd/build/abl/libpm_<name>.so)
set -e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out/ab
alt=${1:-teams2}
out=gpurun_out/ab/ransac_ab_$alt.log
: > $out
for r in 1 2 3; do
  for n in 2275 9175; do
    h=10000; cap=8192; [ $n = 9175 ] && h=100000 && cap=16384
    echo "product n=$n" >> $out
    timeout -k 10 200 python tools/prof_ransac.py $n $h $cap 50 2>&1 | grep -v amdgpu.ids >> $out
    echo "$alt n=$n" >> $out
    PM_LIB_PATH=points_matching_amd/build/abl/libpm_$alt.so timeout -k 10 200 python tools/prof_ransac.py $n $h $cap 50 2>&1 | grep -v amdgpu.ids >> $out
  done
done
cat $out
