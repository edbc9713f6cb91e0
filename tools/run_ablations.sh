#!/bin/bash
# Time every timing-only variant of the coarse kernels (tools/build_ablations.sh) at one shape; results are WRONG by
# construction, only the kernel times mean anything.   bash tools/run_ablations.sh [kind nq nt reps]
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
kind=${1:-sift}; nq=${2:-8192}; nt=${3:-8192}; reps=${4:-50}
mkdir -p gpurun_out/ab
out=gpurun_out/ab/ablations_${kind}_${nq}x${nt}.log
: > $out
for v in BASE NOEPI NOSTAGE NOBARRIER NOLDSREAD NOEPI_NOSTAGE NOEPI_NOSTAGE_NOBARRIER NOEPI_NOSTAGE_NOBARRIER_NOLDSREAD; do
  echo "== $v" >> $out
  PM_LIB_PATH=points_matching_amd/build/abl/libpm_$v.so timeout -k 10 200 python tools/ab_options.py 6 2 $kind $nq $nt $reps 2>&1 | grep -v amdgpu.ids >> $out || exit 1
done
cat $out
