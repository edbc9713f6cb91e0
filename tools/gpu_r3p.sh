#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=gpurun_out/r3p; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_knn_seeded_gpu.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -5 $O/pytest.log
grep -q "rc=0" $O/pytest.log || exit 1
for n in 8192 16384 32768; do timeout -k 10 200 python tools/sweep_u8.py $n $n "" "12=5" "12=6" 2>&1 | grep -v amdgpu.ids | tee -a $O/sweep.log; done
PM_LIB_PATH=points_matching_amd/build/abl/libpm_knnstamps.so timeout -k 10 120 python tools/prof_knn_stamps.py 32768 32768 12=6 2>&1 | grep -v amdgpu.ids | tail -14 | tee -a $O/stamps.log
