#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=gpurun_out/r3k; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_flann_gpu.py tests/test_cli_gpu.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -4 $O/pytest.log
timeout -k 10 300 python tools/prof_flann.py 2>&1 | grep -v amdgpu.ids | tee $O/flann.log
for H in 10000 5000 2500 1280 256; do
  echo "=== H $H (ids per workgroup: $((($H+255)/256)))" >> $O/ransac_stamps.log
  PM_LIB_PATH=points_matching_amd/build/abl/libpm_rfstamps.so timeout -k 10 120 python tools/prof_ransac_stamps.py 2275 $H 8192 2>&1 | grep -v amdgpu.ids >> $O/ransac_stamps.log
done
cat $O/ransac_stamps.log
