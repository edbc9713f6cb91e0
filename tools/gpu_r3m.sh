#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=gpurun_out/r3m; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_ransac_fused_gpu.py tests/test_ransac_gpu.py tests/test_pipeline_gpu.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -6 $O/pytest.log
for f in 1 2; do
  echo "== form $f" | tee -a $O/stamps.log
  PM_RANSAC_FORM=$f PM_LIB_PATH=points_matching_amd/build/abl/libpm_rfstamps.so timeout -k 10 120 python tools/prof_ransac_stamps.py 2>&1 | grep -v amdgpu.ids | tee -a $O/stamps.log
done
timeout -k 10 300 python bench.py --no-large --steps 300 --warmup 50 2>&1 | grep -v amdgpu.ids | tee $O/bench.log
