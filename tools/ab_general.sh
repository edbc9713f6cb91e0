#!/bin/bash
# parity + A/B of the general-float coarse pass (PM_OPT_KNN_GENERAL_F16: 1 = f32-input MFMA, 2 = f16-rounded copies)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out/gen
timeout -k 10 900 python -m pytest tests/test_knn_l2_gpu.py tests/test_pipeline_gpu.py tests/test_independent_gpu.py tests/test_c1_images.py -x -q -m gpu > gpurun_out/gen/pytest.log 2>&1 || { tail -40 gpurun_out/gen/pytest.log; exit 1; }
tail -2 gpurun_out/gen/pytest.log
for sh in "surf 8192 8192 30" "surf 4096 4096 30" "surf 32768 32768 5" "sift 8192 8192 30"; do
  timeout -k 10 300 python tools/ab_options.py 9 1,2 $sh 2>&1 | grep -v amdgpu.ids || exit 1
done
