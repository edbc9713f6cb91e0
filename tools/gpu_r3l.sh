#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=gpurun_out/r3l; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_pipeline_gpu.py tests/test_pair_batch_gpu.py tests/test_mgpu_gpu.py tests/test_knn_seeded_gpu.py tests/test_flann_gpu.py tests/test_bench_gpu.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -6 $O/pytest.log
for n in 2048 8192 32768; do timeout -k 10 200 python tools/sweep_ratio.py $n $n 2>&1 | grep -v amdgpu.ids | tee -a $O/sweep_ratio.log; done
timeout -k 10 300 python tools/prof_flann.py 2>&1 | grep -v amdgpu.ids | tee $O/flann.log
