#!/bin/bash
# A/B of the coarse-kernel options on the GPU box (parity tests first): bash tools/ab_waves.sh
set -e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out/ab
timeout -k 10 600 python -m pytest tests/test_knn_l2_gpu.py tests/test_knn_hamming_gpu.py -x -q -m gpu > gpurun_out/ab/pytest.log 2>&1 && \
timeout -k 10 300 python tools/ab_options.py 6 1,2 sift 8192 8192 50 > gpurun_out/ab/staging_c3.log 2>&1 && \
timeout -k 10 300 python tools/ab_options.py 4 1,2,3 sift 8192 8192 50 > gpurun_out/ab/waves_c3.log 2>&1 && \
timeout -k 10 300 python tools/ab_options.py 4 1,2,3 sift 32768 32768 10 > gpurun_out/ab/waves_32k.log 2>&1 && \
timeout -k 10 300 python tools/ab_options.py 4 1,2 orb 32768 32768 10 > gpurun_out/ab/waves_orb32k.log 2>&1 && \
timeout -k 10 300 python tools/ab_options.py 4 1,2,3 sift 4096 4096 50 > gpurun_out/ab/waves_4k.log 2>&1
tail -3 gpurun_out/ab/pytest.log; cat gpurun_out/ab/staging_*.log gpurun_out/ab/waves_*.log | grep -v amdgpu.ids
