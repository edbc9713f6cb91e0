#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=gpurun_out/r3ad; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_knn_seeded_gpu.py tests/test_pair_batch_gpu.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -2 $O/pytest.log
timeout -k 10 200 python tools/sweep_u8.py 32768 32768 "" "12=2" "12=3" "12=5" "12=6" 2>&1 | grep -v amdgpu.ids | tee $O/sweep.log
