#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=gpurun_out/r3f; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_knn_seeded_gpu.py -x -q -m gpu > $O/pytest_seeded.log 2>&1; echo "seeded rc=$?" | tee -a $O/pytest_seeded.log
tail -3 $O/pytest_seeded.log
timeout -k 10 300 python tools/sweep_u8.py 8192 8192 "" "11=1" "11=3" "flags=4" 2>&1 | grep -v amdgpu.ids | tee $O/sweep.log
timeout -k 10 300 python tools/sweep_u8.py 32768 32768 "" "11=3" 2>&1 | grep -v amdgpu.ids | tee -a $O/sweep.log
timeout -k 10 300 python tools/sweep_u8.py 2048 2048 "" "flags=4" 2>&1 | grep -v amdgpu.ids | tee -a $O/sweep.log
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $O/pytest_all.log 2>&1; echo "all rc=$?" | tee -a $O/pytest_all.log
tail -5 $O/pytest_all.log
