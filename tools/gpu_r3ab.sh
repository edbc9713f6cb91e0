#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=gpurun_out/r3ab; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_knn_seeded_gpu.py tests/test_pair_batch_gpu.py tests/test_pipeline_gpu.py tests/test_knn_l2_gpu.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -2 $O/pytest.log
grep -q "rc=0" $O/pytest.log || exit 1
for rep in 1 2; do
for lib in "" refine_old; do
  if [ -z "$lib" ]; then unset PM_LIB_PATH; else export PM_LIB_PATH=points_matching_amd/build/abl/libpm_$lib.so; fi
  echo "== lib ${lib:-product}" | tee -a $O/ab.log
  timeout -k 10 200 python tools/sweep_u8.py 8192 8192 "" 2>&1 | grep "default" | tee -a $O/ab.log
done; done
unset PM_LIB_PATH
for a in "8192 40 12" "8192 400 12" "32768 40 12"; do timeout -k 10 120 python tools/tie_tail.py $a 2>&1 | grep -v amdgpu.ids | grep "u8 hint" | tee -a $O/tie.log; done
