set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2c
timeout -k 10 600 python -m pytest tests/test_knn_l2_gpu.py tests/test_knn_hamming_gpu.py tests/test_pipeline_gpu.py -x -q -m gpu > gpurun_out/r2c/pytest.log 2>&1 || { tail -30 gpurun_out/r2c/pytest.log; exit 1; }
tail -2 gpurun_out/r2c/pytest.log
timeout -k 10 300 python tools/ab_options.py 6 2 sift 8192 8192 50 2>&1 | grep -v amdgpu.ids
timeout -k 10 300 python tools/ab_options.py 6 2 surf 8192 8192 20 2>&1 | grep -v amdgpu.ids
timeout -k 10 300 python tools/ab_options.py 6 2 orb 32768 32768 10 2>&1 | grep -v amdgpu.ids
timeout -k 10 300 python tools/ab_options.py 6 2 sift 32768 32768 10 2>&1 | grep -v amdgpu.ids
bash tools/gpu_round2.sh r2c pmc-only > gpurun_out/r2c/pmc.log 2>&1
python tools/pmc_summary.py gpurun_out/r2c/pmc_knn rows288 | grep -E "FETCH|WRITE|fetch_bytes|write_bytes"
python tools/pmc_summary.py gpurun_out/r2c/pmc_ham rows288 | grep -E "FETCH|WRITE|fetch_bytes|write_bytes"
python tools/pmc_summary.py gpurun_out/r2c/pmc_knn knn_l2_mfma | grep -E "fetch_bytes|write_bytes"
