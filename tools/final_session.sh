cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2d
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r2d/pytest_full.log 2>&1 || { tail -40 gpurun_out/r2d/pytest_full.log; exit 1; }
tail -2 gpurun_out/r2d/pytest_full.log
bash tools/gpu_round2.sh r2d > gpurun_out/r2d/session.log 2>&1
tail -5 gpurun_out/r2d/session.log
