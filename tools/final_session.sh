#!/bin/bash
# GPU box: the whole -m gpu suite, then the measurement session of tools/gpu_round2.sh.   bash tools/final_session.sh <tag>
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/${1:-r2final}
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/${1:-r2final}/pytest_full.log 2>&1 || { tail -40 gpurun_out/${1:-r2final}/pytest_full.log; exit 1; }
tail -2 gpurun_out/${1:-r2final}/pytest_full.log
bash tools/gpu_round2.sh ${1:-r2final} > gpurun_out/${1:-r2final}/session.log 2>&1
tail -5 gpurun_out/${1:-r2final}/session.log
