#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=gpurun_out/big3; mkdir -p $O
timeout -k 10 900 python -u tools/big_shapes_check.py $1 2>&1 | grep --line-buffered -v amdgpu.ids | tee $O/big_shapes.log
rc=${PIPESTATUS[0]}
if grep -q "Memory access fault" $O/big_shapes.log; then exit 1; fi
exit $rc
