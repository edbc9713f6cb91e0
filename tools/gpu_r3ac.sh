#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=gpurun_out/r3ac; mkdir -p $O
for a in "8192 0 12" "8192 40 12" "32768 0 12" "32768 40 12"; do timeout -k 10 120 python tools/tie_tail.py $a 2>&1 | grep -v amdgpu.ids | tee -a $O/tie.log; done
