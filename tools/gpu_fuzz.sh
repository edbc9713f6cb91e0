#!/bin/bash
# usage: bash tools/gpu_fuzz.sh <seconds> <seed>   (prints a progress line every 30 s)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=gpurun_out/fuzz3; mkdir -p $O
PM_FUZZ_TRACE=${PM_FUZZ_TRACE:-} timeout -k 10 $(( $1 + 120 )) python tools/fuzz_campaign.py $1 $2 2>&1 | grep -v amdgpu.ids | tee -a $O/fuzz_seed$2.log
