#!/bin/bash
# usage: bash tools/gpu_fuzz.sh <seconds> <seed>   (a progress line every 30 s, unbuffered: a silent run is taken to be hung)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=gpurun_out/fuzz3; mkdir -p $O
PM_FUZZ_TRACE=${PM_FUZZ_TRACE:-} timeout -k 10 $(( $1 + 120 )) python -u tools/fuzz_campaign.py $1 $2 2>&1 | grep --line-buffered -v amdgpu.ids | tee -a $O/fuzz_seed$2.log
rc=${PIPESTATUS[0]}
if grep -q "Memory access fault" $O/fuzz_seed$2.log; then echo "GPU fault in the campaign"; exit 1; fi
exit $rc
