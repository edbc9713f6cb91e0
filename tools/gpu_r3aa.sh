#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=gpurun_out/r3aa; mkdir -p $O
for rep in 1 2; do
for lib in "" refine_old refine_w5; do
  if [ -z "$lib" ]; then export -n PM_LIB_PATH; unset PM_LIB_PATH; else export PM_LIB_PATH=points_matching_amd/build/abl/libpm_$lib.so; fi
  echo "== lib ${lib:-product}" | tee -a $O/ab.log
  timeout -k 10 200 python tools/sweep_u8.py 8192 8192 "" 2>&1 | grep "default" | tee -a $O/ab.log
done; done
unset PM_LIB_PATH
timeout -k 10 120 python tools/sweep_u8.py 2048 2048 "" 2>&1 | grep default | tee -a $O/ab.log
PM_LIB_PATH=points_matching_amd/build/abl/libpm_refine_old.so timeout -k 10 120 python tools/sweep_u8.py 2048 2048 "" 2>&1 | grep default | tee -a $O/ab.log
