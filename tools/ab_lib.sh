#!/bin/bash
# same-session A/B of the product library against points_matching_amd/build/abl/libpm_<name>.so on the matcher call
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
alt=${1:-oldl2}; shift
for r in 1 2 3; do
  echo "== product"; timeout -k 10 200 python tools/ab_options.py 6 2 "$@" 2>&1 | grep -v amdgpu.ids | tail -1
  echo "== $alt"; PM_LIB_PATH=points_matching_amd/build/abl/libpm_$alt.so timeout -k 10 200 python tools/ab_options.py 6 2 "$@" 2>&1 | grep -v amdgpu.ids | tail -1
done
