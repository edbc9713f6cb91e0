#!/bin/bash
# Diagnostic build of the one-launch RANSAC kernel with in-kernel phase stamps; never shipped.  The library is the
# product objects with ransac_fused.o swapped for tools/ablation/ransac_fused_stamps.hip (same kernel body, stamping
# policy).  Output: points_matching_amd/build/abl/libpm_rfstamps.so; use with
# PM_LIB_PATH=... python tools/prof_ransac_stamps.py
set -e
cd "$(dirname "$0")/.."
python -m points_matching_amd.build > /dev/null
B=points_matching_amd/build
mkdir -p $B/abl
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -Iinclude -Ipoints_matching_amd/csrc -I/opt/rocm/include"
objs=$(ls $B/*.o | grep -v '/ransac_fused\.o$')
/opt/rocm/bin/hipcc $F -x hip -c tools/ablation/ransac_fused_stamps.hip -o /tmp/rf_stamps.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $B/abl/libpm_rfstamps.so /tmp/rf_stamps.o $objs -ldl
echo built $B/abl/libpm_rfstamps.so
# experiment: unpacked v_fma_f32 scoring instead of v_pk_fma_f32 (same bits)
/opt/rocm/bin/hipcc $F -DRF_SCALAR_FMA=1 -x hip -c tools/ablation/ransac_fused_stamps.hip -o /tmp/rf_stamps2.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $B/abl/libpm_rfstamps_scalar.so /tmp/rf_stamps2.o $objs -ldl
echo built $B/abl/libpm_rfstamps_scalar.so
