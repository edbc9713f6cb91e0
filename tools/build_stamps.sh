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
# coarse kNN kernels with stamps (u8 ring kernel): knn_coarse.o swapped for tools/ablation/knn_coarse_stamps.hip
objs2=$(ls $B/*.o | grep -v '/knn_coarse\.o$')
/opt/rocm/bin/hipcc $F -ffinite-math-only -x hip -c tools/ablation/knn_coarse_stamps.hip -o /tmp/knn_stamps.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $B/abl/libpm_knnstamps.so /tmp/knn_stamps.o $objs2 -ldl
echo built $B/abl/libpm_knnstamps.so
# timing-only ablations of the same (outputs wrong): no selection / no LDS operand reads / neither / no barrier
for v in "NO_EPI" "NO_LDSREAD" "NO_EPI NO_LDSREAD" "NO_BARRIER" "NO_EPI NO_LDSREAD NO_BARRIER NO_STAGE"; do
  name=$(echo "$v" | tr -d '_' | tr ' ' '_')
  defs=""; for w in $v; do defs="$defs -DABL_$w=1"; done
  /opt/rocm/bin/hipcc $F -ffinite-math-only $defs -x hip -c tools/ablation/knn_coarse_stamps.hip -o /tmp/knn_stamps_v.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $B/abl/libpm_knnstamps_$name.so /tmp/knn_stamps_v.o $objs2 -ldl
  echo built $B/abl/libpm_knnstamps_$name.so
done
