#!/bin/bash
# round-3 session B: seeded-route tests, stamps of the u8 ring kernel under its options, bench A/B
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=gpurun_out/r3b; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_knn_seeded_gpu.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -3 $O/pytest.log
S=points_matching_amd/build/abl/libpm_knnstamps.so
for o in "" "11=1" "11=3" "12=1" "4=2" "11=1 13=1" "11=3 4=2"; do
  echo "=== opts: $o" >> $O/stamps.log
  PM_LIB_PATH=$S timeout -k 10 120 python tools/prof_knn_stamps.py 8192 8192 $o 2>&1 | grep -v amdgpu.ids >> $O/stamps.log
done
for o in "" "11=3" "4=2" "11=3 4=2" "12=1"; do
  echo "=== 32k opts: $o" >> $O/stamps32k.log
  PM_LIB_PATH=$S timeout -k 10 120 python tools/prof_knn_stamps.py 32768 32768 $o 2>&1 | grep -v amdgpu.ids >> $O/stamps32k.log
done
echo "=== 2k" >> $O/stamps.log
PM_LIB_PATH=$S timeout -k 10 120 python tools/prof_knn_stamps.py 2048 2048 2>&1 | grep -v amdgpu.ids >> $O/stamps.log
timeout -k 10 200 python bench.py --hint u8 --headline-only --no-cpu-baseline --steps 50 > $O/bench_u8.json 2> $O/bench_u8.err
python - <<PY
import json
d=json.load(open("$O/bench_u8.json"))
print("u8", d["ms_per_step"], d["stage_ms"]["match"], d["kernels_us"], d["parity"], d["roofline"]["frac"])
PY
cat $O/stamps.log | head -150
