#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=gpurun_out/r3i; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_mgpu_gpu.py tests/test_pair_batch_gpu.py tests/test_pipeline_gpu.py tests/test_bench_gpu.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -5 $O/pytest.log
for v in "--c5-desc f32 --c5-layout block" "--c5-desc f32 --c5-layout separate" "--c5-desc u8 --c5-layout block" "--c5-desc u8 --c5-layout block --lanes 4" "--c5-desc f32 --c5-layout block --lanes 2"; do
  n=$(echo $v | tr -d ' -')
  timeout -k 10 300 python bench.py --workload c5 --steps 10 --warmup 10 $v > $O/c5_$n.json 2> $O/c5_$n.err; echo "c5 $v rc=$?"
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/c5_*.json")):
    try:
        d = json.load(open(f))
        print(f.split("/")[-1], "pairs/s %.0f  ms/batch %.2f  h2d GB/s %.1f  parity %s" % (d["image_pairs_per_s"], d["ms_per_step"], d["pcie"]["h2d_GBps"], d["parity"]))
    except Exception as e:
        print(f, "failed", e)
PY
