#!/bin/bash
# C5 with u8 rows: the filter as its own launch against the filter inside the refinement (one launch less per pair; the
# batch is host-bound at 33-40k pairs/s)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=gpurun_out/r3ak; mkdir -p $O
for rep in 1 2; do for f in 1 2; do
  timeout -k 10 200 python bench.py --workload c5 --c5-desc u8 --c5-filter-fusion $f --steps 8 --warmup 5 --no-cpu-baseline > $O/c5_u8_fusion${f}_$rep.json 2> $O/err.txt || exit 1
  python - <<PY
import json
d = json.loads(open("$O/c5_u8_fusion${f}_$rep.json").read().strip().splitlines()[-1])
print("u8 rows, fusion $f, rep $rep:", round(d["value"]), d["unit"], round(d["ms_per_step"], 3), "ms per batch", d.get("parity"))
PY
done; done
for f in 1 2; do
  timeout -k 10 200 python bench.py --workload c5 --c5-filter-fusion $f --steps 5 --warmup 5 --no-cpu-baseline > $O/c5_f32_fusion$f.json 2> $O/err.txt || exit 1
  python - <<PY
import json
d = json.loads(open("$O/c5_f32_fusion$f.json").read().strip().splitlines()[-1])
print("f32 rows, fusion $f:", round(d["value"]), d["unit"], round(d["ms_per_step"], 3), "ms per batch", d.get("parity"))
PY
done
