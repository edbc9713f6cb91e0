#!/bin/bash
# small shapes (the reference's own feature counts) through bench.py's C2 form, then a fuzz leg
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=gpurun_out/r3ag; mkdir -p $O
for n in 512 1024; do
  timeout -k 10 200 python bench.py --workload c2 --nq $n --nt $n --hyps 2000 --no-large --no-cpu-baseline --steps 200 --warmup 20 > $O/bench_c2_$n.json 2> $O/bench_c2_$n.err || exit 1
  python - <<PY
import json
d = json.loads(open("$O/bench_c2_$n.json").read().strip().splitlines()[-1])
print($n, "step ms", d["ms_per_step"], {k: d[k] for k in d if k.startswith("stage") or k in ("value", "sustained")})
PY
done
bash tools/gpu_fuzz.sh 270 101
