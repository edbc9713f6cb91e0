#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=gpurun_out/r3j; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_knn_wide_gpu.py tests/test_knn_l2_gpu.py tests/test_knn_seeded_gpu.py tests/test_cli_gpu.py tests/test_independent_gpu.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -8 $O/pytest.log
timeout -k 10 300 python tools/fallback_perf.py > $O/fallback_perf.json 2> $O/fallback_perf.err; echo "fallback rc=$?"
python - <<PY
import json
d = json.load(open("$O/fallback_perf.json"))
for r in d["knn_l2"]: print("%5dx%5d dim %3d k %d: %8.4f ms  %s" % (r["nq"], r["nt"], r["dim"], r["k"], r["ms"], r["what"]))
PY
