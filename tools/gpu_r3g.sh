#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=gpurun_out/r3g; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_mgpu_gpu.py tests/test_pipeline_gpu.py tests/test_bench_gpu.py tests/test_pair_batch_gpu.py tests/test_cli_gpu.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -15 $O/pytest.log
timeout -k 10 300 python tools/mgpu_stream_bench.py 1 2>&1 | grep -v amdgpu.ids | tee $O/mgpu_stream.log
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 50 > $O/bench_c3.json 2> $O/bench_c3.err; echo "bench rc=$?"
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 50 --exercise-exchange --headline-only > $O/bench_c3_exch.json 2> $O/bench_c3_exch.err; echo "bench exch rc=$?"
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 50 --pipeline 1 --headline-only > $O/bench_c3_pipe.json 2> $O/bench_c3_pipe.err; echo "bench pipe rc=$?"
python - <<PY
import json
for f in ("bench_c3", "bench_c3_exch", "bench_c3_pipe"):
    try:
        d = json.load(open("$O/%s.json" % f))
        print(f, "ms/step %.4f serial %.4f pipe %.4f | match %.4f rest %.4f | %s | parity %s | sustained %s" % (
            d["ms_per_step"], d["ms_per_step_serial"], d["ms_per_step_pipelined"], d["stage_ms"]["match"], d["stage_ms"]["ransac_and_exchange"],
            d["kernels_us"], d["parity"], d.get("sustained")))
        for k in ("collectives", "roofline", "roofline_large", "value_auto_route", "value_general_floats", "roofline_ransac"):
            if k in d: print("   ", k, d[k])
    except Exception as e:
        print(f, "failed:", e); print(open("$O/%s.err" % f).read()[-2000:])
PY
