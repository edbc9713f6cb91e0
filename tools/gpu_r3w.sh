#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=gpurun_out/r3w; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_knn_hamming_gpu.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -3 $O/pytest.log
grep -q "rc=0" $O/pytest.log || exit 1
cat > /tmp/hr.py <<'PY'
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch
import points_matching_amd as pm
from points_matching_amd import synth
for nq, nt in ((32768, 32768), (8192, 8192), (2048, 2048)):
    w = synth.pair_workload(nq, nt, 32, seed=0xC4, kind="orb")
    dev = torch.device("cuda", 0)
    ctx = pm.Context(0)
    d_q = torch.from_numpy(w["q"]).to(dev); d_t = torch.from_numpy(w["t"]).to(dev)
    d_out = torch.empty((nq, 2, 4), dtype=torch.int32, device=dev)
    for form in (1, 2):
        ctx.set_option(pm.api.PM_OPT_HAMMING_REFINE, form)
        for _ in range(2):
            ctx.bf_knn_hamming_dev(d_q.data_ptr(), nq, d_t.data_ptr(), nt, 32, 2, d_out.data_ptr())
        ctx.timing_enable(True); ctx.timing_reset()
        for _ in range(5):
            ctx.bf_knn_hamming_dev(d_q.data_ptr(), nq, d_t.data_ptr(), nt, 32, 2, d_out.data_ptr())
        ctx.synchronize()
        print(nq, nt, "refine form", form, "refine us %.1f" % (ctx.timing_get("knn_hamming_refine")[0] * 1e3), flush=True)
        ctx.timing_enable(False)
PY
timeout -k 10 200 python /tmp/hr.py 2>&1 | grep -v amdgpu.ids | tee -a $O/hr.log
