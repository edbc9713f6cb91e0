#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
O=gpurun_out/r3af; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_knn_l2_gpu.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -8 $O/pytest.log
grep -q "rc=0" $O/pytest.log || exit 1
cat > /tmp/un.py <<'PY'
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch
import points_matching_amd as pm
from points_matching_amd import synth
nq = nt = 8192
w = synth.pair_workload(nq, nt, 128, seed=0xC3, kind="surf")
dev = torch.device("cuda", 0)
s = torch.cuda.Stream(device=dev); torch.cuda.set_stream(s)
ctx = pm.Context(0); ctx.set_stream(s.cuda_stream)
d_q, d_t = torch.from_numpy(w["q"]).to(dev), torch.from_numpy(w["t"]).to(dev)
d_out = torch.empty((nq, 2, 4), dtype=torch.int32, device=dev)
for rnd in range(2):
    for name, flags in (("automatic", 0), ("unit-norm hint", pm.api.PM_KNN_HINT_UNIT_NORM)):
        for _ in range(3): ctx.bf_knn_l2_dev(d_q.data_ptr(), nq, d_t.data_ptr(), nt, 128, 2, d_out.data_ptr(), flags)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(20): ctx.bf_knn_l2_dev(d_q.data_ptr(), nq, d_t.data_ptr(), nt, 128, 2, d_out.data_ptr(), flags)
        e1.record(s); torch.cuda.synchronize()
        ctx.timing_enable(True); ctx.timing_reset()
        for _ in range(10): ctx.bf_knn_l2_dev(d_q.data_ptr(), nq, d_t.data_ptr(), nt, 128, 2, d_out.data_ptr(), flags)
        ctx.synchronize()
        t = {k: round(ctx.timing_get(k)[0] * 1e3, 1) for k in ("knn_l2_prep", "knn_l2_mfma_f16", "knn_l2_refine")}
        ctx.timing_enable(False)
        print("SURF-like 8192 x 8192 x 128, %-16s matcher call %.1f us  %s" % (name, e0.elapsed_time(e1) * 50, t), flush=True)
PY
timeout -k 10 120 python /tmp/un.py 2>&1 | grep -v amdgpu.ids | tee $O/unit.log
