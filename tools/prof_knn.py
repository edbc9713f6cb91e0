"""Profiling driver: a few launches of the L2 matcher alone (for rocprofv3 --pmc passes).
    rocprofv3 --pmc ... --kernel-trace --output-format csv -d out -- python3 tools/prof_knn.py [nq nt dim reps kind flags]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import points_matching_amd as pm  # noqa: E402
from points_matching_amd import synth  # noqa: E402

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
nt = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
dim = int(sys.argv[3]) if len(sys.argv) > 3 else 128
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
kind = sys.argv[5] if len(sys.argv) > 5 else "sift"
flags = int(sys.argv[6]) if len(sys.argv) > 6 else (8 if kind == "sift" else 0)      # 8 = PM_KNN_HINT_U8: the bench's route
w = synth.pair_workload(nq, nt, dim, seed=0xC3, kind=kind)
dev = torch.device("cuda", 0)
s = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(s)
ctx = pm.Context(0)
ctx.set_stream(s.cuda_stream)
d_q = torch.from_numpy(w["q"]).to(dev)
d_t = torch.from_numpy(w["t"]).to(dev)
d_out = torch.empty((nq, 2, 4), dtype=torch.int32, device=dev)
torch.cuda.synchronize()
for _ in range(3):
    ctx.bf_knn_l2_dev(d_q.data_ptr(), nq, d_t.data_ptr(), nt, dim, 2, d_out.data_ptr(), flags)
ctx.timing_enable(True)
ctx.timing_reset()
for _ in range(reps):
    ctx.bf_knn_l2_dev(d_q.data_ptr(), nq, d_t.data_ptr(), nt, dim, 2, d_out.data_ptr(), flags)
torch.cuda.synchronize()
print("done", nq, nt, dim, reps, kind, flags, os.environ.get("PM_LIB_PATH", "default"),
      {k: round(ctx.timing_get(k)[0] * 1e3, 1) for k in ("knn_l2_prep", "knn_l2_mfma_u8", "knn_l2_mfma_f16", "knn_l2_mfma", "knn_l2_refine")})
