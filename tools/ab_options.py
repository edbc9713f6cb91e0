"""A/B timing of a context option on the matcher kernels (hipEvent means from the library's own brackets).
    python tools/ab_options.py <option id> <values,comma> [kind nq nt reps]
e.g. python tools/ab_options.py 6 1,2 sift 8192 8192 50      (PM_OPT_KNN_STAGING: registers vs LDS-DMA)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import points_matching_amd as pm  # noqa: E402
from points_matching_amd import synth  # noqa: E402

opt = int(sys.argv[1])
vals = [int(v) for v in sys.argv[2].split(",")]
kind = sys.argv[3] if len(sys.argv) > 3 else "sift"
nq = int(sys.argv[4]) if len(sys.argv) > 4 else 8192
nt = int(sys.argv[5]) if len(sys.argv) > 5 else 8192
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 50
dim = 32 if kind == "orb" else 128
w = synth.pair_workload(nq, nt, dim, seed=0xC3, kind=kind)
dev = torch.device("cuda", 0)
s = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(s)
ctx = pm.Context(0)
ctx.set_stream(s.cuda_stream)
d_q, d_t = torch.from_numpy(w["q"]).to(dev), torch.from_numpy(w["t"]).to(dev)
d_out = torch.empty((nq, 2, 4), dtype=torch.int32, device=dev)
flags = 4 if kind == "sift" else 0


def run():
    if kind == "orb":
        ctx.bf_knn_hamming_dev(d_q.data_ptr(), nq, d_t.data_ptr(), nt, dim, 2, d_out.data_ptr())
    else:
        ctx.bf_knn_l2_dev(d_q.data_ptr(), nq, d_t.data_ptr(), nt, dim, 2, d_out.data_ptr(), flags)


names = ("knn_l2_prep", "knn_l2_mfma_f16", "knn_l2_mfma", "knn_l2_refine", "knn_hamming_expand", "knn_hamming_mfma_i8", "knn_hamming_refine")
for rnd in range(2):
    for v in vals:
        ctx.set_option(opt, v)
        for _ in range(5):
            run()
        ctx.timing_enable(True)
        ctx.timing_reset()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(reps):
            run()
        e1.record(s)
        torch.cuda.synchronize()
        t = {k: round(ctx.timing_get(k)[0] * 1e3, 2) for k in names if ctx.timing_get(k)[1]}
        ctx.timing_enable(False)
        print("option %d = %d  %s %dx%d  call %.2f us  kernels %s" % (opt, v, kind, nq, nt, e0.elapsed_time(e1) / reps * 1e3, t))
