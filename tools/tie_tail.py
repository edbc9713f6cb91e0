"""How the matcher behaves on tie-heavy data (runs of identical train rows: repeated texture): call time by route and the
number of queries that took the exact re-scan.      python tools/tie_tail.py [n dup_runs run_len]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import points_matching_amd as pm
from points_matching_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 40
ln = int(sys.argv[3]) if len(sys.argv) > 3 else 12
q, t, _ = synth.sift_like(n, n, 128, seed=5)
rng = np.random.default_rng(1)
for _ in range(runs):                       # a run of identical train rows, and a few queries equal to it
    a = int(rng.integers(0, n)); b = int(rng.integers(0, n - ln))
    t[b:b + ln] = t[a]
    q[rng.integers(0, n, 3)] = t[a]
dev = torch.device("cuda", 0)
st_ = torch.cuda.Stream(device=dev); torch.cuda.set_stream(st_)      # one stream for torch's events and the library
ctx = pm.Context(0); ctx.set_stream(st_.cuda_stream)
d_q, d_t = torch.from_numpy(q).to(dev), torch.from_numpy(t).to(dev)
d_out = torch.empty((n, 2, 4), dtype=torch.int32, device=dev)
for name, flags in (("u8 hint", pm.api.PM_KNN_HINT_U8), ("integer hint (f16 pass)", pm.api.PM_KNN_HINT_INTEGER)):
    ctx.knn_diag_enable(True)
    ctx.bf_knn_l2_dev(d_q.data_ptr(), n, d_t.data_ptr(), n, 128, 2, d_out.data_ptr(), flags)
    ctx.synchronize()
    st = ctx.knn_stats()
    ctx.knn_diag_enable(False)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        ctx.bf_knn_l2_dev(d_q.data_ptr(), n, d_t.data_ptr(), n, 128, 2, d_out.data_ptr(), flags)
    e0.record(st_)
    for _ in range(10):
        ctx.bf_knn_l2_dev(d_q.data_ptr(), n, d_t.data_ptr(), n, 128, 2, d_out.data_ptr(), flags)
    e1.record(st_); torch.cuda.synchronize()
    print("%d x %d, %d runs of %d identical train rows: %-24s call %.1f us, re-scanned queries %d" % (n, n, runs, ln, name, e0.elapsed_time(e1) * 100, st["rescans"]))

# ---- the same for the Hamming matcher (ORB-256): runs of identical train rows
qh, th, _ = synth.orb_like(n, n, 32, seed=6)
for _ in range(runs):
    a = int(rng.integers(0, n)); b = int(rng.integers(0, n - ln))
    th[b:b + ln] = th[a]
    qh[rng.integers(0, n, 3)] = th[a]
d_qh, d_th = torch.from_numpy(qh).to(dev), torch.from_numpy(th).to(dev)
for name, form in (("four queries per wave (default)", 0), ("one wave per query", 1)):
    ctx.set_option(pm.api.PM_OPT_HAMMING_REFINE, form)
    for _ in range(3):
        ctx.bf_knn_hamming_dev(d_qh.data_ptr(), n, d_th.data_ptr(), n, 32, 2, d_out.data_ptr())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st_)
    for _ in range(10):
        ctx.bf_knn_hamming_dev(d_qh.data_ptr(), n, d_th.data_ptr(), n, 32, 2, d_out.data_ptr())
    e1.record(st_); torch.cuda.synchronize()
    print("%d x %d ORB-256, %d runs of %d identical train rows: Hamming matcher, refinement %-32s call %.1f us" % (n, n, runs, ln, name, e0.elapsed_time(e1) * 100))
ctx.set_option(pm.api.PM_OPT_HAMMING_REFINE, 0)
