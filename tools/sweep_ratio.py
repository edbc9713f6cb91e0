"""Matcher STAGE (pm_bf_knn_l2_ratio_dev: matcher + ratio test + compaction + gather), un-instrumented, interleaved rounds:
filter as its own launch (PM_OPT_FILTER_FUSION = 1) against the filter inside the refinement launch (2), hints u8 / integer.
    python tools/sweep_ratio.py nq nt"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import points_matching_amd as pm
from points_matching_amd import synth
nq, nt = int(sys.argv[1]), int(sys.argv[2])
w = synth.pair_workload(nq, nt, 128, seed=0xC3, kind="sift")
dev = torch.device("cuda", 0)
s = torch.cuda.Stream(device=dev); torch.cuda.set_stream(s)
ctx = pm.Context(0); ctx.set_stream(s.cuda_stream)
d_q, d_t = torch.from_numpy(w["q"]).to(dev), torch.from_numpy(w["t"]).to(dev)
d_k1, d_k2 = torch.from_numpy(w["kp1"]).to(dev), torch.from_numpy(w["kp2"]).to(dev)
d_knn = torch.empty((nq, 2, 4), dtype=torch.int32, device=dev)
d_good = torch.empty((nq, 4), dtype=torch.int32, device=dev)
d_x1 = torch.empty((nq, 2), dtype=torch.float32, device=dev); d_x2 = torch.empty((nq, 2), dtype=torch.float32, device=dev)
d_n = torch.zeros(1, dtype=torch.int32, device=dev)
cfgs = [("u8 hint, separate filter", 8, 1), ("u8 hint, fused", 8, 2), ("integer hint, separate", 4, 1), ("integer hint, fused", 4, 2)]
res = {c[0]: [] for c in cfgs}
for rnd in range(5):
    for name, flags, fus in cfgs:
        ctx.set_option(pm.api.PM_OPT_FILTER_FUSION, fus)
        def run():
            ctx.bf_knn_l2_ratio_dev(d_q.data_ptr(), nq, d_t.data_ptr(), nt, 128, flags, 0.8, d_k1.data_ptr(), d_k2.data_ptr(), d_knn.data_ptr(),
                                    d_good.data_ptr(), d_x1.data_ptr(), d_x2.data_ptr(), d_n.data_ptr())
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(20):
            run()
        e1.record(s)
        torch.cuda.synchronize()
        res[name].append((e0.elapsed_time(e1) / 20 * 1e3, int(d_n.item())))
print("%d x %d matcher stage (us per call, median of 5 rounds; survivors)" % (nq, nt))
for name, _, _ in cfgs:
    print("  %-28s %7.2f  n_good %d" % (name, float(np.median([r[0] for r in res[name]])), res[name][0][1]))
