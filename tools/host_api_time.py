"""Wall time of the host-pointer entry points (what a drop-in caller of main.cpp:46 / :95-98 pays per call, copies and
allocations included) next to the device-pointer ones.   python tools/host_api_time.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import points_matching_amd as pm
from points_matching_amd import synth
ctx = pm.Context(0)
for (nq, nt) in ((2048, 2048), (8192, 8192)):
    w = synth.pair_workload(nq, nt, 128, seed=1, kind="sift")
    for name, fn in (("bf_knn_l2 host pointers", lambda: ctx.bf_knn_l2(w["q"], w["t"], 2, pm.api.PM_KNN_HINT_INTEGER)),):
        fn(); fn()
        t0 = time.perf_counter()
        for _ in range(20):
            fn()
        dt = (time.perf_counter() - t0) / 20
        print("%dx%d %s: %.3f ms per call (copies: %.1f MB in, %.2f MB out)" % (nq, nt, name, dt * 1e3, (nq + nt) * 512 / 1e6, nq * 32 / 1e6))
    knn = ctx.bf_knn_l2(w["q"], w["t"], 2, pm.api.PM_KNN_HINT_INTEGER)
    good = pm.api.filter_ratio(knn, 0.8)
    x1, x2 = w["kp1"][good["queryIdx"]], w["kp2"][good["trainIdx"]]
    ctx.ransac_fundamental(x1, x2, 10000, 1.0, 5)
    t0 = time.perf_counter()
    for _ in range(20):
        ctx.ransac_fundamental(x1, x2, 10000, 1.0, 5)
    print("ransac_fundamental host pointers, %d matches, 10000 hypotheses: %.3f ms per call" % (x1.shape[0], (time.perf_counter() - t0) / 20 * 1e3))
