"""Throughput of the streamed multi-GPU entry points of the C ABI (pm_mgpu_set_train + pm_mgpu_submit_dev / pm_mgpu_collect)
with the devices this box has: image pairs per second at 1, 2, 3 lanes per device, the collective's latency by itself,
and the blocking host-pointer call (pinned staging) beside them.
    python tools/mgpu_stream_bench.py [n_dev] [nq nt hyps pairs]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import points_matching_amd as pm  # noqa: E402
from points_matching_amd import synth  # noqa: E402

n_dev = int(sys.argv[1]) if len(sys.argv) > 1 else 1
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
nt = int(sys.argv[3]) if len(sys.argv) > 3 else 8192
H = int(sys.argv[4]) if len(sys.argv) > 4 else 10000
P = int(sys.argv[5]) if len(sys.argv) > 5 else 400
w = synth.pair_workload(nq * n_dev, nt, 128, seed=0xC3, kind="sift")
mg = pm.api.MultiGpu(n_dev)
flags = pm.api.PM_KNN_HINT_U8
t0 = time.perf_counter()
for _ in range(3):
    base = mg.match_ransac(w["q"], w["t"], w["kp1"], w["kp2"], 0.8, H, 1.0, 0x5EED, knn_flags=flags)
t_block = (time.perf_counter() - t0) / 3
print("n_dev %d  %d x %d per device, %d hypotheses: blocking pm_mgpu_match_ransac from host memory (pinned staging, train set "
      "uploaded per call) %.3f ms per pair; matches %d inliers %d" % (n_dev, nq, nt, H, t_block * 1e3, base[1].size, base[4]))
print("all-gather latency by itself: 80 B %.2f us, %d B %.2f us per collective%s" % (
    mg.allgather_latency(80), 16 + nq * 32, mg.allgather_latency(16 + nq * 32),
    " (ONE device: RCCL's in-place all-gather of a single rank launches nothing, this is the enqueue cost; bench.py "
    "--exercise-exchange times a torch.distributed collective at world 1: ~10-12 us)" if n_dev == 1 else ""))
mg.set_train(w["t"], w["kp2"])
dq, dk, rows = [], [], []
for g in range(n_dev):
    dev = torch.device("cuda", g)
    dq.append(torch.from_numpy(np.ascontiguousarray(w["q"][g * nq:(g + 1) * nq])).to(dev))
    dk.append(torch.from_numpy(np.ascontiguousarray(w["kp1"][g * nq:(g + 1) * nq])).to(dev))
    rows.append(nq)
    torch.cuda.synchronize(g)
qp, kp = [t.data_ptr() for t in dq], [t.data_ptr() for t in dk]
for lanes in (1, 2, 3):
    mg.set_lanes(lanes)
    for rep in range(2):
        t0 = time.perf_counter()
        tickets = []
        ok = True
        for i in range(P):
            tickets.append(mg.submit_dev(qp, rows, kp, 0.8, H, 1.0, 0x5EED, knn_flags=flags))
            if len(tickets) >= lanes:                       # a lane holds one pair until it is collected
                r, _, _ = mg.collect(tickets.pop(0))
                ok = ok and r.best_key == base[5] and r.n_inliers == base[4]
        for t in tickets:
            r, _, _ = mg.collect(t)
            ok = ok and r.best_key == base[5] and r.n_inliers == base[4]
        dt = time.perf_counter() - t0
    print("lanes %d: %d pairs in %.2f ms = %.1f us per pair = %.0f pairs/s; results %s" % (
        lanes, P, dt * 1e3, dt / P * 1e6, P / dt, "equal to the blocking call" if ok else "MISMATCH"))
mg.close()
