"""Times of the shapes that fall outside the matrix-core routes (k > 2, dim % 4 != 0, dim > 128, unaligned bases, or
PM_KNN_FORCE_EXACT take the exact VALU kernel) and of 7-point LMedS by correspondence count up to its 32 768 cap.
Prints one JSON object.   python tools/fallback_perf.py"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import points_matching_amd as pm  # noqa: E402
from points_matching_amd import synth  # noqa: E402

dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(stream)              # the events below must sit on the stream the library launches on
ctx = pm.Context(0)
ctx.set_stream(stream.cuda_stream)
out = {"knn_l2": [], "lmeds": []}


def time_knn(nq, nt, dim, k, flags, reps=5):
    q, t, _ = synth.surf_like(nq, nt, dim, seed=1)
    d_q, d_t = torch.from_numpy(q).to(dev), torch.from_numpy(t).to(dev)
    d_o = torch.empty((nq, k, 4), dtype=torch.int32, device=dev)
    for _ in range(2):
        ctx.bf_knn_l2_dev(d_q.data_ptr(), nq, d_t.data_ptr(), nt, dim, k, d_o.data_ptr(), flags)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(reps):
        ctx.bf_knn_l2_dev(d_q.data_ptr(), nq, d_t.data_ptr(), nt, dim, k, d_o.data_ptr(), flags)
    e1.record(stream)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    return {"nq": nq, "nt": nt, "dim": dim, "k": k, "flags": flags, "ms": round(ms, 4),
            "pairs_per_s": nq * nt / (ms * 1e-3), "valu_tflops": 3.0 * dim * nq * nt / (ms * 1e-3) / 1e12}


for (nq, nt, dim, k, flags, what) in [(8192, 8192, 128, 2, 0, "general floats, automatic route (f16 matrix pass on rounded copies)"),
                                      (8192, 8192, 128, 2, pm.api.PM_KNN_FORCE_EXACT, "exact VALU kernel forced"),
                                      (8192, 8192, 128, 3, 0, "k = 3 (round 3: matrix pass + 4-deep refinement; round 2: exact kernel)"),
                                      (8192, 8192, 128, 4, 0, "k = 4"),
                                      (8192, 8192, 130, 2, 0, "dim % 4 != 0 (round 3: f16 pass on 256-column padded copies, element loads)"),
                                      (8192, 8192, 256, 2, 0, "dim 256 (round 3: 17 k-chunks)"),
                                      (8192, 8192, 200, 3, 0, "dim 200, k = 3"),
                                      (8192, 8192, 64, 2, 0, "dim 64 (SURF-64), automatic route")]:
    r = time_knn(nq, nt, dim, k, flags)
    r["what"] = what
    out["knn_l2"].append(r)

for n in (2275, 8192, 32768):
    x1, x2, _, _ = synth.two_view(n, seed=5, outlier_frac=0.3, noise_px=0.5)
    iters = pm.api.lmeds_default_iters(0.99, 0.45)
    import time
    pm.api.lmeds_fundamental(ctx, x1, x2, iters, 7)
    t0 = time.perf_counter()
    for _ in range(3):
        rc = pm.api.lmeds_fundamental(ctx, x1, x2, iters, 7)
    dt = (time.perf_counter() - t0) / 3
    out["lmeds"].append({"n": n, "iters": iters, "ms_host_call": round(dt * 1e3, 3), "status": rc[0], "inliers": rc[3]})
print(json.dumps(out))
