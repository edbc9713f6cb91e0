"""flann_search timing (hipEvent, the library's own bracket) at the reference's scale and at C3's query count."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import points_matching_amd as pm
from points_matching_amd import synth
ctx = pm.Context(0)
for (nq, nt) in ((300, 300), (1000, 1000), (8192, 8192)):
    w = synth.pair_workload(nq, nt, 128, seed=5, kind="surf")
    ix = pm.api.FlannIndex(ctx, w["t"], trees=4, checks=32, seed=7)
    dev = torch.device("cuda", 0)
    d_q = torch.from_numpy(w["q"]).to(dev)
    d_o = torch.zeros((nq, 1, 4), dtype=torch.int32, device=dev)
    for _ in range(3):
        ix.knn_dev(d_q.data_ptr(), nq, 1, d_o.data_ptr())
    ctx.synchronize()
    ctx.timing_enable(True); ctx.timing_reset()
    for _ in range(20):
        ix.knn_dev(d_q.data_ptr(), nq, 1, d_o.data_ptr())
    ctx.synchronize()
    us = ctx.timing_get("flann_search")[0] * 1e3
    ctx.timing_enable(False)
    exact = ctx.bf_knn_l2(w["q"], w["t"], 1)
    got = d_o.cpu().numpy().view(pm.MATCH_DTYPE).reshape(nq, 1)
    print("flann_search %d queries x %d train rows (4 trees, 32 checks): %.1f us; recall@1 vs the exact matcher %.3f" % (
        nq, nt, us, float((got["trainIdx"] == exact["trainIdx"]).mean())))
