"""debug: u8 route, ring vs two-buffer staging; where do the results differ from the f16 hint route?"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import points_matching_amd as pm
from points_matching_amd import synth
A = pm.api
ctx = pm.Context(0)
for (nq, nt) in ((700, 1000), (2048, 2048), (8192, 8192)):
    w = synth.pair_workload(nq, nt, 128, seed=nq + 3 * nt, planted=0.4, kind="sift")
    base = ctx.bf_knn_l2(w["q"], w["t"], 2, A.PM_KNN_HINT_INTEGER)
    for ring in (1, 2):
        for group in (1, 2, 3):
            for refine in ((1, 2) if group == 1 else (2,)):
                ctx.set_option(A.PM_OPT_KNN_RING, ring); ctx.set_option(A.PM_OPT_KNN_U8_GROUP, group); ctx.set_option(A.PM_OPT_KNN_U8_REFINE, refine)
                bad_rows = []
                for rep in range(4):
                    ctx.knn_diag_enable(True)
                    got = ctx.bf_knn_l2(w["q"], w["t"], 2, A.PM_KNN_HINT_U8)
                    st = ctx.knn_stats()
                    ctx.knn_diag_enable(False)
                    bad = np.nonzero((got["trainIdx"] != base["trainIdx"]).any(axis=1))[0]
                    bad_rows.append(bad)
                n = [len(b) for b in bad_rows]
                msg = ""
                if max(n):
                    b = bad_rows[int(np.argmax(n))]
                    tr = base["trainIdx"][b, 0]
                    msg = " q%%256 wave hist %s | true NN tile(of 128) hist %s | first bad q %s got %s want %s" % (
                        np.bincount((b % 256) // 32, minlength=8).tolist(), np.bincount(tr // 128)[:40].tolist(), b[:4].tolist(),
                        got["trainIdx"][b[:4]].tolist(), base["trainIdx"][b[:4]].tolist())
                print("%dx%d ring %d group %d refine %d: bad rows per rep %s stats %s%s" % (nq, nt, ring, group, refine, n, st, msg))
