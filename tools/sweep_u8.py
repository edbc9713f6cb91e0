"""Event-timed sweep of the u8 route's options at one shape (interleaved rounds in one process; medians).
    python tools/sweep_u8.py nq nt "12=1" "12=1 7=2" ...      (option=value pairs per configuration)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import points_matching_amd as pm
from points_matching_amd import synth
nq, nt = int(sys.argv[1]), int(sys.argv[2])
cfgs = sys.argv[3:]
w = synth.pair_workload(nq, nt, 128, seed=0xC3, kind="sift")
dev = torch.device("cuda", 0)
s = torch.cuda.Stream(device=dev); torch.cuda.set_stream(s)
ctx = pm.Context(0); ctx.set_stream(s.cuda_stream)
d_q, d_t = torch.from_numpy(w["q"]).to(dev), torch.from_numpy(w["t"]).to(dev)
d_out = torch.empty((nq, 2, 4), dtype=torch.int32, device=dev)
names = ("knn_l2_prep", "knn_l2_mfma_u8", "knn_l2_mfma_f16s", "knn_l2_mfma_f16", "knn_l2_refine")
res = {c: [] for c in cfgs}
ALL = range(1, 18)
for rnd in range(5):
    for c in cfgs:
        for o in ALL:
            ctx.set_option(o, 0)
        flags = pm.api.PM_KNN_HINT_U8
        for kv in c.split():
            k, v = kv.split("=")
            if k == "flags":
                flags = int(v)
            else:
                ctx.set_option(int(k), int(v))
        for _ in range(3):
            ctx.bf_knn_l2_dev(d_q.data_ptr(), nq, d_t.data_ptr(), nt, 128, 2, d_out.data_ptr(), flags)
        reps = 20
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(reps):
            ctx.bf_knn_l2_dev(d_q.data_ptr(), nq, d_t.data_ptr(), nt, 128, 2, d_out.data_ptr(), flags)
        e1.record(s)
        torch.cuda.synchronize()
        call = e0.elapsed_time(e1) / reps * 1e3
        ctx.timing_enable(True); ctx.timing_reset()
        for _ in range(reps):
            ctx.bf_knn_l2_dev(d_q.data_ptr(), nq, d_t.data_ptr(), nt, 128, 2, d_out.data_ptr(), flags)
        torch.cuda.synchronize()
        t = [ctx.timing_get(k)[0] * 1e3 for k in names]
        ctx.timing_enable(False)
        res[c].append([call] + t)
print("%dx%d: call (un-instrumented, us) | prep, coarse u8 / f16s / f16, refine (event-bracketed, us); medians of 5 rounds" % (nq, nt))
for c in cfgs:
    m = np.median(np.array(res[c]), axis=0)
    print("  %-28s call %7.2f | prep %6.2f coarse %7.2f %7.2f %7.2f refine %6.2f" % (c or "(default)", m[0], m[1], m[2], m[3], m[4], m[5]))
