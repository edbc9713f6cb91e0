"""pm_bf_knn_l2_ratio_dev on small shapes: the filter as its own launch (PM_OPT_FILTER_FUSION = 1) against the filter riding
the refinement launch (2), u8 hint and integer hint.      python tools/small_fusion.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import points_matching_amd as pm
from points_matching_amd import synth

dev = torch.device("cuda", 0)
st = torch.cuda.Stream(device=dev); torch.cuda.set_stream(st)
ctx = pm.Context(0); ctx.set_stream(st.cuda_stream)
for n in (128, 256, 512, 1024, 2048, 4096, 8192):
    w = synth.pair_workload(n, n, 128, seed=3, kind="sift")
    d = {k: torch.from_numpy(np.ascontiguousarray(w[k])).to(dev) for k in ("q", "t", "kp1", "kp2")}
    knn = torch.empty((n, 2, 4), dtype=torch.int32, device=dev); good = torch.empty((n, 4), dtype=torch.int32, device=dev)
    cnt = torch.zeros(4, dtype=torch.int32, device=dev)
    xy1 = torch.empty((n, 2), dtype=torch.float32, device=dev); xy2 = torch.empty((n, 2), dtype=torch.float32, device=dev)
    line = "%5d x %-5d" % (n, n)
    for hname, flags in (("u8", pm.api.PM_KNN_HINT_U8), ("int", pm.api.PM_KNN_HINT_INTEGER)):
        res = []
        for form in (1, 2):
            ctx.set_option(pm.api.PM_OPT_FILTER_FUSION, form)
            def call():
                ctx.bf_knn_l2_ratio_dev(d["q"].data_ptr(), n, d["t"].data_ptr(), n, 128, flags, 0.8, d["kp1"].data_ptr(), d["kp2"].data_ptr(),
                                        knn.data_ptr(), good.data_ptr(), xy1.data_ptr(), xy2.data_ptr(), cnt.data_ptr())
            for _ in range(20):
                call()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for _ in range(400):
                call()
            e1.record(st); torch.cuda.synchronize()
            res.append((e0.elapsed_time(e1) / 400 * 1e3, int(cnt[0]), good[:int(cnt[0])].cpu().numpy().tobytes()))
        assert res[0][1:] == res[1][1:]
        line += "   %s hint: two launches %.2f us, fused %.2f us" % (hname, res[0][0], res[1][0])
    print(line, flush=True)
ctx.set_option(pm.api.PM_OPT_FILTER_FUSION, 0)
