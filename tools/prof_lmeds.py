"""Times the 7-point + LMedS estimator (device-resident form) on C3-sized correspondences."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C
import numpy as np
import torch
import points_matching_amd as pm
from points_matching_amd import synth
from points_matching_amd.api import LmedsParams, lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2275
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 300
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
x1, x2, _, _ = synth.two_view(n, seed=1, outlier_frac=0.3, noise_px=0.5)
dev = torch.device("cuda", 0)
s = torch.cuda.Stream(device=dev); torch.cuda.set_stream(s)
ctx = pm.Context(0); ctx.set_stream(s.cuda_stream)
d1 = torch.from_numpy(x1).to(dev); d2 = torch.from_numpy(x2).to(dev)
dF = torch.zeros(9, dtype=torch.float64, device=dev); dm = torch.zeros(n, dtype=torch.uint8, device=dev)
dn = torch.zeros(1, dtype=torch.int32, device=dev); db = torch.zeros(1, dtype=torch.int64, device=dev)
dmed = torch.zeros(1, dtype=torch.float64, device=dev)
prm = LmedsParams(0, iters, 7)
def run():
    rc = lib().pm_lmeds_fundamental_dev(ctx._h, C.c_void_p(d1.data_ptr()), C.c_void_p(d2.data_ptr()), n, C.byref(prm),
                                        C.c_void_p(dF.data_ptr()), C.c_void_p(dm.data_ptr()), C.c_void_p(dn.data_ptr()),
                                        C.c_void_p(db.data_ptr()), C.c_void_p(dmed.data_ptr()))
    assert rc == 0, rc
for _ in range(3): run()
torch.cuda.synchronize()
ctx.timing_enable(True); ctx.timing_reset()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record(s)
for _ in range(reps): run()
e1.record(s); torch.cuda.synchronize()
t = {k: round(ctx.timing_get(k)[0] * 1e3, 1) for k in ("lmeds_solve", "lmeds_median", "lmeds_final")}
print("lmeds n=%d iters=%d: %.1f us per run (wall, incl. event overhead)" % (n, iters, e0.elapsed_time(e1) / reps * 1e3), t,
      "inliers", int(dn.item()), "best", int(db.item()), "median", float(dmed.item()))
