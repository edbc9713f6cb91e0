"""Profiling driver: a few launches of the one-launch RANSAC-F kernel alone at config C3's shape (for rocprofv3 --pmc passes).
    rocprofv3 --pmc ... --kernel-trace --output-format csv -d out -- python3 tools/prof_ransac.py [n hyps cap reps]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import points_matching_amd as pm  # noqa: E402
from points_matching_amd import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2275
H = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
cap = int(sys.argv[3]) if len(sys.argv) > 3 else 8192
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 8
x1, x2, _, _ = synth.two_view(n, seed=0xC3, outlier_frac=0.3, noise_px=0.5)
dev = torch.device("cuda", 0)
b1 = np.zeros((cap, 2), np.float32); b1[:n] = x1
b2 = np.zeros((cap, 2), np.float32); b2[:n] = x2
d1, d2 = torch.from_numpy(b1).to(dev), torch.from_numpy(b2).to(dev)
dn = torch.tensor([n], dtype=torch.int32, device=dev)
d_key = torch.zeros(1, dtype=torch.int64, device=dev)
d_F = torch.zeros(9, dtype=torch.float64, device=dev)
d_mask = torch.zeros(cap, dtype=torch.uint8, device=dev)
d_ninl = torch.zeros(1, dtype=torch.int32, device=dev)
ctx = pm.Context(0)
ctx.timing_enable(True)
for _ in range(reps):
    ctx.ransac_run_dev(d1.data_ptr(), d2.data_ptr(), cap, dn.data_ptr(), 0, H, 1.0, 0x5EED, d_key.data_ptr(),
                       d_F.data_ptr(), d_mask.data_ptr(), d_ninl.data_ptr())
ctx.synchronize()
print("done", n, H, cap, reps, "ransac_fused mean us", round(ctx.timing_get("ransac_fused")[0] * 1e3, 2), "inliers", int(d_ninl.item()))
