"""Phase timeline of the one-launch RANSAC kernel from in-kernel stamps (diagnostic build: tools/build_stamps.sh).
    PM_LIB_PATH=points_matching_amd/build/abl/libpm_rfstamps.so python tools/prof_ransac_stamps.py [n hyps cap]
Prints, over the workgroups of the last launch, the median / max cycles of each phase and the in-kernel clock."""
import ctypes as C
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
if os.environ.get("PM_RANSAC_FORM"):
    ctx.set_option(pm.api.PM_OPT_RANSAC_FORM, int(os.environ["PM_RANSAC_FORM"]))
for _ in range(20):
    ctx.ransac_run_dev(d1.data_ptr(), d2.data_ptr(), cap, dn.data_ptr(), 0, H, 1.0, 0x5EED, d_key.data_ptr(),
                       d_F.data_ptr(), d_mask.data_ptr(), d_ninl.data_ptr())
ctx.synchronize()
print("hipEvent mean us:", ctx.timing_get("ransac_fused")[0] * 1e3, "inliers", int(d_ninl.item()))
lib = pm.api.lib()
if hasattr(lib, "pm_debug_rf_stamps"):
    nwg = 4096
    buf = np.zeros(nwg * 12, np.uint64)
    lib.pm_debug_rf_stamps(buf.ctypes.data_as(C.c_void_p), buf.size)
    s = buf.reshape(nwg, 12).astype(np.int64)
    live = s[:, 0] != 0
    s = s[live]
    t0 = s[:, 0].min()
    names = ["offsets", "load tile", "solve", "barrier", "score", "key+slot+ticket", "scan slots (last)", "mask (last)"]
    print("workgroups:", s.shape[0], "first->last start spread (cycles):", int(s[:, 0].max() - t0))
    for i, nm in enumerate(names):
        a, b = s[:, i], s[:, i + 1]
        ok = (b != 0) & (a != 0) & (b >= a)
        if i >= 6:
            ok &= s[:, 8] != 0
        if ok.any():
            d = (b - a)[ok]
            print("%-22s median %7d  max %7d cycles  (%d wgs)" % (nm, np.median(d), d.max(), ok.sum()))
    end = s[:, 9]
    print("kernel span (first start -> last end): %d cycles" % int(end.max() - t0))
    rt = (s[:, 11] - s[:, 10])
    cyc = (s[:, 9] - s[:, 0])
    okc = rt > 0
    print("in-kernel clock: %.2f GHz (median over workgroups; s_memrealtime = 100 MHz)" % float(np.median(cyc[okc] / rt[okc] * 0.1)))
else:
    print("library has no stamps (build with tools/build_stamps.sh and set PM_LIB_PATH)")

if hasattr(lib, "pm_debug_rf_solve_stamps"):
    b2 = np.zeros(4096 * 8, np.uint64)
    lib.pm_debug_rf_solve_stamps(b2.ctypes.data_as(C.c_void_p), b2.size)
    t = b2.reshape(4096, 8).astype(np.int64)
    t = t[t[:, 5] != 0]
    order = [("sample8", 5, 6), ("load 8 points", 6, 0), ("hartley x2", 0, 1), ("QR 9x8", 1, 2), ("null vector", 2, 3),
             ("jacobi (6 sweeps)", 3, 4), ("rank-2 + denormalise", 4, 7)]
    print("solver phases of thread 0 (cycles, median over workgroups):")
    for nm, a, b in order:
        d = t[:, b] - t[:, a]
        print("  %-22s %7d" % (nm, np.median(d)))
