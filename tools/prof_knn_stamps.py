"""Timeline of the u8 ring coarse kernel from in-kernel stamps (diagnostic build: tools/build_stamps.sh).
    PM_LIB_PATH=points_matching_amd/build/abl/libpm_knnstamps.so python tools/prof_knn_stamps.py [nq nt]
Prints, over the workgroups of the last launch: dispatch spread, cycles to the first tile, per-tile cycles, tail."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import points_matching_amd as pm  # noqa: E402
from points_matching_amd import synth  # noqa: E402

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
nt = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
opts = [a for a in sys.argv[3:]]
w = synth.pair_workload(nq, nt, 128, seed=0xC3, kind="sift")
dev = torch.device("cuda", 0)
ctx = pm.Context(0)
for o in opts:                       # option=value pairs
    k, v = o.split("=")
    ctx.set_option(int(k), int(v))
d_q, d_t = torch.from_numpy(w["q"]).to(dev), torch.from_numpy(w["t"]).to(dev)
d_out = torch.empty((nq, 2, 4), dtype=torch.int32, device=dev)
ctx.timing_enable(True)
for _ in range(30):
    ctx.bf_knn_l2_dev(d_q.data_ptr(), nq, d_t.data_ptr(), nt, 128, 2, d_out.data_ptr(), pm.api.PM_KNN_HINT_U8)
ctx.synchronize()
print("options", opts, "hipEvent mean us: coarse %.2f prep %.2f refine %.2f" % tuple(
    ctx.timing_get(n)[0] * 1e3 for n in ("knn_l2_mfma_u8", "knn_l2_prep", "knn_l2_refine")))
lib = pm.api.lib()
if not hasattr(lib, "pm_debug_knn_stamps"):
    print("library has no stamps (build with tools/build_stamps.sh and set PM_LIB_PATH)")
    sys.exit(0)
buf = np.zeros(4096 * 24, np.uint64)
lib.pm_debug_knn_stamps(buf.ctypes.data_as(C.c_void_p), buf.size)
s = buf.reshape(4096, 24).astype(np.int64)
s = s[s[:, 0] != 0]
if s.shape[0] == 0:
    print("this coarse-kernel form carries no stamps (the ring forms 12=2 / 12=3 and the register-operand forms 12=5 / 12=6 do)")
    sys.exit(0)
rega = any(o.startswith("12=") and int(o[3:]) >= 4 for o in opts)
rt0, rt1 = s[:, 20], s[:, 21]
print("workgroups %d; dispatch spread (first -> last entry): %.2f us; kernel span entry(first) -> exit(last): %.2f us" % (
    s.shape[0], (rt0.max() - rt0.min()) / 100.0, (rt1.max() - rt0.min()) / 100.0))
clk = np.median((s[:, 17] - s[:, 0]) / np.maximum(rt1 - rt0, 1) * 0.1)
print("in-kernel clock %.2f GHz; per-workgroup lifetime median %.2f us max %.2f us" % (
    clk, np.median(rt1 - rt0) / 100.0, (rt1 - rt0).max() / 100.0))
def show(name, a, b):
    d = (s[:, b] - s[:, a])
    ok = (s[:, a] != 0) & (s[:, b] != 0)
    if ok.any():
        d = d[ok]
        print("  %-34s median %7d  p90 %7d  max %7d cycles" % (name, np.median(d), np.percentile(d, 90), d.max()))
show("entry -> %s" % ("query fragments loaded" if rega else "requests issued"), 0, 1)
if not rega:
    show("requests issued -> tile 0 ready", 1, 2)
    ntl = int(((s[0, 2:16]) != 0).sum())
    for t in range(1, ntl):
        show("tile %d (barrier to barrier)" % (t - 1), 1 + t, 2 + t)
    show("last tile + final selection", 1 + ntl, 16)
show("merge + store", 16, 17)
xcc = s[:, 22] & 15
print("workgroups per XCC:", np.bincount(xcc.astype(int), minlength=8).tolist())
if rega:
    print("register-operand form: wave 0's 5th..7th block (wait | ds_read+issue | MFMA chain issued | selection issued)")
    for k in range(3):
        b0 = 2 + 4 * k
        show("  block %d: own pieces landed, A reads issued" % (4 + k), b0, b0 + 1)
        show("  block %d: 16 MFMAs issued (incl. A data)" % (4 + k), b0 + 1, b0 + 2)
        show("  block %d: selection issued" % (4 + k), b0 + 2, b0 + 3)
        if k < 2:
            show("  block %d end -> next block entry" % (4 + k), b0 + 3, b0 + 4)
    show("whole sweep (entry -> done)", 1, 16)
