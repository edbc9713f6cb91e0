"""Times the Hamming matcher alone (C4 matcher: 32k x 32k ORB-256)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import points_matching_amd as pm
from points_matching_amd import synth
nq = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
nt = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
w = synth.pair_workload(nq, nt, 32, seed=0xC4, kind="orb")
dev = torch.device("cuda", 0)
s = torch.cuda.Stream(device=dev); torch.cuda.set_stream(s)
ctx = pm.Context(0); ctx.set_stream(s.cuda_stream)
d_q = torch.from_numpy(w["q"]).to(dev); d_t = torch.from_numpy(w["t"]).to(dev)
d_out = torch.empty((nq, 2, 4), dtype=torch.int32, device=dev)
torch.cuda.synchronize()
for _ in range(2):
    ctx.bf_knn_hamming_dev(d_q.data_ptr(), nq, d_t.data_ptr(), nt, 32, 2, d_out.data_ptr())
ctx.timing_enable(True); ctx.timing_reset()
for _ in range(reps):
    ctx.bf_knn_hamming_dev(d_q.data_ptr(), nq, d_t.data_ptr(), nt, 32, 2, d_out.data_ptr())
torch.cuda.synchronize()
t = {k: ctx.timing_get(k)[0] for k in ("knn_hamming_expand", "knn_hamming_mfma_i8", "knn_hamming_refine", "knn_hamming",
                                       "knn_hamming_merge")}
print("hamming", nq, nt, {k: round(v * 1e3, 1) for k, v in t.items()}, "pairs/s %.3e" % (nq * nt / (sum(t.values()) * 1e-3)))
