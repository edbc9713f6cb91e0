"""hipGraph replay of one image pair's step (matcher + ratio filter + RANSAC, all device-resident calls of the C ABI)
against the same calls enqueued directly: time per step, and whether a REPLAY is valid at all — the launches carry
host-incremented epoch arguments (they stand in for per-call memsets of the side-band words), which a captured graph
freezes.  The record under profiles/ was taken before those calls learned to refuse a capturing stream (PM_REFUSE_CAPTURE);
with the library as it is now this prints "capture refused".      python tools/graph_step.py [n hyps]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import points_matching_amd as pm
from points_matching_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
H = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(device=dev); torch.cuda.set_stream(st)
ctx = pm.Context(0); ctx.set_stream(st.cuda_stream)
flags = pm.api.PM_KNN_HINT_U8


def upload(seed):
    w = synth.pair_workload(n, n, 128, seed=seed, kind="sift")
    return [torch.from_numpy(np.ascontiguousarray(w[k])).to(dev) for k in ("q", "t", "kp1", "kp2")]


d_q, d_t, d_kp1, d_kp2 = upload(0xC2)
alt = upload(0xD7)
knn = torch.empty((n, 2, 4), dtype=torch.int32, device=dev); good = torch.empty((n, 4), dtype=torch.int32, device=dev)
cnt = torch.zeros(4, dtype=torch.int32, device=dev)
xy1 = torch.empty((n, 2), dtype=torch.float32, device=dev); xy2 = torch.empty((n, 2), dtype=torch.float32, device=dev)
key = torch.zeros(1, dtype=torch.int64, device=dev); F = torch.zeros(9, dtype=torch.float64, device=dev)
mask = torch.zeros(n, dtype=torch.uint8, device=dev); ninl = torch.zeros(1, dtype=torch.int32, device=dev)


def step():
    ctx.bf_knn_l2_ratio_dev(d_q.data_ptr(), n, d_t.data_ptr(), n, 128, flags, 0.8, d_kp1.data_ptr(), d_kp2.data_ptr(),
                            knn.data_ptr(), good.data_ptr(), xy1.data_ptr(), xy2.data_ptr(), cnt.data_ptr())
    ctx.ransac_run_dev(xy1.data_ptr(), xy2.data_ptr(), n, cnt.data_ptr(), 0, H, 1.0, 0x5EED, key.data_ptr(), F.data_ptr(),
                       mask.data_ptr(), ninl.data_ptr())


def result():
    torch.cuda.synchronize()
    return (int(cnt[0]), int(key[0]), int(ninl[0]), F.cpu().numpy().tobytes(), knn.cpu().numpy().tobytes(), mask.cpu().numpy().tobytes())


def timed(fn, reps=300):
    for _ in range(20):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps):
        fn()
    e1.record(st); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


step(); base = result()
t_direct = timed(step)
print("%d x %d SIFT-128 + %d hypotheses: direct launches %.2f us per step (matches %d, inliers %d)" % (n, n, H, t_direct, base[0], base[2]))
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g, stream=st):
        step()
except Exception as e:                                    # noqa: BLE001
    print("capture refused:", str(e).splitlines()[0][:300])
    sys.exit(0)
torch.cuda.set_stream(st)
g.replay(); same = result() == base
t_graph = timed(g.replay)
print("graph replay %.2f us per step; replay on the SAME inputs equal to the direct calls: %s" % (t_graph, same))
for dst, src in zip((d_q, d_t, d_kp1, d_kp2), alt):       # other descriptors in the same buffers, then replay
    dst.copy_(src)
torch.cuda.synchronize()
bad = 0
for _ in range(20):
    g.replay(); r_graph = result()
    step(); r_direct = result()
    bad += r_graph != r_direct
print("replay after the inputs changed: %d of 20 replays differ from the direct calls on the same inputs" % bad)
