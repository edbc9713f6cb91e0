import sys; sys.path.insert(0,'.')
import numpy as np, torch
import points_matching_amd as pm
from points_matching_amd import synth
from oracle import pm_oracle as O
nq=nt=32768
w = synth.pair_workload(nq, nt, 32, seed=0xC4, kind="orb")
dev = torch.device("cuda",0)
ctx = pm.Context(0)
s = torch.cuda.Stream(device=dev); torch.cuda.set_stream(s); ctx.set_stream(s.cuda_stream)
d_q = torch.from_numpy(w["q"]).to(dev); d_t = torch.from_numpy(w["t"]).to(dev)
d_kp1 = torch.from_numpy(w["kp1"]).to(dev); d_kp2 = torch.from_numpy(w["kp2"]).to(dev)
d_knn = torch.empty((nq,2,4), dtype=torch.int32, device=dev)
d_good = torch.zeros((nq,4), dtype=torch.int32, device=dev)
d_xy1 = torch.zeros((nq,2), dtype=torch.float32, device=dev); d_xy2 = torch.zeros((nq,2), dtype=torch.float32, device=dev)
d_n = torch.zeros(1, dtype=torch.int32, device=dev)
torch.cuda.synchronize()
for it in range(3):
    ctx.bf_knn_hamming_dev(d_q.data_ptr(), nq, d_t.data_ptr(), nt, 32, 2, d_knn.data_ptr())
    ctx.filter_ratio_gather_dev(d_knn.data_ptr(), nq, 2, 0.8, d_kp1.data_ptr(), d_kp2.data_ptr(), d_good.data_ptr(), d_xy1.data_ptr(), d_xy2.data_ptr(), d_n.data_ptr())
    ctx.synchronize()
    knn = d_knn.cpu().numpy().view(pm.MATCH_DTYPE).reshape(nq,2)
    want = O.bf_knn_hamming(w["q"], w["t"], 2, nthreads=8)
    print("knn idx equal", (knn['trainIdx']==want['trainIdx']).all(), "qidx ok", (knn['queryIdx'][:,0]==np.arange(nq)).all())
    good_o = O.filter_ratio(want, 0.8)
    n = int(d_n.item()); good = d_good.cpu().numpy().view(pm.MATCH_DTYPE).reshape(-1)[:n]
    print("n", n, good_o.size, "good equal", n==good_o.size and (good['queryIdx']==good_o['queryIdx']).all() and (good['trainIdx']==good_o['trainIdx']).all())
    xy1 = O.gather_points(w['kp1'], good_o['queryIdx']); xy2 = O.gather_points(w['kp2'], good_o['trainIdx'])
    print("xy equal", (d_xy1.cpu().numpy()[:n]==xy1).all(), (d_xy2.cpu().numpy()[:n]==xy2).all())
    if not (good['queryIdx']==good_o['queryIdx']).all():
        bad = np.nonzero(good['queryIdx']!=good_o['queryIdx'])[0]; print("first bad", bad[:5], good['queryIdx'][bad[:5]], good_o['queryIdx'][bad[:5]])
x2 = d_xy2.cpu().numpy()[:n]
bad = np.nonzero((x2 != xy2).any(1))[0]
print("bad count", bad.size, "of", n, "first", bad[:6])
for i in bad[:6]:
    # which kp2 row did we get?
    hit = np.nonzero((w['kp2'] == x2[i]).all(1))[0]
    print(" i", i, "q", good_o['queryIdx'][i], "want train", good_o['trainIdx'][i], "got row(s)", hit[:3], "second nn", want['trainIdx'][good_o['queryIdx'][i],1])
print("got", x2[:4].tolist()); print("want", xy2[:4].tolist()); print("xy1", d_xy1.cpu().numpy()[:2].tolist())
print("kp2 dtype", w['kp2'].dtype, w['kp2'].flags['C_CONTIGUOUS'], "kp1", w['kp1'].dtype)
