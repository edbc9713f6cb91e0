import sys, ctypes as C; sys.path.insert(0,'.')
import numpy as np, points_matching_amd as pm
ctx = pm.Context(0)
rng = np.random.default_rng(1)
nt, nq = 128, 64
t = rng.integers(0, 200, (nt, 128)).astype(np.float32)
perm = rng.permutation(nt)[:nq]
q = t[perm].copy()
got = ctx.bf_knn_l2(q,t,1,4)
buf = np.zeros(nq*8, np.float32)
n = pm.api.lib().pm_debug_copy(ctx._h, buf.ctypes.data_as(C.c_void_p), C.c_size_t(buf.nbytes))
print("bytes", n)
cand = buf.reshape(nq, 8)
w_true = (q @ t.T) - 0.5*(t*t).sum(1)[None,:]
for i in range(6):
    bits = cand[i].view(np.uint32)
    print("q",i,"want row",perm[i], "true w max", w_true[i].max(), "vals", cand[i], "ids", bits & 15)
