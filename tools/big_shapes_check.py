"""Large / lopsided shapes against the oracle on the GPU box (not part of the test suite: ~1 minute of
CPU oracle time).  python tools/big_shapes_check.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import points_matching_amd as pm
from points_matching_amd import synth
from oracle import pm_oracle as O
sys.path.insert(0, os.path.join(ROOT, "tests"))
from util import assert_matches_equal
O.build()
ctx = pm.Context(0)
rng = np.random.default_rng(1)
U8 = pm.api.PM_KNN_HINT_U8
for (nq, nt, kind, flags) in [(600, 300000, "sift", 0), (600, 300000, "surf", 0), (70000, 1500, "sift", 4), (300, 70000, "sift", 4),
                              (600, 300000, "sift", U8), (70000, 1500, "sift", U8), (300, 70000, "sift", U8),
                              (20000, 33000, "sift", U8)]:                       # (round 3: the u8 route at the same sizes)
    q, t, _ = (synth.sift_like if kind == "sift" else synth.surf_like)(nq, nt, 128, seed=nq + nt)
    t0 = time.time(); got = ctx.bf_knn_l2(q, t, 2, flags); t1 = time.time()
    want = O.bf_knn_l2(q, t, 2, nthreads=16)
    assert_matches_equal(got, want, str((nq, nt, kind, flags)))
    print("L2", nq, nt, kind, "flags", flags, "ok", round(t1 - t0, 3), "s incl. copies")
    if flags == U8:
        assert_matches_equal(ctx.bf_knn_l2_u8(q.astype(np.uint8), t.astype(np.uint8), 2), want, str((nq, nt, "u8 rows")))
        for form in (5, 6):                                                       # register-operand coarse forms
            ctx.set_option(pm.api.PM_OPT_KNN_RING, form)
            try:
                assert_matches_equal(ctx.bf_knn_l2(q, t, 2, flags), want, str((nq, nt, "coarse form", form)))
            finally:
                ctx.set_option(pm.api.PM_OPT_KNN_RING, 0)
        print("   u8 rows and coarse forms 5, 6 ok")
q, t, _ = synth.orb_like(500, 300000, 32, seed=5)
assert_matches_equal(ctx.bf_knn_hamming(q, t, 2), O.bf_knn_hamming(q, t, 2, nthreads=16), "ham big nt")
q, t, _ = synth.orb_like(100000, 700, 32, seed=6)
assert_matches_equal(ctx.bf_knn_hamming(q, t, 2), O.bf_knn_hamming(q, t, 2, nthreads=16), "ham big nq")
print("hamming big ok")
x1, x2, _, _ = synth.two_view(30000, seed=3, outlier_frac=0.4, noise_px=0.7)
g = ctx.ransac_fundamental(x1, x2, 3000, 1.0, 5)
w = O.ransac_fundamental(x1, x2, 3000, 1.0, 5, nthreads=16)
assert g[0] == w[0] and g[4] == w[4] and (g[2] == w[2]).all() and (g[1].view(np.uint64) == w[1].view(np.uint64)).all()
print("ransac n=30000 ok", g[3])
if "huge" in sys.argv:
    # a train copy beyond 2 GiB (7.6 M rows x 288 B): the coarse kernel's LDS-DMA addresses it through a buffer descriptor
    # with 32-bit offsets, so such sizes fall back to register staging — same result
    nq, nt = 64, 7_600_000
    t = rng.integers(0, 120, size=(nt, 128), dtype=np.uint8).astype(np.float32)
    q = t[rng.integers(0, nt, size=nq)].copy()
    q[:, ::7] += 3.0
    t0 = time.time(); got = ctx.bf_knn_l2(q, t, 2, 4); t1 = time.time()
    want = O.bf_knn_l2(q, t, 2, nthreads=16)
    assert_matches_equal(got, want, "huge nt")
    print("L2", nq, nt, "huge ok", round(t1 - t0, 3), "s incl. copies")
    got = ctx.bf_knn_l2(q, t, 2, U8)        # 7.6 M rows x 128 B = 0.97 GB of byte copies: inside the 32-bit LDS-DMA offsets
    assert_matches_equal(got, want, "huge nt, u8 hint")
    print("L2", nq, nt, "huge, u8 hint ok")
