"""Randomised parity campaign on the GPU box: many more (and wider) cases than the test suite carries, every result
compared bit for bit with the CPU oracle.  Not part of the tests (it runs for minutes); a mismatch prints the case and
exits non-zero.        python tools/fuzz_campaign.py [seconds=300] [seed=1]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import points_matching_amd as pm  # noqa: E402
from points_matching_amd import synth  # noqa: E402
from oracle import pm_oracle as O  # noqa: E402
from util import assert_matches_equal  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
O.build()
ctx = pm.Context(0)
A = pm.api
rng = np.random.default_rng(seed)
t_end = time.time() + budget
counts = {"l2": 0, "hamming": 0, "ransac": 0, "lmeds": 0}
NT = 16
TRACE = bool(os.environ.get("PM_FUZZ_TRACE"))      # print every GPU call before it is made (a GPU fault names no case)


def trace(*a):
    if TRACE:
        print("  >", *a, flush=True)


def l2_case(i):
    dim = int(rng.choice([4, 5, 7, 8, 12, 16, 20, 32, 36, 64, 96, 100, 128, 128, 128, 128, 128, 130, 200, 256]))
    big = rng.random() < 0.15 and dim <= 128
    nq = int(rng.integers(1, 12000 if big else 2500))
    nt = int(rng.integers(1, 40000 if big else 6000))
    k = int(rng.choice([1, 2, 2, 2, 3, 4, 5]))
    if rng.random() < 0.08:                                  # few dimensions, many train rows: hundreds of near ties per query,
        dim = int(rng.choice([4, 8, 12, 16, 20, 24]))        # where a coarse pass that ranks on stale operands loses neighbours
        nq = int(rng.integers(64, 1200)); nt = int(rng.integers(15000, 40000))
    kind = str(rng.choice(["sift", "surf"]))
    q, t, _ = (synth.sift_like if kind == "sift" else synth.surf_like)(nq, nt, dim, seed=int(rng.integers(1 << 30)))
    twist = int(rng.integers(0, 9))
    if twist == 1 and nt > 8:
        for _ in range(int(rng.integers(1, 6))):
            a = int(rng.integers(0, nt)); ln = int(rng.integers(2, min(20, nt)))
            b = int(rng.integers(0, nt - ln + 1))
            t[b:b + ln] = t[a]
            q[int(rng.integers(0, nq))] = t[a]
    elif twist == 2:
        t[int(rng.integers(0, nt)), int(rng.integers(0, dim))] += 0.5
    elif twist == 3:
        q *= np.float32(10.0 ** rng.uniform(-6, 6)); t *= np.float32(10.0 ** rng.uniform(-6, 6))
    elif twist == 4:
        t[::3] *= np.float32(10.0 ** rng.uniform(-4, 0)); q[::2] *= np.float32(10.0 ** rng.uniform(-3, 3))
    elif twist == 5:
        q[int(rng.integers(0, nq))] = 0.0; t[int(rng.integers(0, nt))] = 0.0
    elif twist == 6:                                     # sub-f16-ulp clusters around a few queries
        for _ in range(3):
            a = int(rng.integers(0, nq))
            rows = rng.choice(nt, size=min(nt, int(rng.integers(2, 30))), replace=False)
            t[rows] = q[a] * (1.0 + rng.uniform(-2e-5, 2e-5, size=(rows.size, dim)).astype(np.float32))
    elif twist == 7:
        q = np.abs(q) * np.float32(255.0 / max(1e-9, np.abs(q).max())); q = np.rint(q)   # integers in float queries only
    want = O.bf_knn_l2(q, t, k, nthreads=NT)
    opts = {}
    if rng.random() < 0.4:
        opts = {A.PM_OPT_KNN_F16_WAVES: int(rng.integers(0, 4)), A.PM_OPT_KNN_STAGING: int(rng.integers(0, 3)),
                A.PM_OPT_KNN_XCD_TILE: int(rng.integers(0, 3)), A.PM_OPT_KNN_WG_PER_CU: int(rng.integers(0, 3)),
                A.PM_OPT_KNN_GENERAL_F16: int(rng.integers(0, 3)),
                # round 3: seeded f16 form, u8 group size / coarse-kernel form / refinement, wide-dim passes, prep geometry
                A.PM_OPT_KNN_SEEDED: int(rng.integers(0, 3)), A.PM_OPT_KNN_U8_GROUP: int(rng.integers(0, 4)),
                A.PM_OPT_KNN_RING: int(rng.integers(0, 7)), A.PM_OPT_KNN_U8_REFINE: int(rng.integers(0, 3)),
                A.PM_OPT_KNN_RING_PROLOGUE: int(rng.integers(0, 9)), A.PM_OPT_KNN_WIDE: int(rng.integers(0, 3)),
                A.PM_OPT_KNN_PREP_ROWS: int(rng.integers(0, 3))}
    u8_valued = q.size and t.size and q.min() >= 0 and q.max() <= 255 and t.min() >= 0 and t.max() <= 255 and \
        bool((q == np.rint(q)).all()) and bool((t == np.rint(t)).all())
    try:
        for o, v in opts.items():
            ctx.set_option(o, v)
        for flags in (0, A.PM_KNN_HINT_INTEGER, A.PM_KNN_FORCE_F32, A.PM_KNN_HINT_U8, A.PM_KNN_HINT_UNIT_NORM):     # (hints right or wrong)
            trace("l2", i, kind, nq, nt, dim, "k", k, "twist", twist, "flags", flags, opts)
            assert_matches_equal(ctx.bf_knn_l2(q, t, k, flags), want,
                                 "L2 case %d: %s %dx%dx%d k=%d twist=%d flags=%d opts=%s" % (i, kind, nq, nt, dim, k, twist, flags, opts))
        if u8_valued:                                     # the same values as uint8 rows (pm_bf_knn_l2_u8)
            trace("l2 u8 rows", i, nq, nt, dim, "k", k, opts)
            assert_matches_equal(ctx.bf_knn_l2_u8(q.astype(np.uint8), t.astype(np.uint8), k), want,
                                 "L2 case %d (u8 rows): %dx%dx%d k=%d twist=%d opts=%s" % (i, nq, nt, dim, k, twist, opts))
            counts["l2_u8_rows"] = counts.get("l2_u8_rows", 0) + 1
    finally:
        for o in opts:
            ctx.set_option(o, 0)


def hamming_case(i):
    nbytes = int(rng.choice([4, 8, 16, 32, 32, 32, 32, 64]))
    big = rng.random() < 0.15
    nq = int(rng.integers(1, 9000 if big else 1500))
    nt = int(rng.integers(1, 30000 if big else 4000))
    k = int(rng.choice([1, 2, 2, 3]))
    if rng.random() < 0.3:
        base = rng.integers(0, 256, (int(rng.integers(1, 9)), nbytes), dtype=np.uint8)
        q = base[rng.integers(0, base.shape[0], nq)].copy(); t = base[rng.integers(0, base.shape[0], nt)].copy()
        t[::5, 0] ^= 1
    else:
        q, t, _ = synth.orb_like(nq, nt, nbytes, seed=int(rng.integers(1 << 30)), flip=float(rng.uniform(0.0, 0.3)))
    route = int(rng.integers(0, 3))
    hform = int(rng.integers(0, 3))                          # refinement: automatic / one wave per query / four queries per wave
    try:
        ctx.set_option(A.PM_OPT_HAMMING_ROUTE, route)
        ctx.set_option(A.PM_OPT_HAMMING_REFINE, hform)
        trace("hamming", i, nq, nt, nbytes, "k", k, "route", route, "refine", hform)
        assert_matches_equal(ctx.bf_knn_hamming(q, t, k), O.bf_knn_hamming(q, t, k, nthreads=NT),
                             "Hamming case %d: %dx%d bytes=%d k=%d route=%d" % (i, nq, nt, nbytes, k, route))
    finally:
        ctx.set_option(A.PM_OPT_HAMMING_ROUTE, 0)
        ctx.set_option(A.PM_OPT_HAMMING_REFINE, 0)


def same_ransac(got, want, what):
    assert got[0] == want[0] and got[4] == want[4] and got[3] == want[3], (what, got[0], want[0], got[4], want[4], got[3], want[3])
    assert (got[2] == want[2]).all(), what
    assert (got[1].view(np.uint64) == want[1].view(np.uint64)).all(), what


def ransac_case(i):
    n = int(rng.choice([int(rng.integers(0, 12)), int(rng.integers(8, 3000)), int(rng.integers(3000, 12000))], p=[0.1, 0.7, 0.2]))
    iters = int(rng.integers(1, 3000))
    hb = int(rng.integers(0, 1 << 24))
    thr = float(rng.choice([0.25, 1.0, 3.0, 10.0]))
    kind = int(rng.integers(0, 2))
    x1, x2, _, _ = synth.two_view(max(n, 1), seed=int(rng.integers(1 << 30)), outlier_frac=float(rng.uniform(0, 0.7)),
                                  noise_px=float(rng.uniform(0, 2.0)))
    x1, x2 = x1[:n], x2[:n]
    tw = int(rng.integers(0, 5))
    if tw == 1 and n >= 8:
        x1[: n // 2] = x1[0]; x2[: n // 2] = x2[0]          # many identical correspondences (degenerate samples)
    elif tw == 2 and n >= 8:
        x2[:] = x1                                          # pure identity motion
    elif tw == 3 and n >= 8:
        x1[:, 1] = 3.0                                      # collinear points in image 1
    path = int(rng.integers(0, 3))
    form = int(rng.integers(0, 3))                           # one-launch kernel: automatic / register tiles / LDS tile
    ids = int(rng.choice([0, 0, int(rng.integers(1, 129))]))  # hypothesis ids per workgroup
    try:
        ctx.set_option(A.PM_OPT_RANSAC_PATH, path)
        ctx.set_option(A.PM_OPT_RANSAC_FORM, form)
        ctx.set_option(A.PM_OPT_RANSAC_WG_IDS, ids)
        trace("ransac", i, "n", n, "iters", iters, "hb", hb, "thr", thr, "kind", kind, "tw", tw, "path", path, "form", form, "ids", ids)
        got = ctx.ransac_fundamental(x1, x2, hb + iters, thr, int(rng.integers(1 << 31)) if False else 77 + i, kind, hyp_begin=hb)
        want = O.ransac_fundamental(x1, x2, hb + iters, thr, 77 + i, kind, hyp_begin=hb, nthreads=NT)
        same_ransac(got, want, "RANSAC case %d: n=%d iters=%d hb=%d thr=%g kind=%d tw=%d path=%d form=%d ids=%d" % (i, n, iters, hb, thr, kind, tw, path, form, ids))
    finally:
        ctx.set_option(A.PM_OPT_RANSAC_PATH, 0)
        ctx.set_option(A.PM_OPT_RANSAC_FORM, 0)
        ctx.set_option(A.PM_OPT_RANSAC_WG_IDS, 0)


def lmeds_case(i):
    n = int(rng.integers(7, 6000))
    iters = int(rng.integers(1, 600))
    x1, x2, _, _ = synth.two_view(n, seed=int(rng.integers(1 << 30)), outlier_frac=float(rng.uniform(0, 0.45)),
                                  noise_px=float(rng.uniform(0, 1.5)))
    trace("lmeds", i, n, iters)
    got = A.lmeds_fundamental(ctx, x1, x2, iters, 5 + i)
    want = O.lmeds_fundamental(x1, x2, iters, 5 + i, nthreads=NT)
    what = "LMedS case %d: n=%d iters=%d" % (i, n, iters)
    assert got[0] == want[0] and got[3] == want[3] and got[4] == want[4], (what, got[0], want[0], got[3:5], want[3:5])
    assert (got[2] == want[2]).all(), what
    assert (got[1].view(np.uint64) == np.asarray(want[1]).view(np.uint64)).all(), what
    assert np.float64(got[5]).view(np.uint64) == np.float64(want[5]).view(np.uint64), what


def pipeline_case(i):
    """pm_bf_knn_l2_ratio_dev (matcher + ratio + compaction + gather, fused or as two launches) feeding the one-call
    RANSAC run on device buffers: survivors, points, winner and mask against the oracle's chain."""
    import torch
    dev = torch.device("cuda", 0)
    dim = int(rng.choice([32, 64, 128, 128]))
    nq = int(rng.integers(8, 3000)); nt = int(rng.integers(2, 4000))
    kind = str(rng.choice(["sift", "surf"]))
    w = synth.pair_workload(nq, nt, dim, seed=int(rng.integers(1 << 30)), planted=float(rng.uniform(0.05, 0.6)), kind=kind)
    flags = int(rng.choice([0, A.PM_KNN_HINT_INTEGER, A.PM_KNN_HINT_U8, A.PM_KNN_HINT_U8])) if kind == "sift" else 0
    ratio = float(rng.choice([0.6, 0.8, 0.95]))
    mode = int(rng.integers(0, 3)); with_knn = bool(rng.integers(0, 2)) or mode == 1
    H = int(rng.integers(1, 1500))
    d_q, d_t = torch.from_numpy(w["q"]).to(dev), torch.from_numpy(w["t"]).to(dev)
    d_kp1, d_kp2 = torch.from_numpy(w["kp1"]).to(dev), torch.from_numpy(w["kp2"]).to(dev)
    d_knn = torch.zeros((nq, 2, 4), dtype=torch.int32, device=dev)
    d_good = torch.full((nq, 4), -7, dtype=torch.int32, device=dev)
    d_xy1 = torch.full((nq, 2), -1.0, dtype=torch.float32, device=dev)
    d_xy2 = torch.full((nq, 2), -1.0, dtype=torch.float32, device=dev)
    d_n = torch.full((1,), -1, dtype=torch.int32, device=dev)
    d_key = torch.zeros(1, dtype=torch.int64, device=dev); d_F = torch.zeros(9, dtype=torch.float64, device=dev)
    d_mask = torch.zeros(nq, dtype=torch.uint8, device=dev); d_ninl = torch.zeros(1, dtype=torch.int32, device=dev)
    what = "pipeline case %d: %s %dx%dx%d flags=%d ratio=%g fusion=%d knn=%d H=%d" % (i, kind, nq, nt, dim, flags, ratio, mode, with_knn, H)
    trace(what)
    try:
        ctx.set_option(A.PM_OPT_FILTER_FUSION, mode)
        ctx.bf_knn_l2_ratio_dev(d_q.data_ptr(), nq, d_t.data_ptr(), nt, dim, flags, ratio, d_kp1.data_ptr(), d_kp2.data_ptr(),
                                d_knn.data_ptr() if with_knn else 0, d_good.data_ptr(), d_xy1.data_ptr(), d_xy2.data_ptr(),
                                d_n.data_ptr())
        ctx.ransac_run_dev(d_xy1.data_ptr(), d_xy2.data_ptr(), nq, d_n.data_ptr(), 0, H, 1.0, 99 + i, d_key.data_ptr(),
                           d_F.data_ptr(), d_mask.data_ptr(), d_ninl.data_ptr())
        ctx.synchronize()
    finally:
        ctx.set_option(A.PM_OPT_FILTER_FUSION, 0)
    want = O.filter_ratio(O.bf_knn_l2(w["q"], w["t"], 2, nthreads=NT), ratio)
    n = int(d_n.item())
    assert n == want.size, (what, n, want.size)
    assert_matches_equal(d_good.cpu().numpy().view(pm.MATCH_DTYPE).reshape(-1)[:n], want, what)
    x1, x2 = w["kp1"][want["queryIdx"]], w["kp2"][want["trainIdx"]]
    assert (d_xy1.cpu().numpy()[:n] == x1).all() and (d_xy2.cpu().numpy()[:n] == x2).all(), what
    rw = O.ransac_fundamental(x1, x2, H, 1.0, 99 + i, nthreads=NT)
    assert int(d_key.item()) == (rw[4] if rw[4] < (1 << 63) else rw[4] - (1 << 64)), (what, int(d_key.item()), rw[4])
    if n >= 8:
        assert int(d_ninl.item()) == rw[3] and (d_mask.cpu().numpy()[:n] == rw[2]).all(), what
        assert (d_F.cpu().numpy().view(np.uint64) == rw[1].reshape(9).view(np.uint64)).all(), what


counts["pipeline"] = 0
cases = [("l2", l2_case, 0.4), ("hamming", hamming_case, 0.18), ("ransac", ransac_case, 0.22), ("lmeds", lmeds_case, 0.08),
         ("pipeline", pipeline_case, 0.12)]
i = 0
last = time.time()
while time.time() < t_end:
    name, fn, _ = cases[int(rng.choice(len(cases), p=[c[2] for c in cases]))]
    fn(i)
    counts[name] += 1
    i += 1
    if time.time() - last > 30:
        print("...", counts, flush=True)
        last = time.time()
print("fuzz campaign ok", counts, "seed", seed)
