"""GPU parity: pm_lmeds_fundamental (7-point minimal solver + LMedS, SURVEY 8f-3: the estimator the
reference's findFundamentalMat(..., CV_FM_7POINT) call selects, main.cpp:95-98) vs the CPU oracle:
winning model id, median, inlier mask and F bit-exact (docs/SPEC.md S13-S15)."""
import numpy as np
import pytest

import points_matching_amd as pm
from points_matching_amd import synth
from points_matching_amd.api import PM_E_NO_MODEL, PM_E_TOO_FEW, PM_OK, lmeds_default_iters, lmeds_fundamental

pytestmark = pytest.mark.gpu


def _same(got, want, what):
    rc_g, F_g, mask_g, n_g, best_g, med_g = got
    rc_w, F_w, mask_w, n_w, best_w, med_w = want
    assert rc_g == rc_w, what
    assert best_g == best_w, (what, best_g, best_w, med_g, med_w)
    assert np.float64(med_g).view(np.uint64) == np.float64(med_w).view(np.uint64), (what, med_g, med_w)
    assert n_g == n_w and (mask_g == mask_w).all(), what
    assert (F_g.view(np.uint64) == F_w.view(np.uint64)).all(), (what, F_g, F_w)


@pytest.mark.parametrize("n,iters,out_frac,noise", [(500, 300, 0.0, 0.0), (501, 300, 0.3, 0.5), (1000, 300, 0.3, 0.5),
                                                    (8, 40, 0.0, 0.3), (9, 64, 0.2, 0.5), (2275, 300, 0.4, 1.0),
                                                    (64, 1000, 0.5, 2.0)])
def test_lmeds_parity(ctx, oracle, n, iters, out_frac, noise):
    x1, x2, Fgt, inl = synth.two_view(n, seed=11 * n + iters, outlier_frac=out_frac, noise_px=noise)
    got = lmeds_fundamental(ctx, x1, x2, iters, 0x7EED)
    want = oracle.lmeds_fundamental(x1, x2, iters, 0x7EED, nthreads=8)
    _same(got, want, str((n, iters)))
    if noise == 0.0 and out_frac == 0.0:
        assert got[3] == n
        assert min(np.abs(got[1] - Fgt).max(), np.abs(got[1] + Fgt).max()) < 1e-5


def test_lmeds_hyp_range_and_default_iters(ctx, oracle):
    assert lmeds_default_iters() == 300                      # OpenCV's count for (0.99, 0.45) [recalled]
    x1, x2, _, _ = synth.two_view(700, seed=4, outlier_frac=0.3, noise_px=0.5)
    full = lmeds_fundamental(ctx, x1, x2, 300, 5)
    _same(full, oracle.lmeds_fundamental(x1, x2, 300, 5, nthreads=8), "full")
    # a sub-range that contains the winner gives the same answer; model ids are global (3h + root)
    h = full[4] // 3
    part = lmeds_fundamental(ctx, x1, x2, h + 1, 5, hyp_begin=h)
    assert part[4] == full[4] and (part[1].view(np.uint64) == full[1].view(np.uint64)).all()
    _same(part, oracle.lmeds_fundamental(x1, x2, h + 1, 5, hyp_begin=h), "part")


def test_lmeds_too_few_degenerate_and_limits(ctx, oracle):
    x1, x2, _, _ = synth.two_view(7, seed=1)
    assert lmeds_fundamental(ctx, x1, x2, 10, 1)[0] == PM_E_TOO_FEW
    # all correspondences identical: every sample is degenerate
    x1 = np.tile(np.float32([[10, 20]]), (50, 1))
    x2 = np.tile(np.float32([[11, 21]]), (50, 1))
    got = lmeds_fundamental(ctx, x1, x2, 30, 1)
    want = oracle.lmeds_fundamental(x1, x2, 30, 1)
    assert got[0] == want[0] == PM_E_NO_MODEL and not got[2].any() and (got[1] == 0).all() and got[4] == -1
    # collinear image-1 points: whatever the oracle says, the GPU says the same
    rng = np.random.default_rng(0)
    t = rng.uniform(0, 900, 60).astype(np.float32)
    x1 = np.stack([t, 0.5 * t + 3], 1).astype(np.float32)
    x2 = rng.uniform(0, 600, (60, 2)).astype(np.float32)
    _same(lmeds_fundamental(ctx, x1, x2, 50, 2), oracle.lmeds_fundamental(x1, x2, 50, 2), "collinear")
    with pytest.raises(pm.PmError):
        lmeds_fundamental(ctx, np.zeros((40000, 2), np.float32), np.zeros((40000, 2), np.float32), 10, 1)


def test_lmeds_on_the_reference_image_pair_fixture(ctx, oracle):
    """Config C1 data (img01/img02 descriptors fixture): matcher -> ratio -> LMedS, GPU == oracle."""
    import os
    fx = np.load(os.path.join(os.path.dirname(__file__), "golden", "img01_img02_sift.npz"))
    knn = ctx.bf_knn_l2(fx["desc1"].astype(np.float32), fx["desc2"].astype(np.float32), 2)
    good = pm.api.filter_ratio(knn, 0.8)
    x1 = fx["kp1"][good["queryIdx"]]
    x2 = fx["kp2"][good["trainIdx"]]
    assert x1.shape[0] >= 8
    _same(lmeds_fundamental(ctx, x1, x2, 300, 0xC1), oracle.lmeds_fundamental(x1, x2, 300, 0xC1, nthreads=8), "C1")


# ---- adaptive-iteration RANSAC over 7-point models (SPEC S16; OpenCV CV_FM_RANSAC structure) --------
@pytest.mark.parametrize("n,out_frac,noise,max_iters,thresh", [(1000, 0.0, 0.5, 2000, 3.0), (1000, 0.3, 0.5, 2000, 3.0),
                                                               (700, 0.6, 0.5, 2000, 3.0), (300, 0.4, 1.0, 700, 1.0),
                                                               (9, 0.0, 0.2, 50, 3.0), (2275, 0.3, 0.5, 2000, 1.0)])
def test_adaptive_ransac7_parity(ctx, oracle, n, out_frac, noise, max_iters, thresh):
    from points_matching_amd.api import ransac7_adaptive
    x1, x2, _, inl = synth.two_view(n, seed=7 * n + max_iters, outlier_frac=out_frac, noise_px=noise)
    got = ransac7_adaptive(ctx, x1, x2, max_iters, 0.99, thresh, 0xADA)
    want = oracle.ransac7_adaptive(x1, x2, max_iters, 0.99, thresh, 0xADA)
    assert got[0] == want[0] and got[4] == want[4] and got[5] == want[5], (got[0], want[0], got[4:], want[4:])
    assert got[3] == want[3] and (got[2] == want[2]).all()
    assert (got[1].view(np.uint64) == want[1].view(np.uint64)).all()
    if out_frac <= 0.3 and n >= 300:
        assert got[5] < max_iters                    # the budget shrank: fewer hypotheses visited than the cap


def test_adaptive_ransac7_edges(ctx, oracle):
    from points_matching_amd.api import ransac7_adaptive
    x1, x2, _, _ = synth.two_view(7, seed=1)
    assert ransac7_adaptive(ctx, x1, x2, 100, 0.99, 3.0, 1)[0] == PM_E_TOO_FEW
    x1 = np.tile(np.float32([[10, 20]]), (50, 1))
    x2 = np.tile(np.float32([[11, 21]]), (50, 1))
    got = ransac7_adaptive(ctx, x1, x2, 600, 0.99, 3.0, 1)            # every sample degenerate: two full batches
    want = oracle.ransac7_adaptive(x1, x2, 600, 0.99, 3.0, 1)
    assert got[0] == want[0] == PM_E_NO_MODEL and got[5] == want[5] == 600 and not got[2].any()
    with pytest.raises(pm.PmError):
        ransac7_adaptive(ctx, np.zeros((20, 2), np.float32), np.zeros((20, 2), np.float32), 0, 0.99, 3.0, 1)
