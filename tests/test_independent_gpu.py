"""GPU twins of tests/test_independent_cpu.py: the HIP 8-point solver and the HIP inlier predicate against
DIFFERENT algorithms (numpy SVD 8-point, float64 textbook residuals), not against the oracle that shares their
specification.  Slot in the reference: cv::findFundamentalMat, main.cpp:95-98."""
import numpy as np
import pytest

from points_matching_amd import synth
from test_independent_cpu import _sampson64, np_eight_point

pytestmark = pytest.mark.gpu


def test_hip_solver_agrees_with_numpy_svd_eight_point(ctx, oracle):
    x1, x2, _, _ = synth.two_view(900, seed=21, outlier_frac=0.2, noise_px=0.7)
    worst, checked = 0.0, 0
    for h in range(1100):
        rc, F, mask, ninl = ctx.ransac_model_from_hyp(x1, x2, h, 1.0, 0xBEEF)
        assert rc == 0
        idx = oracle.sample8(0xBEEF, h, x1.shape[0])          # SPEC S6 (which 8 correspondences id h draws)
        F_np, sv, S = np_eight_point(x1[idx].astype(np.float64), x2[idx].astype(np.float64))
        gap = min(sv[7] / sv[0], (S[1] - S[2]) / S[0])
        if gap < 1e-5:
            continue
        err = min(np.linalg.norm(F - F_np), np.linalg.norm(F + F_np))
        assert err <= 5e-13 / gap + 1e-13, (h, err, gap)
        worst = max(worst, err)
        checked += 1
        # the device's mask for this model against the float64 Sampson distance, away from the threshold
        samp, _ = _sampson64(F.astype(np.float32).astype(np.float64), x1, x2)
        far = np.abs(samp - 1.0) > 1e-3
        assert ((samp <= 1.0)[far] == mask.astype(bool)[far]).all(), h
    assert checked >= 1000 and worst < 1e-9


def test_hip_winner_is_the_best_model_by_an_independent_count(ctx):
    """The run's winner has at least as many float64-counted inliers as 300 other hypotheses of the same run
    (up to the correspondences within 0.1 % of the threshold)."""
    x1, x2, _, _ = synth.two_view(1500, seed=8, outlier_frac=0.35, noise_px=0.6)
    rc, F, mask, ninl, key = ctx.ransac_fundamental(x1, x2, 300, 1.0, 42)
    assert rc == 0
    samp, _ = _sampson64(F.astype(np.float32).astype(np.float64), x1, x2)
    border = int((np.abs(samp - 1.0) <= 1e-3).sum())
    assert abs(int((samp <= 1.0).sum()) - ninl) <= border
    for h in range(300):
        rc_h, F_h, mask_h, n_h = ctx.ransac_model_from_hyp(x1, x2, h, 1.0, 42)
        s_h, _ = _sampson64(F_h.astype(np.float32).astype(np.float64), x1, x2)
        assert int((s_h <= 1.0).sum()) <= ninl + border + int((np.abs(s_h - 1.0) <= 1e-3).sum()), h


@pytest.mark.parametrize("kind,flags", [("sift", 0), ("sift", 4), ("surf", 0), ("surf", 2)])
def test_hip_knn_agrees_with_scikit_learn(ctx, kind, flags):
    """The HIP matcher (integer route, hinted route, general-float route, f32-input route) against scikit-learn's
    brute-force neighbours — not against the oracle.  Slot: matcher.match, main.cpp:46."""
    from test_independent_cpu import check_knn_against_sklearn
    q, t, _ = (synth.sift_like if kind == "sift" else synth.surf_like)(600, 3000, 128, seed=29)
    check_knn_against_sklearn(ctx.bf_knn_l2(q, t, 2, flags), q, t, 2, "%s flags %d" % (kind, flags))


def test_hip_hamming_agrees_with_scikit_learn(ctx):
    from test_independent_cpu import sklearn_knn
    q, t, _ = synth.orb_like(500, 2500, 32, seed=31)
    got = ctx.bf_knn_hamming(q, t, 2)
    qb, tb = np.unpackbits(q, axis=1).astype(np.float64), np.unpackbits(t, axis=1).astype(np.float64)
    d_sk, i_sk = sklearn_knn(qb, tb, 3, metric="hamming")
    d_sk = np.rint(d_sk * 256.0)
    assert (got["distance"] == d_sk[:, :2]).all()
    for j in range(2):
        clear = (d_sk[:, j + 1] > d_sk[:, j]) & ((j == 0) | (d_sk[:, j] > d_sk[:, j - 1]))
        assert (got["trainIdx"][clear, j] == i_sk[clear, j]).all()


def test_hip_lmeds_winner_is_a_numpy_seven_point_solution(ctx, oracle):
    """7-point + LMedS on the device (SPEC S13-S15): the F it returns is one of numpy's (SVD null space + np.roots)
    solutions for the seven correspondences its winning id samples.  Slot: findFundamentalMat(CV_FM_7POINT)."""
    from points_matching_amd.api import lmeds_fundamental
    from test_independent_cpu import np_seven_point
    checked = 0
    for s in range(12):
        x1, x2, _, _ = synth.two_view(600 + 50 * s, seed=100 + s, outlier_frac=0.3, noise_px=0.5)
        rc, F, mask, ninl, best, med = lmeds_fundamental(ctx, x1, x2, 300, 9 + s)
        assert rc == 0 and best >= 0
        idx = oracle.sample7(9 + s, best // 3, x1.shape[0])
        sols, sv = np_seven_point(x1[idx].astype(np.float64), x2[idx].astype(np.float64))
        if sv[6] / sv[0] < 1e-6:
            continue
        err = min(min(np.linalg.norm(F - G), np.linalg.norm(F + G)) for G in sols)
        sep = min([np.linalg.norm(G - H) for gi, G in enumerate(sols) for H in sols[gi + 1:]] + [1.0])
        assert err <= 1e-9 / max(sep, 1e-6) + 1e-11, (s, err, sep)
        checked += 1
    assert checked >= 10
