"""Independent correctness evidence for the parts of the path whose OpenCV arithmetic cannot be pinned (SURVEY.md 8c:
"parity unpinned"): the oracle and the kernels share one specification, so bit-equality between them proves a
consistent transcription, not correctness.  Here the oracle is checked against DIFFERENT algorithms:
  * the 8-point solver (docs/SPEC.md S7: Householder QR null vector + one-sided Jacobi rank-2) against numpy's
    SVD-based normalised 8-point (Hartley) on 1500 random minimal samples;
  * the division-free fp32 Sampson / symmetric-epipolar predicates (S8) against a float64 evaluation of the textbook
    formulas, away from the threshold;
  * the canonical fp32 squared distance (S1) against math.fsum within its a-priori error bound;
  * the sampler (S6) for uniformity;
  * the 7-point solver (S14: QR null space, bisection + deflation of the cubic) against numpy's SVD null space +
    np.roots on 600 random minimal samples;
  * the matcher's definition of "nearest" (S1-S3) against scikit-learn's brute-force neighbours (another code base,
    float64 GEMM-expansion distances; Hamming on unpacked bits), indices wherever neighbours are not tied.
GPU twins live in tests/test_independent_gpu.py."""
import math

import numpy as np
import pytest


def np_eight_point(p1, p2):
    """Textbook normalised 8-point with numpy's LAPACK SVD.  p1, p2: 8 x 2 float64.  x2^T F x1 = 0."""
    def hartley(p):
        c = p.mean(axis=0)
        md = np.sqrt(((p - c) ** 2).sum(axis=1)).mean()
        s = math.sqrt(2.0) / md
        T = np.array([[s, 0, -s * c[0]], [0, s, -s * c[1]], [0, 0, 1.0]])
        return (p - c) * s, T
    a, T1 = hartley(p1)
    b, T2 = hartley(p2)
    A = np.stack([b[:, 0] * a[:, 0], b[:, 0] * a[:, 1], b[:, 0], b[:, 1] * a[:, 0], b[:, 1] * a[:, 1], b[:, 1],
                  a[:, 0], a[:, 1], np.ones(8)], axis=1)
    _, sv, Vt = np.linalg.svd(A)                     # 8 x 9: the null vector is the last row of Vt
    Fn = Vt[-1].reshape(3, 3)
    U, S, Wt = np.linalg.svd(Fn)
    Fn = U @ np.diag([S[0], S[1], 0.0]) @ Wt
    F = T2.T @ Fn @ T1
    F /= np.linalg.norm(F)
    if F[2, 2] < 0:
        F = -F
    return F, sv, S


def test_solver_agrees_with_numpy_svd_eight_point(oracle):
    rng = np.random.default_rng(2024)
    worst, checked = 0.0, 0
    for trial in range(1500):
        # a real two-view geometry (random camera pair, points in front of both) plus pixel noise
        X = np.column_stack([rng.uniform(-3, 3, 8), rng.uniform(-2, 2, 8), rng.uniform(4, 12, 8)])
        ang = rng.normal(0, 0.08, 3)
        Rx = np.array([[1, 0, 0], [0, math.cos(ang[0]), -math.sin(ang[0])], [0, math.sin(ang[0]), math.cos(ang[0])]])
        Ry = np.array([[math.cos(ang[1]), 0, math.sin(ang[1])], [0, 1, 0], [-math.sin(ang[1]), 0, math.cos(ang[1])]])
        Rz = np.array([[math.cos(ang[2]), -math.sin(ang[2]), 0], [math.sin(ang[2]), math.cos(ang[2]), 0], [0, 0, 1]])
        R, t = Rz @ Ry @ Rx, np.array([1.0, rng.normal(0, 0.1), rng.normal(0, 0.1)])
        K = np.array([[1000.0, 0, 496.5], [0, 1000.0, 330.0], [0, 0, 1]])
        x1 = (K @ X.T).T
        x2 = (K @ (R @ X.T + t[:, None])).T
        p1 = (x1[:, :2] / x1[:, 2:]).astype(np.float32).astype(np.float64) + rng.normal(0, 0.5, (8, 2))
        p2 = (x2[:, :2] / x2[:, 2:]).astype(np.float32).astype(np.float64) + rng.normal(0, 0.5, (8, 2))
        ok, F = oracle.solve8(p1, p2)
        F_np, sv, S = np_eight_point(p1, p2)
        assert ok
        # conditioning of this sample: the null vector is determined up to ~eps / (relative gap of A's 8th singular
        # value), the rank-2 projection up to ~eps / (relative gap between F's 2nd and 3rd singular values)
        gap = min(sv[7] / sv[0], (S[1] - S[2]) / S[0])
        if gap < 1e-5:
            continue                                  # nearly degenerate sample: the two algorithms may legitimately differ
        err = min(np.linalg.norm(F - F_np), np.linalg.norm(F + F_np))
        assert err <= 5e-13 / gap + 1e-13, (trial, err, gap)
        worst = max(worst, err)
        checked += 1
        assert abs(np.linalg.norm(F) - 1) < 1e-14 and F[2, 2] >= 0
        assert abs(np.linalg.det(F)) < 1e-12          # rank 2
    assert checked >= 1300
    assert worst < 1e-9


def _sampson64(F, x1, x2):
    x1h = np.column_stack([x1, np.ones(len(x1))]).astype(np.float64)
    x2h = np.column_stack([x2, np.ones(len(x2))]).astype(np.float64)
    Fx = x1h @ F.T                                   # rows: F x1
    Ftx = x2h @ F                                    # rows: F^T x2
    num = (x2h * Fx).sum(axis=1)
    samp = num ** 2 / (Fx[:, 0] ** 2 + Fx[:, 1] ** 2 + Ftx[:, 0] ** 2 + Ftx[:, 1] ** 2)
    d2 = num ** 2 / (Fx[:, 0] ** 2 + Fx[:, 1] ** 2)   # squared distance of x2 to the line F x1
    d1 = num ** 2 / (Ftx[:, 0] ** 2 + Ftx[:, 1] ** 2)
    return samp, np.maximum(d1, d2)


@pytest.mark.parametrize("kind", [0, 1])
def test_inlier_predicates_agree_with_float64_formulas(oracle, kind):
    from points_matching_amd import synth
    total = 0
    for seed in range(6):
        x1, x2, Fgt, _ = synth.two_view(3000, seed=seed, outlier_frac=0.5, noise_px=1.0)
        ok, F, F32 = oracle.hyp_model(x1, x2, 77, seed)     # some hypothesis of this pair
        if not ok:
            continue
        for thr in (0.5, 1.0, 3.0):
            cnt, mask = oracle.score(F32, x1, x2, thr, kind)
            samp, sym = _sampson64(F32.astype(np.float64), x1, x2)
            e = samp if kind == 0 else sym
            far = np.abs(e / thr ** 2 - 1.0) > 1e-3          # fp32 evaluation vs fp64: decisive only away from tau^2
            assert far.sum() > 0.98 * far.size
            assert ((e <= thr ** 2)[far] == mask.astype(bool)[far]).all()
            assert cnt == int(mask.sum())
            total += int(far.sum())
    assert total > 20000


def test_canonical_distance_within_its_bound_of_the_exact_sum(oracle):
    rng = np.random.default_rng(5)
    for dim in (1, 7, 8, 31, 64, 128, 200):
        for _ in range(200):
            a = rng.normal(0, 1, dim).astype(np.float32)
            b = rng.normal(0, 1, dim).astype(np.float32)
            exact = math.fsum((float(x) - float(y)) ** 2 for x, y in zip(a, b))
            got = float(oracle.l2sqr(a, b))
            # each term: one rounding of the difference (2 eps relative in the square) + one of the product; the sum:
            # at most ceil(dim/8) + 4 additions on any path.  gamma = that many half-ulps of 2^-24, bounded generously.
            gamma = (math.ceil(dim / 8) + 8) * 2.0 ** -24 * 1.01
            assert abs(got - exact) <= gamma * exact + 1e-45, (dim, got, exact)


def test_sampler_is_uniform_and_distinct(oracle):
    n = 97
    counts = np.zeros(n, np.int64)
    for h in range(20000):
        idx = oracle.sample8(0xABCDEF, h, n)
        assert len(set(idx.tolist())) == 8 and idx.min() >= 0 and idx.max() < n
        counts[idx] += 1
    exp = 20000 * 8 / n
    chi2 = ((counts - exp) ** 2 / exp).sum()
    assert chi2 < 170                                # 96 degrees of freedom: P(chi2 > 170) ~ 5e-6


def sklearn_knn(q, t, k, metric="euclidean"):
    """Brute-force k-NN by scikit-learn: a different code base and a different arithmetic (float64 GEMM expansion
    ||x||^2 + ||y||^2 - 2 x.y for 'euclidean', per-element mismatch fraction for 'hamming')."""
    from sklearn.neighbors import NearestNeighbors
    nn = NearestNeighbors(n_neighbors=k, algorithm="brute", metric=metric).fit(t)
    return nn.kneighbors(q)


def check_knn_against_sklearn(got, q, t, k, what):
    """Indices must agree wherever sklearn's own float64 distances separate neighbour j from j+1 by more than the fp32
    error of either side; distances within the a-priori bound of SPEC S1 (sqrt of a sum with (dim + 2) roundings)."""
    d_sk, i_sk = sklearn_knn(q.astype(np.float64), t.astype(np.float64), min(k + 1, t.shape[0]))
    dim = q.shape[1]
    scale = np.sqrt((q.astype(np.float64) ** 2).sum(1))[:, None] + np.sqrt((t.astype(np.float64) ** 2).sum(1)).max()
    tol = 64.0 * (dim + 8) * 2.0 ** -24 * scale            # generous: covers sklearn's cancellation as well
    kk = min(k, t.shape[0])
    assert np.all(np.abs(got["distance"][:, :kk].astype(np.float64) - d_sk[:, :kk]) <= tol), what
    for j in range(kk):
        # neighbour j is unambiguous when it is clear of both its neighbours in the sorted list
        lo = d_sk[:, j] - (d_sk[:, j - 1] if j > 0 else -np.inf) > 4 * tol[:, 0]
        hi = (d_sk[:, j + 1] if j + 1 < d_sk.shape[1] else np.inf) - d_sk[:, j] > 4 * tol[:, 0]
        clear = lo & hi
        assert clear.mean() > 0.5, (what, j, clear.mean())
        assert (got["trainIdx"][clear, j] == i_sk[clear, j]).all(), (what, j)


@pytest.mark.parametrize("kind", ["sift", "surf"])
def test_oracle_knn_agrees_with_scikit_learn(oracle, kind):
    """SPEC S1/S3 (what the oracle defines as THE result) against scikit-learn's brute-force neighbours."""
    from points_matching_amd import synth
    q, t, _ = (synth.sift_like if kind == "sift" else synth.surf_like)(400, 1500, 128, seed=17)
    check_knn_against_sklearn(oracle.bf_knn_l2(q, t, 2), q, t, 2, kind)


def test_oracle_hamming_agrees_with_scikit_learn(oracle):
    """SPEC S2/S3: Hamming distances are integers, so scikit-learn ('hamming' on the unpacked bits = mismatching
    fraction) must reproduce them exactly, and the indices wherever the k-th and (k+1)-th distances differ."""
    from points_matching_amd import synth
    q, t, _ = synth.orb_like(300, 1200, 32, seed=23)
    got = oracle.bf_knn_hamming(q, t, 2)
    qb, tb = np.unpackbits(q, axis=1).astype(np.float64), np.unpackbits(t, axis=1).astype(np.float64)
    d_sk, i_sk = sklearn_knn(qb, tb, 3, metric="hamming")
    d_sk = np.rint(d_sk * 256.0)
    assert (got["distance"] == d_sk[:, :2]).all()
    for j in range(2):
        clear = (d_sk[:, j + 1] > d_sk[:, j]) & ((j == 0) | (d_sk[:, j] > d_sk[:, j - 1]))
        assert clear.mean() > 0.5
        assert (got["trainIdx"][clear, j] == i_sk[clear, j]).all()


def np_seven_point(p1, p2):
    """Textbook 7-point with numpy: Hartley normalisation, null space of the 7 x 9 system by LAPACK SVD, the cubic
    det(a F1 + (1 - a) F2) = 0 by polynomial interpolation + np.roots.  Returns the real solutions, each with unit
    Frobenius norm and F[2,2] >= 0."""
    def hartley(p):
        c = p.mean(axis=0)
        md = np.sqrt(((p - c) ** 2).sum(axis=1)).mean()
        s = math.sqrt(2.0) / md
        T = np.array([[s, 0, -s * c[0]], [0, s, -s * c[1]], [0, 0, 1.0]])
        return (p - c) * s, T
    a, T1 = hartley(p1)
    b, T2 = hartley(p2)
    A = np.stack([b[:, 0] * a[:, 0], b[:, 0] * a[:, 1], b[:, 0], b[:, 1] * a[:, 0], b[:, 1] * a[:, 1], b[:, 1],
                  a[:, 0], a[:, 1], np.ones(7)], axis=1)
    _, sv, Vt = np.linalg.svd(A)
    F1, F2 = Vt[-1].reshape(3, 3), Vt[-2].reshape(3, 3)
    xs = np.array([-1.0, 0.0, 1.0, 2.0])
    ys = np.array([np.linalg.det(x * F1 + (1 - x) * F2) for x in xs])
    coef = np.polyfit(xs, ys, 3)
    out = []
    for r in np.roots(coef):
        if abs(r.imag) > 1e-9 * max(1.0, abs(r.real)):
            continue
        Fn = r.real * F1 + (1 - r.real) * F2
        F = T2.T @ Fn @ T1
        F /= np.linalg.norm(F)
        if F[2, 2] < 0:
            F = -F
        out.append(F)
    return out, sv


def test_seven_point_solver_agrees_with_numpy(oracle):
    """SPEC S14 (QR null space, bisection + deflation of the cubic) against numpy's SVD null space + np.roots: every
    model the oracle returns is one of numpy's solutions, and well-separated numpy solutions are all found."""
    from points_matching_amd import synth
    x1, x2, _, _ = synth.two_view(700, seed=41, outlier_frac=0.2, noise_px=0.6)
    rng = np.random.default_rng(43)
    checked, worst = 0, 0.0
    for _ in range(600):
        idx = rng.choice(700, size=7, replace=False)
        p1, p2 = x1[idx].astype(np.float64), x2[idx].astype(np.float64)
        Fo, valid = oracle.solve7(p1, p2)
        sols, sv = np_seven_point(p1, p2)
        if sv[6] / sv[0] < 1e-6 or not sols:           # (near-)degenerate sample: the null space itself is ill-defined
            continue
        mine = [Fo[r] for r in range(3) if valid[r]]
        assert mine, "no model for a regular sample"
        for F in mine:                                 # each returned model is a solution
            err = min(min(np.linalg.norm(F - G), np.linalg.norm(F + G)) for G in sols)
            # conditioning of a root: closeness of two solutions degrades both; scale the bound by it
            sep = min([np.linalg.norm(G - H) for gi, G in enumerate(sols) for H in sols[gi + 1:]] + [1.0])
            assert err <= 1e-9 / max(sep, 1e-6) + 1e-11, (err, sep)
            worst = max(worst, err)
            # and it satisfies the 7 epipolar constraints and det F = 0
            h1 = np.c_[p1, np.ones(7)]; h2 = np.c_[p2, np.ones(7)]
            # (the repo's convention, docs/SPEC.md S7: x2^T F x1 = 0)
            res = np.abs(np.einsum("ij,jk,ik->i", h2, F, h1))
            scale = np.linalg.norm(h1, axis=1) * np.linalg.norm(h2, axis=1)
            assert (res <= 1e-9 * scale).all() and abs(np.linalg.det(F)) <= 1e-12
        if len(sols) == len(mine) or all(np.linalg.norm(G - H) > 1e-3 for gi, G in enumerate(sols) for H in sols[gi + 1:]):
            assert len(mine) == len(sols), (len(mine), len(sols))
        checked += 1
    assert checked >= 500 and worst < 1e-8
