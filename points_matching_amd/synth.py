"""Deterministic synthetic inputs for the BASELINE configs.

The reference ships no usable inputs for its own path (main.cpp:14-15 reads img1.bmp/img2.bmp,
which are missing), so every workload is generated here from a seed:
  * SIFT-like descriptors: integer-valued floats in [0,255]   (what OpenCV SIFT emits)
  * SURF-like descriptors: unit-L2-norm general floats         (what main.cpp:37-40 emits)
  * ORB-like descriptors : 256-bit packed binary
  * two-view geometry with a known F_gt, inlier noise and gross outliers
"""
import numpy as np


def _planted(rng, nq, nt, frac):
    """Planted ground truth: query i < n_pl is a noisy copy of train row perm[i]."""
    n_pl = int(round(frac * min(nq, nt)))
    src = rng.permutation(nt)[:n_pl]
    return n_pl, src


def sift_like(nq, nt, dim=128, seed=0xC2, planted=0.5, sigma=0.05):
    """|N(0,1)| -> L2-normalise -> clip 0.2 -> renormalise -> x512 -> round -> saturate [0,255]."""
    rng = np.random.default_rng(seed)

    def quant(x):
        x = x / np.linalg.norm(x, axis=1, keepdims=True)
        x = np.minimum(x, 0.2)
        x = x / np.linalg.norm(x, axis=1, keepdims=True)
        return np.clip(np.rint(x * 512.0), 0, 255).astype(np.float32)

    t_raw = np.abs(rng.standard_normal((nt, dim)))
    q_raw = np.abs(rng.standard_normal((nq, dim)))
    n_pl, src = _planted(rng, nq, nt, planted)
    if n_pl:
        base = t_raw[src] / np.linalg.norm(t_raw[src], axis=1, keepdims=True)
        q_raw[:n_pl] = np.abs(base + sigma * rng.standard_normal((n_pl, dim)))
    truth = np.full(nq, -1, np.int32)
    truth[:n_pl] = src
    return quant(q_raw), quant(t_raw), truth


def surf_like(nq, nt, dim=128, seed=0xC2, planted=0.5, sigma=0.05):
    """N(0,1)^dim, L2-normalised: general (non-integer) floats."""
    rng = np.random.default_rng(seed)
    t = rng.standard_normal((nt, dim))
    t /= np.linalg.norm(t, axis=1, keepdims=True)
    q = rng.standard_normal((nq, dim))
    n_pl, src = _planted(rng, nq, nt, planted)
    if n_pl:
        q[:n_pl] = t[src] + sigma * rng.standard_normal((n_pl, dim))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    truth = np.full(nq, -1, np.int32)
    truth[:n_pl] = src
    return q.astype(np.float32), t.astype(np.float32), truth


def orb_like(nq, nt, nbytes=32, seed=0xC4, planted=0.5, flip=0.1):
    """iid Bernoulli(1/2) bits; planted copies flip each bit with probability `flip`."""
    rng = np.random.default_rng(seed)
    t = rng.integers(0, 256, (nt, nbytes), dtype=np.uint8)
    q = rng.integers(0, 256, (nq, nbytes), dtype=np.uint8)
    n_pl, src = _planted(rng, nq, nt, planted)
    if n_pl:
        flips = (rng.random((n_pl, nbytes * 8)) < flip)
        q[:n_pl] = t[src] ^ np.packbits(flips, axis=1)
    truth = np.full(nq, -1, np.int32)
    truth[:n_pl] = src
    return q, t, truth


def two_view(n, seed=0xC3, outlier_frac=0.3, noise_px=0.5, width=993, height=660, focal=1000.0):
    """3-D points in z in [4,12] seen by two 1000-px-focal cameras (baseline 1, small rotation).

    Returns xy1, xy2 (n x 2 float32 pixels), F_gt (3x3 float64, x2^T F x1 = 0, unit Frobenius
    norm) and the boolean ground-truth inlier flags.  Image size = img01/img02 (993 x 660).
    """
    rng = np.random.default_rng(seed)
    K = np.array([[focal, 0, width / 2.0], [0, focal, height / 2.0], [0, 0, 1.0]])
    ang = rng.uniform(-0.08, 0.08, 3)
    cx, cy, cz = np.cos(ang)
    sx, sy, sz = np.sin(ang)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    R = Rz @ Ry @ Rx
    t = np.array([1.0, 0.05, 0.02])
    t /= np.linalg.norm(t)
    z = rng.uniform(4.0, 12.0, n)
    X = np.stack([rng.uniform(-0.45, 0.45, n) * z * width / focal,
                  rng.uniform(-0.45, 0.45, n) * z * height / focal, z], axis=1)
    x1 = (K @ X.T).T
    x1 = x1[:, :2] / x1[:, 2:3]
    X2 = (R @ X.T).T + t
    x2 = (K @ X2.T).T
    x2 = x2[:, :2] / x2[:, 2:3]
    x1 = x1 + rng.normal(0, noise_px, x1.shape)
    x2 = x2 + rng.normal(0, noise_px, x2.shape)
    inl = np.ones(n, bool)
    n_out = int(round(outlier_frac * n))
    if n_out:
        bad = rng.permutation(n)[:n_out]
        inl[bad] = False
        x1[bad] = rng.uniform([0, 0], [width, height], (n_out, 2))
        x2[bad] = rng.uniform([0, 0], [width, height], (n_out, 2))
    tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
    Kinv = np.linalg.inv(K)
    F = Kinv.T @ tx @ R @ Kinv
    F /= np.linalg.norm(F)
    if F[2, 2] < 0:
        F = -F
    return np.ascontiguousarray(x1.astype(np.float32)), np.ascontiguousarray(x2.astype(np.float32)), F, inl


def _sift_quant(x):
    x = x / np.linalg.norm(x, axis=1, keepdims=True)
    x = np.minimum(x, 0.2)
    x = x / np.linalg.norm(x, axis=1, keepdims=True)
    return np.clip(np.rint(x * 512.0), 0, 255).astype(np.float32)


def pair_workload(nq=8192, nt=8192, dim=128, seed=0xC3, rank=0, planted=0.28, outlier_frac=0.3,
                  noise_px=0.5, sigma=0.05, kind="sift", width=993, height=660, focal=1000.0):
    """One image pair of BASELINE config C3/C4: descriptors + keypoints + geometry.

    The train image (descriptors `t`, keypoints `kp2`, 3-D scene, cameras) depends on `seed`
    only; the query image depends on (seed, rank), so the ranks of a multi-GPU run hold
    different query-row shards of one global problem against a replicated train set.
    A fraction `planted` of the query rows are noisy copies of distinct train rows; their
    keypoints are true projections of the same 3-D point (+N(0, noise_px)), except that
    `outlier_frac` of them get a uniform-random image-1 position (false matches that survive
    the ratio test).  kind: "sift" (integer-valued floats), "surf" (unit-norm floats), "orb"
    (dim = bytes per descriptor, uint8).
    """
    rt = np.random.default_rng([seed, 0])
    rq = np.random.default_rng([seed, 1, rank])
    # scene + cameras (train-side, rank-independent)
    K = np.array([[focal, 0, width / 2.0], [0, focal, height / 2.0], [0, 0, 1.0]])
    ang = rt.uniform(-0.08, 0.08, 3)
    ca, cb, cc = np.cos(ang)
    sa, sb, sc = np.sin(ang)
    R = (np.array([[cc, -sc, 0], [sc, cc, 0], [0, 0, 1]]) @ np.array([[cb, 0, sb], [0, 1, 0], [-sb, 0, cb]])
         @ np.array([[1, 0, 0], [0, ca, -sa], [0, sa, ca]]))
    tv = np.array([1.0, 0.05, 0.02])
    tv /= np.linalg.norm(tv)
    z = rt.uniform(4.0, 12.0, nt)
    X = np.stack([rt.uniform(-0.45, 0.45, nt) * z * width / focal,
                  rt.uniform(-0.45, 0.45, nt) * z * height / focal, z], axis=1)
    p2 = (K @ ((R @ X.T).T + tv).T).T
    kp2 = (p2[:, :2] / p2[:, 2:3] + rt.normal(0, noise_px, (nt, 2))).astype(np.float32)
    tx = np.array([[0, -tv[2], tv[1]], [tv[2], 0, -tv[0]], [-tv[1], tv[0], 0]])
    Kinv = np.linalg.inv(K)
    F = Kinv.T @ tx @ R @ Kinv
    F /= np.linalg.norm(F)
    if F[2, 2] < 0:
        F = -F
    # descriptors
    n_pl = int(round(planted * min(nq, nt)))
    src = rq.permutation(nt)[:n_pl]
    if kind == "orb":
        t = rt.integers(0, 256, (nt, dim), dtype=np.uint8)
        q = rq.integers(0, 256, (nq, dim), dtype=np.uint8)
        flips = rq.random((n_pl, dim * 8)) < 0.1
        q[:n_pl] = t[src] ^ np.packbits(flips, axis=1)
    elif kind == "sift":
        t_raw = np.abs(rt.standard_normal((nt, dim)))
        q_raw = np.abs(rq.standard_normal((nq, dim)))
        base = t_raw[src] / np.linalg.norm(t_raw[src], axis=1, keepdims=True)
        q_raw[:n_pl] = np.abs(base + sigma * rq.standard_normal((n_pl, dim)))
        t, q = _sift_quant(t_raw), _sift_quant(q_raw)
    else:
        t = rt.standard_normal((nt, dim))
        t /= np.linalg.norm(t, axis=1, keepdims=True)
        q = rq.standard_normal((nq, dim))
        q[:n_pl] = t[src] + sigma * rq.standard_normal((n_pl, dim))
        q /= np.linalg.norm(q, axis=1, keepdims=True)
        t, q = t.astype(np.float32), q.astype(np.float32)
    # query keypoints
    kp1 = rq.uniform([0, 0], [width, height], (nq, 2))
    p1 = (K @ X[src].T).T
    kp1[:n_pl] = p1[:, :2] / p1[:, 2:3] + rq.normal(0, noise_px, (n_pl, 2))
    true_inlier = np.zeros(nq, bool)
    true_inlier[:n_pl] = True
    n_bad = int(round(outlier_frac * n_pl))
    bad = rq.permutation(n_pl)[:n_bad]
    kp1[bad] = rq.uniform([0, 0], [width, height], (n_bad, 2))
    true_inlier[bad] = False
    # shuffle the query rows so planted rows are not a prefix
    perm = rq.permutation(nq)
    truth = np.full(nq, -1, np.int32)
    truth[:n_pl] = src
    # every array C-contiguous: the device entry points take raw pointers
    return {"q": np.ascontiguousarray(q[perm]), "t": np.ascontiguousarray(t),
            "kp1": np.ascontiguousarray(kp1[perm].astype(np.float32)), "kp2": np.ascontiguousarray(kp2),
            "truth": truth[perm], "true_inlier": true_inlier[perm], "F_gt": F}
