"""ctypes binding of the CPU oracle (oracle/libpm_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (points_matching_amd/) never imports this module.
Parity status: see the header of oracle/pm_oracle.c ("parity unpinned" for the OpenCV-internal
arithmetic; the in-tree reference logic main.cpp:49-79, :89-91, :103-123 is pinned exactly).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# PM_ORACLE_LIB: another build of the same source (oracle/Makefile `sanitize`: AddressSanitizer + UBSan), tests only
_LIB_PATH = os.environ.get("PM_ORACLE_LIB") or os.path.join(_HERE, "libpm_oracle.so")

MATCH_DTYPE = np.dtype([("queryIdx", "<i4"), ("trainIdx", "<i4"), ("imgIdx", "<i4"),
                        ("distance", "<f4")])


class RansacParams(C.Structure):
    _fields_ = [("hyp_begin", C.c_int64), ("hyp_end", C.c_int64), ("seed", C.c_uint64),
                ("thresh_px", C.c_float), ("error_kind", C.c_int32)]


def build(force=False):
    if os.environ.get("PM_ORACLE_LIB"):
        return _LIB_PATH
    if force or not os.path.exists(_LIB_PATH) or \
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "pm_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.pmo_l2sqr.restype = C.c_float
        _lib.pmo_format_match_list.restype = C.c_long
    return _lib


def _p(a, t=C.c_void_p):
    return a.ctypes.data_as(t)


def l2sqr(a, b):
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    return np.float32(lib().pmo_l2sqr(_p(a), _p(b), C.c_int(a.size)))


def bf_knn_l2(q, t, k, nthreads=1):
    q = np.ascontiguousarray(q, np.float32)
    t = np.ascontiguousarray(t, np.float32)
    dim = q.shape[1] if q.ndim == 2 else t.shape[1]
    out = np.zeros((q.shape[0], k), MATCH_DTYPE)
    rc = lib().pmo_bf_knn_l2_f32(_p(q), q.shape[0], _p(t), t.shape[0], dim, k, _p(out), nthreads)
    assert rc == 0, rc
    return out


def bf_knn_hamming(q, t, k, nthreads=1):
    q = np.ascontiguousarray(q, np.uint8)
    t = np.ascontiguousarray(t, np.uint8)
    nbytes = q.shape[1] if q.ndim == 2 else t.shape[1]
    out = np.zeros((q.shape[0], k), MATCH_DTYPE)
    rc = lib().pmo_bf_knn_hamming_u8(_p(q), q.shape[0], _p(t), t.shape[0], nbytes, k, _p(out),
                                     nthreads)
    assert rc == 0, rc
    return out


def filter_midpoint(m):
    m = np.ascontiguousarray(m, MATCH_DTYPE).reshape(-1)
    out = np.zeros(max(m.size, 1), MATCH_DTYPE)
    mn, mx, n = C.c_double(), C.c_double(), C.c_int()
    lib().pmo_filter_midpoint(_p(m), m.size, C.byref(mn), C.byref(mx), _p(out), C.byref(n))
    return out[:n.value].copy(), mn.value, mx.value


def filter_ratio(knn, ratio):
    knn = np.ascontiguousarray(knn, MATCH_DTYPE)
    nq, k = knn.shape
    out = np.zeros(max(nq, 1), MATCH_DTYPE)
    n = C.c_int()
    rc = lib().pmo_filter_ratio(_p(knn), nq, k, C.c_float(ratio), _p(out), C.byref(n))
    assert rc == 0
    return out[:n.value].copy()


def gather_points(kp_xy, idx):
    kp_xy = np.ascontiguousarray(kp_xy, np.float32)
    idx = np.ascontiguousarray(idx, np.int32)
    out = np.zeros((idx.size, 2), np.float32)
    rc = lib().pmo_gather_points(_p(kp_xy), kp_xy.shape[0], _p(idx), idx.size, _p(out))
    if rc != 0:
        raise IndexError("keypoint index out of range")
    return out


def format_match_list(m):
    m = np.ascontiguousarray(m, MATCH_DTYPE).reshape(-1)
    need = lib().pmo_format_match_list(_p(m), m.size, None, C.c_size_t(0))
    buf = C.create_string_buffer(need + 1)
    lib().pmo_format_match_list(_p(m), m.size, buf, C.c_size_t(need + 1))
    return buf.value.decode()


def sample8(seed, h, n):
    idx = np.zeros(8, np.int32)
    lib().pmo_sample8(C.c_uint64(seed), C.c_uint64(h), n, _p(idx))
    return idx


def solve8(p1, p2):
    p1 = np.ascontiguousarray(p1, np.float64)
    p2 = np.ascontiguousarray(p2, np.float64)
    F = np.zeros(9, np.float64)
    ok = lib().pmo_solve8(_p(p1), _p(p2), _p(F))
    return bool(ok), F.reshape(3, 3)


def hyp_model(xy1, xy2, seed, h):
    xy1 = np.ascontiguousarray(xy1, np.float32)
    xy2 = np.ascontiguousarray(xy2, np.float32)
    F = np.zeros(9, np.float64)
    F32 = np.zeros(9, np.float32)
    ok = lib().pmo_hyp_model(_p(xy1), _p(xy2), xy1.shape[0], C.c_uint64(seed), C.c_uint64(h),
                             _p(F), _p(F32))
    return bool(ok), F.reshape(3, 3), F32.reshape(3, 3)


def score(F32, xy1, xy2, thresh_px, kind=0):
    F32 = np.ascontiguousarray(F32, np.float32)
    xy1 = np.ascontiguousarray(xy1, np.float32)
    xy2 = np.ascontiguousarray(xy2, np.float32)
    mask = np.zeros(xy1.shape[0], np.uint8)
    cnt = lib().pmo_score(_p(F32), _p(xy1), _p(xy2), xy1.shape[0], C.c_float(thresh_px), kind,
                          _p(mask))
    return cnt, mask


def ransac_fundamental(xy1, xy2, iters, thresh_px, seed, kind=0, hyp_begin=0, nthreads=1):
    """Returns (status, F(3x3), mask, n_inliers, best_key)."""
    xy1 = np.ascontiguousarray(xy1, np.float32)
    xy2 = np.ascontiguousarray(xy2, np.float32)
    n = xy1.shape[0]
    prm = RansacParams(hyp_begin, iters, seed, thresh_px, kind)
    F = np.zeros(9, np.float64)
    mask = np.zeros(max(n, 1), np.uint8)
    ninl, key = C.c_int(), C.c_uint64()
    rc = lib().pmo_ransac_fundamental(_p(xy1), _p(xy2), n, C.byref(prm), _p(F), _p(mask),
                                      C.byref(ninl), C.byref(key), nthreads)
    return rc, F.reshape(3, 3), mask[:n], ninl.value, key.value


def ransac_model_from_hyp(xy1, xy2, hyp, thresh_px, seed, kind=0):
    xy1 = np.ascontiguousarray(xy1, np.float32)
    xy2 = np.ascontiguousarray(xy2, np.float32)
    n = xy1.shape[0]
    prm = RansacParams(0, 0, seed, thresh_px, kind)
    F = np.zeros(9, np.float64)
    mask = np.zeros(max(n, 1), np.uint8)
    ninl = C.c_int()
    rc = lib().pmo_ransac_model_from_hyp(_p(xy1), _p(xy2), n, C.byref(prm), C.c_int64(hyp), _p(F),
                                         _p(mask), C.byref(ninl))
    return rc, F.reshape(3, 3), mask[:n], ninl.value


class LmedsParams(C.Structure):
    _fields_ = [("hyp_begin", C.c_int64), ("hyp_end", C.c_int64), ("seed", C.c_uint64)]


def lmeds_fundamental(xy1, xy2, iters, seed, hyp_begin=0, nthreads=1):
    """SPEC S13-S15 (7-point + LMedS).  Returns (status, F(3x3), mask, n_inliers, best_model, median)."""
    xy1 = np.ascontiguousarray(xy1, np.float32)
    xy2 = np.ascontiguousarray(xy2, np.float32)
    n = xy1.shape[0]
    prm = LmedsParams(hyp_begin, iters, seed)
    F = np.zeros(9, np.float64)
    mask = np.zeros(max(n, 1), np.uint8)
    ninl, best, med = C.c_int(), C.c_int64(), C.c_double()
    rc = lib().pmo_lmeds_fundamental(_p(xy1), _p(xy2), n, C.byref(prm), _p(F), _p(mask), C.byref(ninl),
                                     C.byref(best), C.byref(med), nthreads)
    return rc, F.reshape(3, 3), mask[:n], ninl.value, best.value, med.value


class AdaptiveParams(C.Structure):
    _fields_ = [("max_iters", C.c_int64), ("confidence", C.c_double), ("thresh_px", C.c_float), ("pad", C.c_int32),
                ("seed", C.c_uint64)]


def ransac7_adaptive(xy1, xy2, max_iters, confidence, thresh_px, seed):
    """SPEC S16.  Returns (status, F(3x3), mask, n_inliers, best_model, iters_run)."""
    xy1 = np.ascontiguousarray(xy1, np.float32)
    xy2 = np.ascontiguousarray(xy2, np.float32)
    n = xy1.shape[0]
    prm = AdaptiveParams(max_iters, confidence, thresh_px, 0, seed)
    F = np.zeros(9, np.float64)
    mask = np.zeros(max(n, 1), np.uint8)
    ninl, best, it = C.c_int(), C.c_int64(), C.c_int()
    rc = lib().pmo_ransac7_adaptive(_p(xy1), _p(xy2), n, C.byref(prm), _p(F), _p(mask), C.byref(ninl), C.byref(best),
                                    C.byref(it))
    return rc, F.reshape(3, 3), mask[:n], ninl.value, best.value, it.value


def sample7(seed, h, n):
    idx = np.zeros(7, np.int32)
    lib().pmo_sample7(C.c_uint64(seed), C.c_uint64(h), n, _p(idx))
    return idx


def solve7(p1, p2):
    """p1, p2: 7x2 float64.  Returns (F (3,3,3), valid (3,))."""
    p1 = np.ascontiguousarray(p1, np.float64)
    p2 = np.ascontiguousarray(p2, np.float64)
    F = np.zeros(27, np.float64)
    valid = np.zeros(3, np.int32)
    lib().pmo_solve7(_p(p1), _p(p2), _p(F), _p(valid))
    return F.reshape(3, 3, 3), valid


def lmeds_median(F, xy1, xy2):
    xy1 = np.ascontiguousarray(xy1, np.float32)
    xy2 = np.ascontiguousarray(xy2, np.float32)
    F = np.ascontiguousarray(F, np.float64).reshape(9)
    scratch = np.zeros(xy1.shape[0], np.float32)
    lib().pmo_lmeds_median.restype = C.c_double
    return lib().pmo_lmeds_median(_p(F), _p(xy1), _p(xy2), xy1.shape[0], _p(scratch)), scratch


def epipolar_residuals(xy1, xy2, F, transposed=1):
    xy1 = np.ascontiguousarray(xy1, np.float32)
    xy2 = np.ascontiguousarray(xy2, np.float32)
    F = np.ascontiguousarray(F, np.float64).reshape(9)
    n = xy1.shape[0]
    r = np.zeros(max(n, 1), np.float64)
    mean = C.c_double()
    lib().pmo_epipolar_residuals(_p(xy1), _p(xy2), n, _p(F), transposed, _p(r), C.byref(mean))
    return r[:n], mean.value


def f_scale_f33(F):
    F = np.ascontiguousarray(F, np.float64).reshape(9).copy()
    lib().pmo_f_scale_f33(_p(F))
    return F.reshape(3, 3)


def epilines(xy, which_image, F):
    xy = np.ascontiguousarray(xy, np.float32)
    F = np.ascontiguousarray(F, np.float64).reshape(9)
    lines = np.zeros((xy.shape[0], 3), np.float32)
    lib().pmo_epilines(_p(xy), xy.shape[0], which_image, _p(F), _p(lines))
    return lines


def epiline_endpoints(lines, cols):
    lines = np.ascontiguousarray(lines, np.float32)
    out = np.zeros((lines.shape[0], 4), np.int32)
    lib().pmo_epiline_endpoints(_p(lines), lines.shape[0], cols, _p(out))
    return out


def flann_search(nodes, roots, train, q, k=1, checks=32, heap_cap=1024):
    """SPEC S17 search over an exported kd-forest (points_matching_amd.api.FlannIndex.export())."""
    nodes = np.ascontiguousarray(nodes)
    roots = np.ascontiguousarray(roots, np.int32)
    train = np.ascontiguousarray(train, np.float32)
    q = np.ascontiguousarray(q, np.float32)
    out = np.zeros((q.shape[0], k), MATCH_DTYPE)
    rc = lib().pmo_flann_search(_p(nodes), _p(roots), roots.size, _p(train), train.shape[0], train.shape[1], _p(q),
                                q.shape[0], k, checks, heap_cap, _p(out))
    assert rc == 0, rc
    return out
