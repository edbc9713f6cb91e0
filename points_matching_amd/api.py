"""Python mirror of include/pm.h over ctypes (plumbing for tests and bench.py, not the product).

The product is libpm_hip.so (HIP kernels behind a C ABI) driven by the C++ host in host/;
this module only forwards numpy arrays / torch device pointers to that ABI.  There is NO CPU
fallback: if the library is missing or a call fails, PmError is raised.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PM_LIB_PATH") or os.path.join(_HERE, "libpm_hip.so")   # override: experiments only

MATCH_DTYPE = np.dtype([("queryIdx", "<i4"), ("trainIdx", "<i4"), ("imgIdx", "<i4"),
                        ("distance", "<f4")])
PM_MAX_K = 16
PM_KNN_FORCE_EXACT = 1
PM_KNN_FORCE_F32 = 2
PM_KNN_HINT_INTEGER = 4
PM_KNN_HINT_U8 = 8
PM_KNN_HINT_UNIT_NORM = 16
PM_ERR_SAMPSON = 0
PM_ERR_SYM_EPIPOLAR = 1
PM_OK, PM_E_INVALID, PM_E_TOO_FEW, PM_E_NO_MODEL, PM_E_HIP, PM_E_NOMEM, PM_E_UNSUPPORTED = \
    0, -1, -2, -3, -4, -5, -6

# every extern "C" symbol include/pm.h declares (tests check the library exports all of them)
EXPORTS = [
    "pm_ctx_create", "pm_ctx_destroy", "pm_ctx_set_stream", "pm_ctx_synchronize",
    "pm_ctx_timing_enable", "pm_ctx_timing_reset", "pm_ctx_timing_get", "pm_ctx_knn_diag_enable", "pm_ctx_knn_stats", "pm_ctx_knn_route",
    "pm_ctx_filter_fusion_status",
    "pm_last_error",
    "pm_status_string", "pm_version",
    "pm_bf_knn_l2_f32", "pm_bf_knn_l2_f32_dev", "pm_bf_knn_hamming_u8", "pm_bf_knn_hamming_u8_dev",
    "pm_filter_midpoint", "pm_filter_ratio", "pm_match_indices", "pm_gather_points",
    "pm_format_match_list",
    "pm_filter_ratio_gather_dev", "pm_filter_midpoint_gather_dev", "pm_concat_points_dev",
    "pm_ransac_fundamental", "pm_ransac_score_dev", "pm_ransac_score_devn", "pm_ransac_model_from_hyp",
    "pm_ransac_model_from_key_dev", "pm_ransac_run_dev", "pm_ransac_shard_parts_dev", "pm_ransac_finish_parts_dev",
    "pm_ctx_set_option", "pm_ctx_get_option", "pm_bf_knn_l2_ratio_dev",
    "pm_flann_build", "pm_flann_destroy", "pm_flann_knn_l2_f32", "pm_flann_knn_l2_f32_dev", "pm_flann_export",
    "pm_mgpu_create", "pm_mgpu_destroy", "pm_mgpu_size", "pm_mgpu_ctx", "pm_mgpu_ransac_fundamental", "pm_mgpu_match_ransac",
    "pm_mgpu_lane_ctx", "pm_mgpu_set_lanes", "pm_mgpu_set_train", "pm_mgpu_set_train_dev", "pm_mgpu_submit_dev", "pm_mgpu_collect",
    "pm_mgpu_allgather_latency", "pm_mgpu_batch_run", "pm_mgpu_batch_set_option",
    "pm_batch_create", "pm_batch_destroy", "pm_batch_run", "pm_batch_set_option", "pm_batch_set_desc_type", "pm_batch_set_host_threads",
    "pm_bf_knn_l2_u8", "pm_bf_knn_l2_u8_dev", "pm_bf_knn_l2_u8_ratio_dev", "pm_host_register", "pm_host_unregister",
    "pm_lmeds_fundamental", "pm_lmeds_fundamental_dev", "pm_lmeds_default_iters", "pm_ransac7_adaptive",
    "pm_epipolar_residuals", "pm_f_scale_f33", "pm_epilines", "pm_epiline_endpoints",
]


class PmError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__("pm status %d: %s" % (status, msg))
        self.status = status


class RansacParams(C.Structure):
    _fields_ = [("hyp_begin", C.c_int64), ("hyp_end", C.c_int64), ("seed", C.c_uint64),
                ("thresh_px", C.c_float), ("error_kind", C.c_int32)]


class PointsView(C.Structure):
    """pm_points_view: `parts` padded blocks of correspondences with device-side counts (include/pm.h)."""
    _fields_ = [("xy1", C.c_void_p), ("xy2", C.c_void_p), ("counts", C.c_void_p), ("parts", C.c_int32),
                ("cap", C.c_int32), ("pitch_xy", C.c_int64), ("pitch_cnt", C.c_int32), ("reserved", C.c_int32)]


RANSAC_RECORD_DTYPE = np.dtype([("key", "<u8"), ("F", "<f8", (9,))])       # pm_ransac_record, 80 bytes
PM_MAX_PARTS = 64
PM_OPT_RANSAC_PATH, PM_OPT_SCORE_OPERANDS, PM_OPT_HAMMING_ROUTE, PM_OPT_KNN_F16_WAVES, PM_OPT_FILTER_FUSION = 1, 2, 3, 4, 5
PM_OPT_KNN_STAGING = 6
PM_OPT_KNN_WG_PER_CU = 7
PM_OPT_KNN_XCD_TILE = 8
PM_OPT_KNN_GENERAL_F16 = 9
PM_OPT_KNN_SEEDED = 10
PM_OPT_KNN_U8_GROUP = 11
PM_OPT_KNN_RING = 12
PM_OPT_KNN_U8_REFINE = 13
PM_OPT_KNN_RING_PROLOGUE = 14
PM_OPT_KNN_WIDE = 15
PM_OPT_KNN_PREP_ROWS = 16
PM_OPT_RANSAC_FORM = 17
PM_OPT_RANSAC_WG_IDS = 18
PM_OPT_HAMMING_REFINE = 19


_lib = None


def lib():
    """Loads libpm_hip.so; raises (never falls back) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise PmError(PM_E_UNSUPPORTED,
                          "%s is missing: run `python -m points_matching_amd.build`" % LIB_PATH)
        # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64 and cannot
        # initialise after another copy (the /opt/rocm one libpm_hip.so would pull in) has taken
        # the device.  Importing torch first makes libpm_hip.so bind to torch's runtime, which is
        # also what lets bench.py share torch streams/tensors with the library.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        _lib = C.CDLL(LIB_PATH)
        _lib.pm_last_error.restype = C.c_char_p
        _lib.pm_status_string.restype = C.c_char_p
        _lib.pm_format_match_list.restype = C.c_long
    return _lib


def _check(rc):
    if rc != PM_OK:
        raise PmError(rc, (lib().pm_last_error() or b"").decode() or
                      lib().pm_status_string(rc).decode())


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def ransac_key(inliers, hyp):
    return (int(inliers) << 32) | (0xFFFFFFFF - int(hyp))


def ransac_key_hyp(key):
    return 0xFFFFFFFF - (int(key) & 0xFFFFFFFF)


def ransac_key_inliers(key):
    return int(key) >> 32


# ---- host-side stages (no GPU needed) ---------------------------------------------------------

def filter_midpoint(m):
    """main.cpp:49-69.  Returns (good, minMatch, maxMatch)."""
    m = np.ascontiguousarray(m, MATCH_DTYPE).reshape(-1)
    out = np.zeros(max(m.size, 1), MATCH_DTYPE)
    mn, mx, n = C.c_double(), C.c_double(), C.c_int()
    _check(lib().pm_filter_midpoint(_p(m), m.size, C.byref(mn), C.byref(mx), _p(out), C.byref(n)))
    return out[:n.value].copy(), mn.value, mx.value


def filter_ratio(knn, ratio):
    knn = np.ascontiguousarray(knn, MATCH_DTYPE)
    nq, k = knn.shape
    out = np.zeros(max(nq, 1), MATCH_DTYPE)
    n = C.c_int()
    _check(lib().pm_filter_ratio(_p(knn), nq, k, C.c_float(ratio), _p(out), C.byref(n)))
    return out[:n.value].copy()


def match_indices(m):
    m = np.ascontiguousarray(m, MATCH_DTYPE).reshape(-1)
    qi = np.zeros(m.size, np.int32)
    ti = np.zeros(m.size, np.int32)
    _check(lib().pm_match_indices(_p(m), m.size, _p(qi), _p(ti)))
    return qi, ti


def gather_points(kp_xy, idx):
    kp_xy = np.ascontiguousarray(kp_xy, np.float32).reshape(-1, 2)
    idx = np.ascontiguousarray(idx, np.int32)
    out = np.zeros((idx.size, 2), np.float32)
    _check(lib().pm_gather_points(_p(kp_xy), kp_xy.shape[0], _p(idx), idx.size, _p(out)))
    return out


def format_match_list(m):
    m = np.ascontiguousarray(m, MATCH_DTYPE).reshape(-1)
    need = lib().pm_format_match_list(_p(m), m.size, None, C.c_size_t(0))
    buf = C.create_string_buffer(need + 1)
    lib().pm_format_match_list(_p(m), m.size, buf, C.c_size_t(need + 1))
    return buf.value.decode()


def epipolar_residuals(xy1, xy2, F, transposed=1):
    xy1 = np.ascontiguousarray(xy1, np.float32)
    xy2 = np.ascontiguousarray(xy2, np.float32)
    F = np.ascontiguousarray(F, np.float64).reshape(9)
    n = xy1.shape[0]
    r = np.zeros(max(n, 1), np.float64)
    mean = C.c_double()
    _check(lib().pm_epipolar_residuals(_p(xy1), _p(xy2), n, _p(F), transposed, _p(r), C.byref(mean)))
    return r[:n], mean.value


def f_scale_f33(F):
    F = np.ascontiguousarray(F, np.float64).reshape(9).copy()
    _check(lib().pm_f_scale_f33(_p(F)))
    return F.reshape(3, 3)


def epilines(xy, which_image, F):
    xy = np.ascontiguousarray(xy, np.float32)
    F = np.ascontiguousarray(F, np.float64).reshape(9)
    lines = np.zeros((xy.shape[0], 3), np.float32)
    _check(lib().pm_epilines(_p(xy), xy.shape[0], which_image, _p(F), _p(lines)))
    return lines


def epiline_endpoints(lines, cols):
    lines = np.ascontiguousarray(lines, np.float32)
    out = np.zeros((lines.shape[0], 4), np.int32)
    _check(lib().pm_epiline_endpoints(_p(lines), lines.shape[0], cols, _p(out)))
    return out


# ---- GPU context ------------------------------------------------------------------------------

class Context:
    """pm_ctx wrapper: one HIP device + stream + scratch arena."""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        _check(lib().pm_ctx_create(device, C.byref(self._h)))

    def close(self):
        if self._h:
            lib().pm_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def set_stream(self, stream_handle):
        _check(lib().pm_ctx_set_stream(self._h, C.c_void_p(stream_handle or 0)))

    def synchronize(self):
        _check(lib().pm_ctx_synchronize(self._h))

    def timing_enable(self, on=True):
        _check(lib().pm_ctx_timing_enable(self._h, int(on)))

    def timing_reset(self):
        _check(lib().pm_ctx_timing_reset(self._h))

    def timing_get(self, name):
        ms, n = C.c_double(), C.c_int()
        _check(lib().pm_ctx_timing_get(self._h, name.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def set_option(self, option, value):
        _check(lib().pm_ctx_set_option(self._h, option, value))

    def get_option(self, option):
        v = C.c_int()
        _check(lib().pm_ctx_get_option(self._h, option, C.byref(v)))
        return v.value

    def ransac_shard_parts_dev(self, view, hyp_begin, hyp_end, thresh_px, seed, drec_ptr, kind=PM_ERR_SAMPSON):
        prm = RansacParams(hyp_begin, hyp_end, seed, thresh_px, kind)
        _check(lib().pm_ransac_shard_parts_dev(self._h, C.byref(view), C.byref(prm), C.c_void_p(drec_ptr)))

    def ransac_finish_parts_dev(self, view, thresh_px, drecs_ptr, n_records, dkey_ptr, dF_ptr, dmask_ptr, mask_len,
                                dninl_ptr, dntotal_ptr=0, kind=PM_ERR_SAMPSON):
        prm = RansacParams(0, 0, 0, thresh_px, kind)
        _check(lib().pm_ransac_finish_parts_dev(self._h, C.byref(view), C.byref(prm), C.c_void_p(drecs_ptr), n_records,
                                                C.c_void_p(dkey_ptr or 0), C.c_void_p(dF_ptr or 0), C.c_void_p(dmask_ptr),
                                                mask_len, C.c_void_p(dninl_ptr or 0), C.c_void_p(dntotal_ptr or 0)))

    def knn_diag_enable(self, on=True):
        _check(lib().pm_ctx_knn_diag_enable(self._h, int(on)))

    def filter_fusion_gave_up(self):
        g = C.c_int()
        _check(lib().pm_ctx_filter_fusion_status(self._h, C.byref(g)))
        return g.value

    def knn_stats(self):
        r, nf = C.c_int(), C.c_int()
        _check(lib().pm_ctx_knn_stats(self._h, C.byref(r), C.byref(nf)))
        rt = C.c_int()
        _check(lib().pm_ctx_knn_route(self._h, C.byref(rt)))
        return {"rescans": r.value, "nonfinite": nf.value, "route": rt.value}

    def filter_ratio_gather_dev(self, dknn_ptr, nq, k, ratio, dkp1_ptr, dkp2_ptr, dgood_ptr, dxy1_ptr,
                                dxy2_ptr, dn_ptr):
        _check(lib().pm_filter_ratio_gather_dev(self._h, C.c_void_p(dknn_ptr), nq, k, C.c_float(ratio),
                                                C.c_void_p(dkp1_ptr), C.c_void_p(dkp2_ptr),
                                                C.c_void_p(dgood_ptr), C.c_void_p(dxy1_ptr),
                                                C.c_void_p(dxy2_ptr), C.c_void_p(dn_ptr)))

    def filter_midpoint_gather_dev(self, dm_ptr, n, k, dkp1_ptr, dkp2_ptr, dgood_ptr, dxy1_ptr, dxy2_ptr, dn_ptr,
                                   dminmax_ptr=0):
        _check(lib().pm_filter_midpoint_gather_dev(self._h, C.c_void_p(dm_ptr), n, k, C.c_void_p(dkp1_ptr),
                                                   C.c_void_p(dkp2_ptr), C.c_void_p(dgood_ptr), C.c_void_p(dxy1_ptr),
                                                   C.c_void_p(dxy2_ptr), C.c_void_p(dn_ptr),
                                                   C.c_void_p(dminmax_ptr or 0)))

    def concat_points_dev(self, dxy1_parts, dxy2_parts, dcounts, parts, stride, dxy1, dxy2, dn_total):
        _check(lib().pm_concat_points_dev(self._h, C.c_void_p(dxy1_parts), C.c_void_p(dxy2_parts),
                                          C.c_void_p(dcounts), parts, stride, C.c_void_p(dxy1),
                                          C.c_void_p(dxy2), C.c_void_p(dn_total)))

    def ransac_score_devn(self, dxy1_ptr, dxy2_ptr, n_max, dn_ptr, hyp_begin, hyp_end, thresh_px, seed,
                          dkey_ptr, kind=PM_ERR_SAMPSON):
        prm = RansacParams(hyp_begin, hyp_end, seed, thresh_px, kind)
        _check(lib().pm_ransac_score_devn(self._h, C.c_void_p(dxy1_ptr), C.c_void_p(dxy2_ptr), n_max,
                                          C.c_void_p(dn_ptr), C.byref(prm), C.c_void_p(dkey_ptr)))

    def ransac_run_dev(self, dxy1_ptr, dxy2_ptr, n_max, dn_ptr, hyp_begin, hyp_end, thresh_px, seed, dkey_ptr,
                       dF_ptr, dmask_ptr, dninl_ptr, kind=PM_ERR_SAMPSON):
        prm = RansacParams(hyp_begin, hyp_end, seed, thresh_px, kind)
        _check(lib().pm_ransac_run_dev(self._h, C.c_void_p(dxy1_ptr), C.c_void_p(dxy2_ptr), n_max,
                                       C.c_void_p(dn_ptr or 0), C.byref(prm), C.c_void_p(dkey_ptr),
                                       C.c_void_p(dF_ptr), C.c_void_p(dmask_ptr), C.c_void_p(dninl_ptr)))

    def ransac_model_from_key_dev(self, dxy1_ptr, dxy2_ptr, n_max, dn_ptr, thresh_px, seed, dkey_ptr, dF_ptr,
                                  dmask_ptr, dninl_ptr, kind=PM_ERR_SAMPSON):
        prm = RansacParams(0, 0, seed, thresh_px, kind)
        _check(lib().pm_ransac_model_from_key_dev(self._h, C.c_void_p(dxy1_ptr), C.c_void_p(dxy2_ptr), n_max,
                                                  C.c_void_p(dn_ptr or 0), C.byref(prm), C.c_void_p(dkey_ptr),
                                                  C.c_void_p(dF_ptr), C.c_void_p(dmask_ptr),
                                                  C.c_void_p(dninl_ptr)))

    def bf_knn_l2_u8(self, q, t, k):
        """u8 descriptor rows, host arrays (pm_bf_knn_l2_u8)."""
        q = np.ascontiguousarray(q, np.uint8)
        t = np.ascontiguousarray(t, np.uint8)
        nq, dim = q.shape
        out = np.zeros((nq, k), MATCH_DTYPE)
        _check(lib().pm_bf_knn_l2_u8(self._h, _p(q), nq, _p(t), t.shape[0], dim, k, _p(out)))
        return out

    def bf_knn_l2_u8_dev(self, dq_ptr, nq, dt_ptr, nt, dim, k, dout_ptr):
        _check(lib().pm_bf_knn_l2_u8_dev(self._h, C.c_void_p(dq_ptr), nq, C.c_void_p(dt_ptr), nt, dim, k, C.c_void_p(dout_ptr)))

    def bf_knn_l2_u8_ratio_dev(self, dq_ptr, nq, dt_ptr, nt, dim, ratio, dkp1_ptr, dkp2_ptr, dknn_ptr, dgood_ptr, dxy1_ptr,
                               dxy2_ptr, dn_ptr):
        _check(lib().pm_bf_knn_l2_u8_ratio_dev(self._h, C.c_void_p(dq_ptr), nq, C.c_void_p(dt_ptr), nt, dim, C.c_float(ratio),
                                               C.c_void_p(dkp1_ptr or 0), C.c_void_p(dkp2_ptr or 0), C.c_void_p(dknn_ptr),
                                               C.c_void_p(dgood_ptr), C.c_void_p(dxy1_ptr or 0), C.c_void_p(dxy2_ptr or 0),
                                               C.c_void_p(dn_ptr)))

    # -- matcher (main.cpp:46) -------------------------------------------------------------------
    def bf_knn_l2(self, q, t, k, flags=0):
        q = np.ascontiguousarray(q, np.float32)
        t = np.ascontiguousarray(t, np.float32)
        dim = q.shape[1]
        assert t.ndim == 2 and t.shape[1] == dim
        out = np.zeros((q.shape[0], k), MATCH_DTYPE)
        _check(lib().pm_bf_knn_l2_f32(self._h, _p(q), q.shape[0], _p(t), t.shape[0], dim, k, flags,
                                      _p(out)))
        return out

    def bf_knn_l2_dev(self, dq_ptr, nq, dt_ptr, nt, dim, k, dout_ptr, flags=0):
        _check(lib().pm_bf_knn_l2_f32_dev(self._h, C.c_void_p(dq_ptr), nq, C.c_void_p(dt_ptr), nt,
                                          dim, k, flags, C.c_void_p(dout_ptr)))

    def bf_knn_l2_ratio_dev(self, dq_ptr, nq, dt_ptr, nt, dim, flags, ratio, dkp1_ptr, dkp2_ptr, dknn_ptr, dgood_ptr,
                            dxy1_ptr, dxy2_ptr, dn_ptr):
        _check(lib().pm_bf_knn_l2_ratio_dev(self._h, C.c_void_p(dq_ptr), nq, C.c_void_p(dt_ptr), nt, dim, flags,
                                            C.c_float(ratio), C.c_void_p(dkp1_ptr or 0), C.c_void_p(dkp2_ptr or 0),
                                            C.c_void_p(dknn_ptr or 0), C.c_void_p(dgood_ptr), C.c_void_p(dxy1_ptr or 0),
                                            C.c_void_p(dxy2_ptr or 0), C.c_void_p(dn_ptr)))

    def bf_knn_hamming(self, q, t, k):
        q = np.ascontiguousarray(q, np.uint8)
        t = np.ascontiguousarray(t, np.uint8)
        nbytes = q.shape[1]
        assert t.ndim == 2 and t.shape[1] == nbytes
        out = np.zeros((q.shape[0], k), MATCH_DTYPE)
        _check(lib().pm_bf_knn_hamming_u8(self._h, _p(q), q.shape[0], _p(t), t.shape[0], nbytes, k,
                                          _p(out)))
        return out

    def bf_knn_hamming_dev(self, dq_ptr, nq, dt_ptr, nt, nbytes, k, dout_ptr):
        _check(lib().pm_bf_knn_hamming_u8_dev(self._h, C.c_void_p(dq_ptr), nq, C.c_void_p(dt_ptr),
                                              nt, nbytes, k, C.c_void_p(dout_ptr)))

    # -- robust F (main.cpp:95-98) ---------------------------------------------------------------
    def ransac_fundamental(self, xy1, xy2, iters, thresh_px, seed, kind=PM_ERR_SAMPSON,
                           hyp_begin=0):
        """Returns (status, F(3x3), mask, n_inliers, best_key); raises on anything other than
        PM_OK / PM_E_NO_MODEL / PM_E_TOO_FEW (those are data outcomes, reported as status)."""
        xy1 = np.ascontiguousarray(xy1, np.float32).reshape(-1, 2)
        xy2 = np.ascontiguousarray(xy2, np.float32).reshape(-1, 2)
        n = xy1.shape[0]
        prm = RansacParams(hyp_begin, iters, seed, thresh_px, kind)
        F = np.zeros(9, np.float64)
        mask = np.zeros(max(n, 1), np.uint8)
        ninl, key = C.c_int(), C.c_uint64()
        rc = lib().pm_ransac_fundamental(self._h, _p(xy1), _p(xy2), n, C.byref(prm), _p(F), _p(mask),
                                         C.byref(ninl), C.byref(key))
        if rc not in (PM_OK, PM_E_NO_MODEL, PM_E_TOO_FEW):
            _check(rc)
        return rc, F.reshape(3, 3), mask[:n], ninl.value, key.value

    def ransac_model_from_hyp(self, xy1, xy2, hyp, thresh_px, seed, kind=PM_ERR_SAMPSON):
        xy1 = np.ascontiguousarray(xy1, np.float32).reshape(-1, 2)
        xy2 = np.ascontiguousarray(xy2, np.float32).reshape(-1, 2)
        n = xy1.shape[0]
        prm = RansacParams(0, 0, seed, thresh_px, kind)
        F = np.zeros(9, np.float64)
        mask = np.zeros(max(n, 1), np.uint8)
        ninl = C.c_int()
        rc = lib().pm_ransac_model_from_hyp(self._h, _p(xy1), _p(xy2), n, C.byref(prm),
                                            C.c_int64(hyp), _p(F), _p(mask), C.byref(ninl))
        if rc not in (PM_OK, PM_E_NO_MODEL, PM_E_TOO_FEW):
            _check(rc)
        return rc, F.reshape(3, 3), mask[:n], ninl.value

    def ransac_score_dev(self, dxy1_ptr, dxy2_ptr, n, hyp_begin, hyp_end, thresh_px, seed,
                         dkey_ptr, kind=PM_ERR_SAMPSON):
        prm = RansacParams(hyp_begin, hyp_end, seed, thresh_px, kind)
        _check(lib().pm_ransac_score_dev(self._h, C.c_void_p(dxy1_ptr), C.c_void_p(dxy2_ptr), n,
                                         C.byref(prm), C.c_void_p(dkey_ptr)))


class LmedsParams(C.Structure):
    _fields_ = [("hyp_begin", C.c_int64), ("hyp_end", C.c_int64), ("seed", C.c_uint64)]


class AdaptiveParams(C.Structure):
    _fields_ = [("max_iters", C.c_int64), ("confidence", C.c_double), ("thresh_px", C.c_float), ("reserved", C.c_int32),
                ("seed", C.c_uint64)]


def ransac7_adaptive(ctx, xy1, xy2, max_iters, confidence, thresh_px, seed):
    """Adaptive-iteration RANSAC over 7-point models (SPEC S16).
    Returns (status, F(3x3), mask, n_inliers, best_model, iters_run)."""
    xy1 = np.ascontiguousarray(xy1, np.float32).reshape(-1, 2)
    xy2 = np.ascontiguousarray(xy2, np.float32).reshape(-1, 2)
    n = xy1.shape[0]
    prm = AdaptiveParams(max_iters, confidence, thresh_px, 0, seed)
    F = np.zeros(9, np.float64)
    mask = np.zeros(max(n, 1), np.uint8)
    ninl, best, it = C.c_int(), C.c_int64(), C.c_int()
    rc = lib().pm_ransac7_adaptive(ctx._h, _p(xy1), _p(xy2), n, C.byref(prm), _p(F), _p(mask), C.byref(ninl),
                                   C.byref(best), C.byref(it))
    if rc not in (PM_OK, PM_E_NO_MODEL, PM_E_TOO_FEW):
        _check(rc)
    return rc, F.reshape(3, 3), mask[:n], ninl.value, best.value, it.value


def lmeds_default_iters(confidence=0.99, outlier_ratio=0.45):
    return lib().pm_lmeds_default_iters(C.c_double(confidence), C.c_double(outlier_ratio))


def lmeds_fundamental(ctx, xy1, xy2, iters, seed, hyp_begin=0):
    """7-point + LMedS (SPEC S13-S15).  Returns (status, F(3x3), mask, n_inliers, best_model, median)."""
    xy1 = np.ascontiguousarray(xy1, np.float32).reshape(-1, 2)
    xy2 = np.ascontiguousarray(xy2, np.float32).reshape(-1, 2)
    n = xy1.shape[0]
    prm = LmedsParams(hyp_begin, iters, seed)
    F = np.zeros(9, np.float64)
    mask = np.zeros(max(n, 1), np.uint8)
    ninl, best, med = C.c_int(), C.c_int64(), C.c_double()
    rc = lib().pm_lmeds_fundamental(ctx._h, _p(xy1), _p(xy2), n, C.byref(prm), _p(F), _p(mask), C.byref(ninl),
                                    C.byref(best), C.byref(med))
    if rc not in (PM_OK, PM_E_NO_MODEL, PM_E_TOO_FEW):
        _check(rc)
    return rc, F.reshape(3, 3), mask[:n], ninl.value, best.value, med.value


# ---- batch of image pairs (BASELINE config C5) ---------------------------------------------------

class PairJob(C.Structure):
    _fields_ = [("desc1", C.c_void_p), ("desc2", C.c_void_p), ("kp1_xy", C.c_void_p), ("kp2_xy", C.c_void_p),
                ("n1", C.c_int32), ("n2", C.c_int32)]


class PairResult(C.Structure):
    _fields_ = [("F", C.c_double * 9), ("best_key", C.c_uint64), ("n_good", C.c_int32), ("n_inliers", C.c_int32),
                ("status", C.c_int32), ("reserved", C.c_int32)]


class PairBatch:
    """pm_batch wrapper: `n_lanes` streams, each running whole pairs (H2D -> match -> ratio+gather ->
    RANSAC-F -> D2H).  Jobs are (desc1_ptr, n1, desc2_ptr, n2, kp1_ptr, kp2_ptr) with HOST addresses
    (ideally page-locked: torch pinned tensors, or host_register())."""

    def __init__(self, device, n_lanes, max_n1, max_n2, dim):
        self._h = C.c_void_p()
        self.max_n1, self.dim = max_n1, dim
        _check(lib().pm_batch_create(device, n_lanes, max_n1, max_n2, dim, C.byref(self._h)))

    def close(self):
        if self._h:
            lib().pm_batch_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, option, value):
        _check(lib().pm_batch_set_option(self._h, option, value))

    def set_desc_u8(self, on=True):
        """the jobs' desc1 / desc2 point at uint8 rows"""
        _check(lib().pm_batch_set_desc_type(self._h, int(bool(on))))

    def set_host_threads(self, n):
        """0 = automatic (two host threads with >= 4 lanes), 1, 2"""
        _check(lib().pm_batch_set_host_threads(self._h, int(n)))

    @staticmethod
    def make_jobs(jobs):
        arr = (PairJob * len(jobs))()
        for a, (d1, n1, d2, n2, k1, k2) in zip(arr, jobs):
            a.desc1, a.n1, a.desc2, a.n2, a.kp1_xy, a.kp2_xy = d1, n1, d2, n2, k1, k2
        return arr

    def run(self, jobs, ratio, iters, thresh_px, seed, knn_flags=0, kind=PM_ERR_SAMPSON, want_good=False,
            want_masks=False):
        """jobs: list of tuples or a prepared (PairJob * n) array.  Returns (results array, good, masks)."""
        arr = jobs if isinstance(jobs, C.Array) else self.make_jobs(jobs)
        n = len(arr)
        res = (PairResult * n)()
        good = np.zeros((n, self.max_n1), MATCH_DTYPE) if want_good else None
        masks = np.zeros((n, self.max_n1), np.uint8) if want_masks else None
        prm = RansacParams(0, iters, seed, thresh_px, kind)
        _check(lib().pm_batch_run(self._h, arr, n, C.c_float(ratio), knn_flags, C.byref(prm), res,
                                  _p(good) if want_good else None, _p(masks) if want_masks else None))
        return res, good, masks


def host_register(arr):
    _check(lib().pm_host_register(C.c_void_p(arr.ctypes.data), arr.nbytes))


def host_unregister(arr):
    _check(lib().pm_host_unregister(C.c_void_p(arr.ctypes.data)))


# ---- the path over the GPUs of one node (single process, RCCL behind the C ABI) ----------------------

class MultiGpu:
    """pm_mgpu wrapper."""

    def __init__(self, n_dev, devices=None):
        self._h = C.c_void_p()
        arr = (C.c_int * n_dev)(*devices) if devices is not None else None
        _check(lib().pm_mgpu_create(n_dev, arr, C.byref(self._h)))
        self.n = n_dev

    def close(self):
        if self._h:
            lib().pm_mgpu_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def ransac_fundamental(self, xy1, xy2, iters, thresh_px, seed, kind=PM_ERR_SAMPSON, hyp_begin=0):
        xy1 = np.ascontiguousarray(xy1, np.float32).reshape(-1, 2)
        xy2 = np.ascontiguousarray(xy2, np.float32).reshape(-1, 2)
        n = xy1.shape[0]
        prm = RansacParams(hyp_begin, iters, seed, thresh_px, kind)
        F = np.zeros(9, np.float64)
        mask = np.zeros(max(n, 1), np.uint8)
        ninl, key = C.c_int(), C.c_uint64()
        rc = lib().pm_mgpu_ransac_fundamental(self._h, _p(xy1), _p(xy2), n, C.byref(prm), _p(F), _p(mask), C.byref(ninl),
                                              C.byref(key))
        if rc not in (PM_OK, PM_E_NO_MODEL, PM_E_TOO_FEW):
            _check(rc)
        return rc, F.reshape(3, 3), mask[:n], ninl.value, key.value

    def match_ransac(self, desc1, desc2, kp1, kp2, ratio, iters, thresh_px, seed, knn_flags=0, kind=PM_ERR_SAMPSON):
        """Returns (status, good, F(3x3), mask, n_inliers, best_key)."""
        binary = desc1.dtype == np.uint8
        desc1 = np.ascontiguousarray(desc1, np.uint8 if binary else np.float32)
        desc2 = np.ascontiguousarray(desc2, desc1.dtype)
        kp1 = np.ascontiguousarray(kp1, np.float32).reshape(-1, 2)
        kp2 = np.ascontiguousarray(kp2, np.float32).reshape(-1, 2)
        n1, dim = desc1.shape
        prm = RansacParams(0, iters, seed, thresh_px, kind)
        good = np.zeros(n1, MATCH_DTYPE)
        F = np.zeros(9, np.float64)
        mask = np.zeros(n1, np.uint8)
        ngood, ninl, key = C.c_int(), C.c_int(), C.c_uint64()
        rc = lib().pm_mgpu_match_ransac(self._h, _p(desc1), n1, _p(desc2), desc2.shape[0], dim, int(binary), _p(kp1), _p(kp2),
                                        C.c_float(ratio), knn_flags, C.byref(prm), _p(good), C.byref(ngood), _p(F), _p(mask),
                                        C.byref(ninl), C.byref(key))
        if rc not in (PM_OK, PM_E_NO_MODEL, PM_E_TOO_FEW):
            _check(rc)
        return rc, good[:ngood.value].copy(), F.reshape(3, 3), mask[:ngood.value].copy(), ninl.value, key.value


    # ---- streamed form: resident train side, device pointers in, tickets out -------------------------------------
    def set_lanes(self, n_lanes):
        _check(lib().pm_mgpu_set_lanes(self._h, n_lanes))

    def set_option(self, option, value):
        _check(lib().pm_mgpu_batch_set_option(self._h, option, value))

    def set_train(self, desc2, kp2):
        binary = desc2.dtype == np.uint8
        desc2 = np.ascontiguousarray(desc2, np.uint8 if binary else np.float32)
        kp2 = np.ascontiguousarray(kp2, np.float32).reshape(-1, 2)
        _check(lib().pm_mgpu_set_train(self._h, _p(desc2), desc2.shape[0], desc2.shape[1], int(binary), _p(kp2)))

    def set_train_dev(self, desc2_ptrs, n2, dim, binary, kp2_ptrs):
        """per-device device pointers (the caller keeps the buffers alive)"""
        a = (C.c_void_p * self.n)(*desc2_ptrs)
        b = (C.c_void_p * self.n)(*kp2_ptrs)
        _check(lib().pm_mgpu_set_train_dev(self._h, a, n2, dim, int(binary), b))

    def submit_dev(self, desc1_ptrs, rows, kp1_ptrs, ratio, iters, thresh_px, seed, knn_flags=0, kind=PM_ERR_SAMPSON):
        a = (C.c_void_p * self.n)(*desc1_ptrs)
        b = (C.c_void_p * self.n)(*kp1_ptrs)
        r = (C.c_int32 * self.n)(*rows)
        prm = RansacParams(0, iters, seed, thresh_px, kind)
        t = C.c_int()
        _check(lib().pm_mgpu_submit_dev(self._h, a, r, b, C.c_float(ratio), knn_flags, C.byref(prm), C.byref(t)))
        return t.value

    def collect(self, ticket, n1=0, want_good=False, want_mask=False):
        """Returns (PairResult, good records or None, mask or None); n1 = total query rows of the pair (buffer sizes)."""
        res = PairResult()
        good = np.zeros(max(n1, 1), MATCH_DTYPE) if want_good else None
        mask = np.zeros(max(n1, 1), np.uint8) if want_mask else None
        _check(lib().pm_mgpu_collect(self._h, ticket, C.byref(res), _p(good) if want_good else None, _p(mask) if want_mask else None))
        ng = max(res.n_good, 0)
        return res, (good[:ng].copy() if want_good else None), (mask[:ng].copy() if want_mask else None)

    def allgather_latency(self, bytes_per_device, reps=200):
        us = C.c_double()
        _check(lib().pm_mgpu_allgather_latency(self._h, bytes_per_device, reps, C.byref(us)))
        return us.value

    def batch_run(self, jobs, n_lanes, max_n1, max_n2, dim, ratio, iters, thresh_px, seed, knn_flags=0, kind=PM_ERR_SAMPSON,
                  want_good=False, want_masks=False):
        """BASELINE config C5 over the devices (pair p -> device p mod n): like PairBatch.run."""
        arr = jobs if isinstance(jobs, C.Array) else PairBatch.make_jobs(jobs)
        n = len(arr)
        res = (PairResult * n)()
        good = np.zeros((n, max_n1), MATCH_DTYPE) if want_good else None
        masks = np.zeros((n, max_n1), np.uint8) if want_masks else None
        prm = RansacParams(0, iters, seed, thresh_px, kind)
        _check(lib().pm_mgpu_batch_run(self._h, n_lanes, max_n1, max_n2, dim, arr, n, C.c_float(ratio), knn_flags, C.byref(prm), res,
                                       _p(good) if want_good else None, _p(masks) if want_masks else None))
        return res, good, masks


# ---- FlannBasedMatcher-compatible approximate matcher (main.cpp:44; SPEC S17) ---------------------------

class FlannParams(C.Structure):
    _fields_ = [("trees", C.c_int32), ("checks", C.c_int32), ("seed", C.c_uint64)]


FLANN_NODE_DTYPE = np.dtype([("child1", "<i4"), ("child2", "<i4"), ("divfeat", "<i4"), ("divval", "<f4")])


class FlannIndex:
    """pm_flann_index wrapper: kd-forest of the train descriptors (built on the host, searched on the GPU)."""

    def __init__(self, ctx, train, trees=4, checks=32, seed=0):
        self._ctx = ctx
        self.train = np.ascontiguousarray(train, np.float32)
        self.trees, self.checks = trees, checks
        self._h = C.c_void_p()
        prm = FlannParams(trees, checks, seed)
        _check(lib().pm_flann_build(ctx._h, _p(self.train), self.train.shape[0], self.train.shape[1], C.byref(prm),
                                    C.byref(self._h)))

    def close(self):
        if self._h:
            lib().pm_flann_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def knn(self, q, k=1):
        q = np.ascontiguousarray(q, np.float32)
        out = np.zeros((q.shape[0], k), MATCH_DTYPE)
        _check(lib().pm_flann_knn_l2_f32(self._ctx._h, self._h, _p(q), q.shape[0], k, _p(out)))
        return out

    def knn_dev(self, dq_ptr, nq, k, dout_ptr):
        _check(lib().pm_flann_knn_l2_f32_dev(self._ctx._h, self._h, C.c_void_p(dq_ptr), nq, k, C.c_void_p(dout_ptr)))

    def export(self):
        """(nodes structured array, roots int32[trees])"""
        n = C.c_int32()
        roots = np.zeros(16, np.int32)
        _check(lib().pm_flann_export(self._h, C.byref(n), _p(roots), None, 0))
        nodes = np.zeros(n.value, FLANN_NODE_DTYPE)
        _check(lib().pm_flann_export(self._h, C.byref(n), _p(roots), _p(nodes), n.value))
        return nodes, roots[:self.trees].copy()
