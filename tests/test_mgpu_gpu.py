"""GPU: the multi-GPU entry points of the C ABI (csrc/mgpu.cpp: one process, one context per device, RCCL bound at
run time) with the devices this box has.  The answers must not depend on the device count, so n_dev = 1 is compared
bit for bit with the single-context entry points and the oracle; boxes with more GPUs also run n_dev = 2."""
import numpy as np
import pytest
import torch

import points_matching_amd as pm
from points_matching_amd import synth
from points_matching_amd.api import PM_KNN_HINT_INTEGER

pytestmark = pytest.mark.gpu


def _ndevs():
    n = torch.cuda.device_count()
    return [1] + ([2] if n >= 2 else [])


@pytest.mark.parametrize("n_dev", _ndevs())
def test_mgpu_ransac_equals_single_context(ctx, oracle, n_dev):
    mg = pm.api.MultiGpu(n_dev)
    try:
        for n, H in ((2275, 10000), (300, 777), (9, 64)):
            x1, x2, _, _ = synth.two_view(n, seed=n, outlier_frac=0.3, noise_px=0.5)
            got = mg.ransac_fundamental(x1, x2, H, 1.0, 0x5EED)
            one = ctx.ransac_fundamental(x1, x2, H, 1.0, 0x5EED)
            want = oracle.ransac_fundamental(x1, x2, H, 1.0, 0x5EED, nthreads=8)
            for ref in (one, want):
                assert got[0] == ref[0] and got[4] == ref[4] and got[3] == ref[3], (n, H)
                assert (got[2] == ref[2]).all() and (got[1].view(np.uint64) == ref[1].view(np.uint64)).all(), (n, H)
        assert mg.ransac_fundamental(x1[:5], x2[:5], 100, 1.0, 1)[0] == pm.api.PM_E_TOO_FEW
    finally:
        mg.close()


@pytest.mark.parametrize("n_dev", _ndevs())
@pytest.mark.parametrize("kind", ["sift", "orb"])
def test_mgpu_match_ransac_equals_the_oracle_pipeline(oracle, n_dev, kind):
    nq, nt = 1500, 1300
    dim = 32 if kind == "orb" else 128
    w = synth.pair_workload(nq, nt, dim, seed=41, planted=0.4, kind=kind)
    mg = pm.api.MultiGpu(n_dev)
    try:
        rc, good, F, mask, ninl, key = mg.match_ransac(w["q"], w["t"], w["kp1"], w["kp2"], 0.8, 3000, 1.0, 0x5EED,
                                                       knn_flags=PM_KNN_HINT_INTEGER if kind == "sift" else 0)
    finally:
        mg.close()
    knn = (oracle.bf_knn_hamming if kind == "orb" else oracle.bf_knn_l2)(w["q"], w["t"], 2, nthreads=8)
    g_o = oracle.filter_ratio(knn, 0.8)
    assert good.size == g_o.size and (good["queryIdx"] == g_o["queryIdx"]).all() and (good["trainIdx"] == g_o["trainIdx"]).all()
    assert (good["distance"].view(np.uint32) == g_o["distance"].view(np.uint32)).all()
    xy1, xy2 = oracle.gather_points(w["kp1"], g_o["queryIdx"]), oracle.gather_points(w["kp2"], g_o["trainIdx"])
    rc_o, F_o, mask_o, ninl_o, key_o = oracle.ransac_fundamental(xy1, xy2, 3000, 1.0, 0x5EED, nthreads=8)
    assert rc == rc_o == 0 and key == key_o and ninl == ninl_o
    assert (mask == mask_o).all() and (F.view(np.uint64) == F_o.view(np.uint64)).all()
