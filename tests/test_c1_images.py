"""BASELINE config C1: img01.JPG <-> img02.JPG.  Inputs are the committed descriptor fixture
(tests/golden/img01_img02_sift.npz, made by tests/golden/make_img_fixture.py from the reference's
two photographs); the expectation is the CPU oracle's answer recorded with it.
CPU: the oracle still reproduces it.  GPU: the HIP path yields the identical match list, inlier
set and F — the north-star clause "inlier set identical to CPU on img01/img02"."""
import os

import numpy as np
import pytest

import points_matching_amd as pm

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "img01_img02_sift.npz")


def _load():
    return np.load(GOLD, allow_pickle=False)


def _check_flow(matcher, ransac, filt_ratio, filt_mid, gather, g):
    q, t = g["desc1"].astype(np.float32), g["desc2"].astype(np.float32)
    knn = matcher(q, t, 2)
    assert (knn["trainIdx"] == g["knn_idx"]).all()
    assert (knn["distance"].view(np.uint32) == g["knn_dist_bits"]).all()
    good = filt_ratio(knn, 0.8)
    assert (good["queryIdx"] == g["ratio_query"]).all() and (good["trainIdx"] == g["ratio_train"]).all()
    xy1, xy2 = gather(g["kp1"], good["queryIdx"]), gather(g["kp2"], good["trainIdx"])
    rc, F, mask, ninl, key = ransac(xy1, xy2, 10000, 1.0, 0x5EED)
    assert rc == 0 and key == int(g["ratio_key"][0])
    assert (mask == g["ratio_mask"]).all() and ninl == int(g["ratio_mask"].sum())
    assert (F.reshape(9).view(np.uint64) == g["ratio_F_bits"]).all()
    # the reference's literal flow (main.cpp:46-69): 1-NN + midpoint filter
    m1 = matcher(q, t, 1).reshape(-1)
    g1, mn, mx = filt_mid(m1)
    assert (g1["queryIdx"] == g["mid_query"]).all() and (g1["trainIdx"] == g["mid_train"]).all()
    assert mn == g["mid_minmax"][0] and mx == g["mid_minmax"][1]
    xy1, xy2 = gather(g["kp1"], g1["queryIdx"]), gather(g["kp2"], g1["trainIdx"])
    rc, F, mask, ninl, key = ransac(xy1, xy2, 10000, 1.0, 0x5EED)
    assert rc == 0 and key == int(g["mid_key"][0]) and (mask == g["mid_mask"]).all()
    assert (F.reshape(9).view(np.uint64) == g["mid_F_bits"]).all()
    return int(g["ratio_mask"].sum()), good.size


def test_c1_fixture_sane():
    g = _load()
    assert g["desc1"].dtype == np.uint8 and g["desc1"].shape[1] == 128 and g["kp1"].shape[0] == g["desc1"].shape[0]
    assert tuple(g["image_size"]) == (993, 660)
    assert g["ratio_mask"].sum() > 0.8 * g["ratio_mask"].size      # two views of one scene: mostly inliers


def test_c1_oracle_reproduces_fixture(oracle):
    _check_flow(oracle.bf_knn_l2, oracle.ransac_fundamental, oracle.filter_ratio, oracle.filter_midpoint,
                oracle.gather_points, _load())


@pytest.mark.gpu
def test_c1_hip_inlier_set_identical_to_cpu(ctx):
    ninl, nmatch = _check_flow(ctx.bf_knn_l2, ctx.ransac_fundamental, pm.api.filter_ratio, pm.api.filter_midpoint,
                               pm.api.gather_points, _load())
    assert ninl > 100
