"""CPU: the C++ feature front-end of pm_cli (host/pm_features.cpp; slot of main.cpp:22-26, :36-40) on the reference's two
photographs (half-resolution PGM fixtures, tests/golden/make_img_fixture.py) against its numpy twin
(tools/sift_numpy.py, which made the C1 descriptor fixture).  The two implement the same published scheme with
different filter code (own separable Gaussian vs scipy), so keypoint sets and descriptors are compared, not bits."""
import os
import subprocess

import numpy as np
import pytest

from points_matching_amd import build, io

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def extracted(tmp_path_factory):
    build.build_host()
    d = tmp_path_factory.mktemp("feat")
    cmd = [build.HOST_BIN, "--img1", os.path.join(GOLD, "img01_half.pgm"), "--img2", os.path.join(GOLD, "img02_half.pgm"),
           "--extract-only", "--save-features", str(d / "f"), "--quiet"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    return {k: io.load_pmm(str(d / ("f_%s.pmm" % k))) for k in ("desc1", "desc2", "kp1", "kp2")}


def test_cpp_extractor_agrees_with_its_numpy_twin(extracted):
    g = np.load(os.path.join(GOLD, "img01_img02_half_features.npz"), allow_pickle=False)
    for img, kp_key, d_key in (("img01", "kp1", "desc1"), ("img02", "kp2", "desc2")):
        kp, desc = extracted[kp_key], extracted[d_key]
        kp_n, desc_n = g[img + "_kp"], g[img + "_desc"].astype(np.float32)
        assert desc.shape[1] == 128 and kp.shape == (desc.shape[0], 2)
        assert (desc == np.rint(desc)).all() and desc.min() >= 0 and desc.max() <= 255      # u8-valued floats: the f16 route's premise
        assert abs(kp.shape[0] - kp_n.shape[0]) <= 0.1 * kp_n.shape[0]
        # keypoints of the twin found at the same pixel
        d2 = ((kp[:, None, :] - kp_n[None, :, :]) ** 2).sum(axis=2)
        nearest = d2.argmin(axis=0)
        same = d2[nearest, np.arange(kp_n.shape[0])] == 0
        assert same.mean() > 0.9, same.mean()
        # ... carry nearly the same descriptor (filter taps differ in the last float bits; bins are u8)
        diff = np.abs(desc[nearest[same]] - desc_n[same])
        assert np.median(diff.max(axis=1)) <= 3 and diff.mean() < 0.5, (np.median(diff.max(axis=1)), diff.mean())


def test_descriptors_of_the_two_views_match(extracted):
    """Sanity of the front-end as a whole: nearest-neighbour ratio matching between the two views finds many pairs."""
    q, t = extracted["desc1"], extracted["desc2"]
    d = ((q[:, None, :] - t[None, :, :]) ** 2).sum(axis=2)
    srt = np.sort(d, axis=1)
    good = np.sqrt(srt[:, 0]) < 0.8 * np.sqrt(srt[:, 1])
    assert good.sum() > 60
