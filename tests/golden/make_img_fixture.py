"""BASELINE config C1 fixture: img01.JPG <-> img02.JPG (the reference's two sample photographs,
`Points Matching/img01.JPG`, `img02.JPG`; named in the commented-out main.cpp:12-13).

Run in the BUILD container only (it reads /root/reference, which does not exist on the GPU box):
decodes both JPEGs with PIL, extracts keypoints + u8-valued 128-D descriptors with the build-owned
extractor tools/sift_numpy.py (the reference's own front-end, OpenCV SURF at main.cpp:22-40, is
out of scope and unavailable), and records what the CPU oracle makes of them:
  * 2-NN + ratio 0.8 match list, 10 000-hypothesis RANSAC-F (seed 0x5EED, tau 1 px): mask, F, key
  * the reference's literal flow: 1-NN + midpoint filter (main.cpp:46-69) and its RANSAC result
The .npz holds data only (keypoints, descriptors, expected outputs).

    python tests/golden/make_img_fixture.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

from PIL import Image                        # noqa: E402
from oracle import pm_oracle as O            # noqa: E402
from sift_numpy import detect_and_describe   # noqa: E402

REF = "/root/reference/Points Matching"


def main():
    imgs = [np.asarray(Image.open(os.path.join(REF, n)).convert("L"), np.float32) / 255.0
            for n in ("img01.JPG", "img02.JPG")]
    (kp1, d1), (kp2, d2) = (detect_and_describe(im) for im in imgs)
    q, t = d1.astype(np.float32), d2.astype(np.float32)
    out = {"kp1": kp1, "kp2": kp2, "desc1": d1, "desc2": d2, "image_size": np.array(imgs[0].shape[::-1])}
    # north-star flow: 2-NN + ratio + RANSAC
    knn = O.bf_knn_l2(q, t, 2)
    good = O.filter_ratio(knn, 0.8)
    xy1, xy2 = O.gather_points(kp1, good["queryIdx"]), O.gather_points(kp2, good["trainIdx"])
    rc, F, mask, ninl, key = O.ransac_fundamental(xy1, xy2, 10000, 1.0, 0x5EED, nthreads=8)
    assert rc == 0
    out.update(knn_idx=knn["trainIdx"], knn_dist_bits=knn["distance"].view(np.uint32),
               ratio_query=good["queryIdx"], ratio_train=good["trainIdx"], ratio_F_bits=F.reshape(9).view(np.uint64),
               ratio_mask=mask, ratio_key=np.array([key], np.uint64))
    # the reference's literal flow: 1-NN + midpoint filter
    m1 = O.bf_knn_l2(q, t, 1).reshape(-1)
    g1, mn, mx = O.filter_midpoint(m1)
    xy1, xy2 = O.gather_points(kp1, g1["queryIdx"]), O.gather_points(kp2, g1["trainIdx"])
    rc, F, mask, ninl2, key = O.ransac_fundamental(xy1, xy2, 10000, 1.0, 0x5EED, nthreads=8)
    assert rc == 0
    out.update(mid_query=g1["queryIdx"], mid_train=g1["trainIdx"], mid_minmax=np.array([mn, mx]),
               mid_F_bits=F.reshape(9).view(np.uint64), mid_mask=mask, mid_key=np.array([key], np.uint64))
    np.savez_compressed(os.path.join(HERE, "img01_img02_sift.npz"), **out)
    # image-pair-in fixtures for the C++ front-end (host/pm_features.cpp, pm_cli --img1/--img2): the two photographs at
    # half resolution as binary PGM (data derived from the reference's JPEGs; 164 KB each) and what the numpy twin of
    # the extractor (tools/sift_numpy.py) finds in exactly these pixels
    half = {}
    for name, src in (("img01", "img01.JPG"), ("img02", "img02.JPG")):
        im = Image.open(os.path.join(REF, src)).convert("L")
        im = im.resize((im.width // 2, im.height // 2), Image.BILINEAR)
        a = np.asarray(im, np.uint8)
        with open(os.path.join(HERE, name + "_half.pgm"), "wb") as f:
            f.write(b"P5\n%d %d\n255\n" % (a.shape[1], a.shape[0]))
            f.write(a.tobytes())
        kp, d = detect_and_describe(a.astype(np.float32) / 255.0)
        half[name + "_kp"], half[name + "_desc"] = kp, d
    np.savez_compressed(os.path.join(HERE, "img01_img02_half_features.npz"), **half)
    print("half resolution: %d / %d keypoints" % (half["img01_kp"].shape[0], half["img02_kp"].shape[0]))
    print("img01/img02: %d x %d keypoints, ratio matches %d (inliers %d), midpoint matches %d (inliers %d)"
          % (kp1.shape[0], kp2.shape[0], good.size, ninl, g1.size, ninl2))


if __name__ == "__main__":
    main()
