"""Fixture of the image-pair-in path (pm_cli --img1/--img2; SURVEY.md 8f-2, main.cpp:14-15, :22-40, :46-98) on the reference's
own two photographs at half resolution (tests/golden/img0{1,2}_half.pgm, made by make_img_fixture.py).

What is frozen: the keypoints and u8 descriptors host/pm_features.cpp (the C++ front end, run through
`pm_cli --extract-only --save-features`) finds in exactly these pixels, and what the CPU ORACLE makes of them:
2-NN + ratio 0.8 match list, 5 000-hypothesis RANSAC-F (seed 0x5EED, tau 1 px): key, F bits, inlier mask.
The .npz holds data only.  No GPU needed.

    python tests/golden/make_img_cli_fixture.py
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import pm_oracle as O                    # noqa: E402
from points_matching_amd import build, io            # noqa: E402

ITERS, SEED, TAU, RATIO = 5000, 0x5EED, 1.0, 0.8


def main():
    build.build_host()
    with tempfile.TemporaryDirectory() as tmp:
        pre = os.path.join(tmp, "f")
        subprocess.run([build.HOST_BIN, "--img1", os.path.join(HERE, "img01_half.pgm"), "--img2", os.path.join(HERE, "img02_half.pgm"),
                        "--extract-only", "--save-features", pre], check=True)
        d1, d2 = io.load_pmm(pre + "_desc1.pmm"), io.load_pmm(pre + "_desc2.pmm")
        kp1, kp2 = io.load_pmm(pre + "_kp1.pmm"), io.load_pmm(pre + "_kp2.pmm")
    q, t = d1.astype(np.float32), d2.astype(np.float32)
    assert (q == np.rint(q)).all() and q.min() >= 0 and q.max() <= 255
    knn = O.bf_knn_l2(q, t, 2)
    good = O.filter_ratio(knn, RATIO)
    xy1, xy2 = O.gather_points(kp1, good["queryIdx"]), O.gather_points(kp2, good["trainIdx"])
    rc, F, mask, ninl, key = O.ransac_fundamental(xy1, xy2, ITERS, TAU, SEED, nthreads=8)
    assert rc == 0
    np.savez_compressed(os.path.join(HERE, "img_half_cli.npz"), kp1=kp1, kp2=kp2, desc1=d1.astype(np.uint8), desc2=d2.astype(np.uint8),
                        ratio_query=good["queryIdx"], ratio_train=good["trainIdx"], ratio_dist_bits=good["distance"].view(np.uint32),
                        mask=mask, F_bits=F.reshape(9).view(np.uint64), key=np.array([key], np.uint64),
                        params=np.array([ITERS, SEED], np.int64))
    print("%d / %d keypoints, %d ratio matches, %d inliers, hypothesis %d" % (kp1.shape[0], kp2.shape[0], good.size, ninl,
                                                                                 0xFFFFFFFF - (key & 0xFFFFFFFF)))


if __name__ == "__main__":
    main()
