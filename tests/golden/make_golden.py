"""Generates the golden fixtures under tests/golden/ from the CPU oracle.

The reference holds no golden vectors, tests or recorded output for this path, and
its arithmetic lives in an OpenCV build that is not available, so these vectors pin OUR spec
(docs/SPEC.md) — "parity unpinned" with respect to OpenCV itself.  They guard (a) the oracle
against silent drift and (b) the HIP path against the same frozen answers on the GPU box, where
neither /root/reference nor this generator's history is needed.

    python tests/golden/make_golden.py        # rewrites the .npz files
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import pm_oracle as O          # noqa: E402
from points_matching_amd import synth      # noqa: E402


def main():
    q, t, truth = synth.sift_like(256, 256, 128, seed=0x601D)
    m = O.bf_knn_l2(q, t, 2)
    np.savez_compressed(os.path.join(HERE, "knn_l2_sift_256x256x128.npz"), q=q.astype(np.uint8),
                        t=t.astype(np.uint8), idx=m["trainIdx"], dist_bits=m["distance"].view(np.uint32),
                        truth=truth)
    q, t, truth = synth.surf_like(96, 160, 128, seed=0x601E)
    m = O.bf_knn_l2(q, t, 2)
    np.savez_compressed(os.path.join(HERE, "knn_l2_surf_96x160x128.npz"), q=q, t=t, idx=m["trainIdx"],
                        dist_bits=m["distance"].view(np.uint32), truth=truth)
    q, t, truth = synth.surf_like(40, 50, 20, seed=0x601F)          # dim % 8 != 0: scalar tail of S1
    m = O.bf_knn_l2(q, t, 3)
    np.savez_compressed(os.path.join(HERE, "knn_l2_surf_40x50x20_k3.npz"), q=q, t=t, idx=m["trainIdx"],
                        dist_bits=m["distance"].view(np.uint32), truth=truth)
    q, t, truth = synth.orb_like(256, 256, 32, seed=0x6020)
    m = O.bf_knn_hamming(q, t, 2)
    np.savez_compressed(os.path.join(HERE, "knn_hamming_256x256x32.npz"), q=q, t=t, idx=m["trainIdx"],
                        dist=m["distance"], truth=truth)
    for name, out_frac, noise in (("clean", 0.0, 0.0), ("noisy", 0.0, 0.5), ("outliers", 0.35, 0.5)):
        x1, x2, Fgt, inl = synth.two_view(512, seed=0x6021, outlier_frac=out_frac, noise_px=noise)
        res = {}
        for kind in (0, 1):
            rc, F, mask, n, key = O.ransac_fundamental(x1, x2, 500, 1.0, 0x5EED, kind)
            assert rc == 0
            res["F_bits_%d" % kind] = F.reshape(9).view(np.uint64)
            res["mask_%d" % kind] = mask
            res["key_%d" % kind] = np.array([key], np.uint64)
        samples = np.stack([O.sample8(0x5EED, h, 512) for h in range(16)])
        np.savez_compressed(os.path.join(HERE, "twoview_N512_%s.npz" % name), xy1=x1, xy2=x2, F_gt=Fgt,
                            gt_inlier=inl, samples_h0_15=samples, **res)
    lmeds()
    print("golden fixtures written to", HERE)


def lmeds():
    """7-point + LMedS (SPEC S13-S15) answers on the two-view fixtures' own point sets."""
    res = {}
    for name, out_frac, noise in (("clean", 0.0, 0.0), ("noisy", 0.0, 0.5), ("outliers", 0.35, 0.5)):
        x1, x2, Fgt, inl = synth.two_view(512, seed=0x6021, outlier_frac=out_frac, noise_px=noise)
        rc, F, mask, n, best, med = O.lmeds_fundamental(x1, x2, 300, 0x7EED)
        assert rc == 0
        res["F_bits_" + name] = F.reshape(9).view(np.uint64)
        res["mask_" + name] = mask
        res["best_" + name] = np.array([best], np.int64)
        res["median_bits_" + name] = np.array([med], np.float64).view(np.uint64)
    x1, x2, _, _ = synth.two_view(64, seed=0x6022)
    p1 = x1[:7].astype(np.float64)
    p2 = x2[:7].astype(np.float64)
    F7, valid = O.solve7(p1, p2)
    np.savez_compressed(os.path.join(HERE, "lmeds_N512.npz"), solve7_p1=p1, solve7_p2=p2,
                        solve7_F_bits=F7.reshape(27).view(np.uint64), solve7_valid=valid, **res)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "lmeds":
        lmeds()
    else:
        main()
