"""GPU parity: pm_bf_knn_hamming_u8 vs the CPU oracle (bit-exact; docs/SPEC.md S2/S3)."""
import numpy as np
import pytest

import points_matching_amd as pm
from points_matching_amd import synth
from util import assert_matches_equal

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("nq,nt,nbytes,k", [(256, 256, 32, 1), (256, 256, 32, 2), (100, 77, 32, 2),
                                            (1, 500, 32, 2), (300, 1, 32, 2), (65, 130, 16, 2),
                                            (64, 200, 64, 3), (40, 90, 12, 2), (33, 47, 32, 16),
                                            (20, 400, 32, 7), (10, 3, 32, 5)])
def test_hamming_shapes(ctx, oracle, nq, nt, nbytes, k):
    q, t, _ = synth.orb_like(nq, nt, nbytes, seed=nq + 31 * nt + k)
    assert_matches_equal(ctx.bf_knn_hamming(q, t, k), oracle.bf_knn_hamming(q, t, k),
                         str((nq, nt, nbytes, k)))


def test_hamming_ties_lowest_index(ctx, oracle):
    q, t, _ = synth.orb_like(32, 512, 32, seed=8)
    t[400] = t[3]
    t[100] = t[3]
    q[0] = t[3]
    want = oracle.bf_knn_hamming(q, t, 3)
    assert list(want["trainIdx"][0]) == [3, 100, 400]
    assert_matches_equal(ctx.bf_knn_hamming(q, t, 3), want, "ties")


def test_hamming_empty_and_invalid(ctx):
    t = np.zeros((4, 32), np.uint8)
    assert ctx.bf_knn_hamming(np.zeros((0, 32), np.uint8), t, 2).shape == (0, 2)
    out = ctx.bf_knn_hamming(t, np.zeros((0, 32), np.uint8), 2)
    assert (out["trainIdx"] == -1).all() and np.isinf(out["distance"]).all()
    with pytest.raises(pm.PmError):
        ctx.bf_knn_hamming(np.zeros((4, 30), np.uint8), np.zeros((4, 30), np.uint8), 1)


def test_hamming_c4_slice_and_full_properties(ctx, oracle):
    """BASELINE config C4 matcher: 32k x 32k ORB-256, k=2, ratio 0.8 on integer distances."""
    q, t, truth = synth.orb_like(32768, 32768, 32, seed=0xC4)
    got = ctx.bf_knn_hamming(q, t, 2)
    # oracle on a 2k-query slice against all 32k train rows
    sl = slice(1000, 3048)
    assert_matches_equal(got[sl], _requery(oracle.bf_knn_hamming(q[sl], t, 2, nthreads=8), 1000), "C4")
    planted = truth >= 0
    assert (got["trainIdx"][planted, 0] == truth[planted]).mean() > 0.999
    assert (got["distance"][:, 0] <= got["distance"][:, 1]).all()
    # reported distance is the Hamming distance of the reported pair (checksum over all rows)
    x = np.bitwise_xor(q, t[got["trainIdx"][:, 0]])
    pop = np.unpackbits(x, axis=1).sum(axis=1)
    assert (pop == got["distance"][:, 0]).all()
    good = pm.api.filter_ratio(got, 0.8)
    assert planted.sum() * 0.95 < good.size < planted.sum() * 1.05


def _requery(m, off):
    m = m.copy()
    m["queryIdx"] += off
    return m
