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


# ---- matrix-core (i8) route: 256-bit descriptors, k <= 2 -------------------------------------
@pytest.mark.parametrize("nq,nt,k", [(700, 1000, 2), (257, 129, 1), (5, 128, 2), (1030, 4100, 2), (64, 2, 2)])
def test_hamming_i8_route_matches_valu_route_and_oracle(ctx, oracle, nq, nt, k):
    q, t, _ = synth.orb_like(nq, nt, 32, seed=7 * nq + nt)
    got = ctx.bf_knn_hamming(q, t, k)
    try:
        ctx.set_option(pm.api.PM_OPT_HAMMING_ROUTE, 1)      # integer-VALU scan
        valu = ctx.bf_knn_hamming(q, t, k)
        ctx.set_option(pm.api.PM_OPT_HAMMING_ROUTE, 2)      # 64-bit keys (what >= 2^23 train rows would take)
        wide = ctx.bf_knn_hamming(q, t, k)
    finally:
        ctx.set_option(pm.api.PM_OPT_HAMMING_ROUTE, 0)
    assert_matches_equal(got, valu, "i8 vs valu")
    assert_matches_equal(got, wide, "32-bit vs 64-bit keys")
    assert_matches_equal(got, oracle.bf_knn_hamming(q, t, k), "i8 vs oracle")


@pytest.mark.parametrize("nq,nt,k", [(700, 1000, 2), (257, 129, 1), (5, 128, 2), (1030, 4100, 2), (64, 2, 2), (33, 9000, 2), (4099, 600, 1),
                                     (2048, 2048, 2)])
def test_hamming_refinement_forms_same_result(ctx, oracle, nq, nt, k):
    """PM_OPT_HAMMING_REFINE: one wave per query (round 1) and four queries per wave on 16-lane rows (round 3), on planted
    data and on tie-heavy alphabets (every group ties, sub-lists overflow): the same records, = the oracle's."""
    rng = np.random.default_rng(nq + nt)
    cases = [synth.orb_like(nq, nt, 32, seed=11 * nq + nt)[:2]]
    base = rng.integers(0, 256, (4, 32), dtype=np.uint8)
    tq, tt = base[rng.integers(0, 4, nq)].copy(), base[rng.integers(0, 4, nt)].copy()
    tt[::5, 0] ^= 1
    cases.append((tq, tt))
    for ci, (q, t) in enumerate(cases):
        want = oracle.bf_knn_hamming(q, t, k, nthreads=8)
        try:
            for form in (1, 2, 0):
                ctx.set_option(pm.api.PM_OPT_HAMMING_REFINE, form)
                for route in (0, 2):                         # 32-bit / 64-bit keys
                    ctx.set_option(pm.api.PM_OPT_HAMMING_ROUTE, route)
                    assert_matches_equal(ctx.bf_knn_hamming(q, t, k), want, "case %d refinement form %d route %d" % (ci, form, route))
        finally:
            ctx.set_option(pm.api.PM_OPT_HAMMING_REFINE, 0)
            ctx.set_option(pm.api.PM_OPT_HAMMING_ROUTE, 0)


def test_hamming_i8_route_is_the_one_timed(ctx):
    q, t, _ = synth.orb_like(512, 512, 32, seed=5)
    ctx.timing_enable(True)
    ctx.timing_reset()
    ctx.bf_knn_hamming(q, t, 2)
    launches = {n: ctx.timing_get(n)[1] for n in ("knn_hamming_mfma_i8", "knn_hamming_refine", "knn_hamming")}
    ctx.timing_enable(False)
    assert launches == {"knn_hamming_mfma_i8": 1, "knn_hamming_refine": 1, "knn_hamming": 0}, launches


def test_hamming_i8_degenerate_ties_force_sublist_scans(ctx, oracle):
    """All-equal and few-valued descriptors: every group ties, every sub-list overflows."""
    rng = np.random.default_rng(3)
    t = np.zeros((600, 32), np.uint8)
    q = np.zeros((70, 32), np.uint8)
    assert_matches_equal(ctx.bf_knn_hamming(q, t, 2), oracle.bf_knn_hamming(q, t, 2), "all zero")
    # three distinct rows repeated: many exact ties at the k-th distance
    base = rng.integers(0, 256, (3, 32), dtype=np.uint8)
    t = base[rng.integers(0, 3, 900)]
    q = base[rng.integers(0, 3, 130)]
    q[::7] ^= 1
    assert_matches_equal(ctx.bf_knn_hamming(q, t, 2), oracle.bf_knn_hamming(q, t, 2), "three values")
    # complemented queries: every distance > 128, pad rows (coarse distance 8256) must still lose
    t = rng.integers(0, 256, (130, 32), dtype=np.uint8)
    q = (~t[:40]).copy()
    assert_matches_equal(ctx.bf_knn_hamming(q, t, 2), oracle.bf_knn_hamming(q, t, 2), "complement")


def test_hamming_i8_cluster_in_one_group_stream(ctx, oracle):
    """More than 4 near rows inside one lane stream (same split, same half): the 4-deep list overflows."""
    rng = np.random.default_rng(11)
    t = rng.integers(0, 256, (2048, 32), dtype=np.uint8)
    q = rng.integers(0, 256, (96, 32), dtype=np.uint8)
    rows = [0, 8, 16, 24, 32, 40, 64, 72]            # rows with (row % 8) < 4: lane half 0, distinct groups
    for i, r in enumerate(rows):
        t[r] = q[0]
        t[r, 31] ^= np.uint8(1 << (i % 8))          # distance 1 each
    t[1000] = q[0]
    assert_matches_equal(ctx.bf_knn_hamming(q, t, 2), oracle.bf_knn_hamming(q, t, 2), "cluster")


def test_hamming_randomised_shapes(ctx, oracle):
    """Seeded sweep over shapes, descriptor sizes and k: both routes must give the oracle's answer."""
    rng = np.random.default_rng(0xBEEF)
    for case in range(30):
        nq = int(rng.integers(1, 900))
        nt = int(rng.integers(1, 1500))
        nbytes = int(rng.choice([4, 16, 32, 32, 32, 64]))
        k = int(rng.choice([1, 2, 2, 3, 5]))
        if case % 3 == 0:                              # few distinct values: ties everywhere
            base = rng.integers(0, 256, (4, nbytes), dtype=np.uint8)
            q = base[rng.integers(0, 4, nq)].copy()
            t = base[rng.integers(0, 4, nt)].copy()
            t[::5, 0] ^= 1
        else:
            q = rng.integers(0, 256, (nq, nbytes), dtype=np.uint8)
            t = rng.integers(0, 256, (nt, nbytes), dtype=np.uint8)
            sel = rng.integers(0, nt, max(1, nq // 2))
            q[:sel.size] = t[sel]
            q[:sel.size, rng.integers(0, nbytes)] ^= np.uint8(rng.integers(0, 256))
        assert_matches_equal(ctx.bf_knn_hamming(q, t, k), oracle.bf_knn_hamming(q, t, k),
                             "case %d: nq=%d nt=%d bytes=%d k=%d" % (case, nq, nt, nbytes, k))


def test_hamming_lds_dma_staging_same_result(ctx, oracle):
    q, t, _ = synth.orb_like(3000, 5000, 32, seed=99)
    base = ctx.bf_knn_hamming(q, t, 2)
    ctx.set_option(pm.api.PM_OPT_KNN_STAGING, 1)
    try:
        assert_matches_equal(ctx.bf_knn_hamming(q, t, 2), base, "register staging vs LDS-DMA (default)")
    finally:
        ctx.set_option(pm.api.PM_OPT_KNN_STAGING, 0)
    assert_matches_equal(base, oracle.bf_knn_hamming(q, t, 2, nthreads=8), "vs oracle")
