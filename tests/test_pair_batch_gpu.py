"""GPU: pm_batch_run (BASELINE config C5: a batch of independent image pairs streamed over several
lanes) against the CPU oracle run pair by pair: matches, F bits, inlier masks and keys identical,
whatever the number of lanes."""
import numpy as np
import pytest

import points_matching_amd as pm
from points_matching_amd import synth

pytestmark = pytest.mark.gpu

SIZES = [(700, 640), (512, 900), (300, 300), (1024, 1024), (650, 40), (900, 777), (64, 2000)]
H, TAU, SEED, RATIO = 600, 1.0, 0xC5, 0.8


def _pairs():
    out = []
    for i, (n1, n2) in enumerate(SIZES):
        w = synth.pair_workload(nq=n1, nt=n2, dim=128, seed=0xC5 + i, planted=0.5)
        out.append(w)
    return out


def _oracle_pair(oracle, w):
    knn = oracle.bf_knn_l2(w["q"], w["t"], 2)
    good = oracle.filter_ratio(knn, RATIO)
    xy1 = w["kp1"][good["queryIdx"]]
    xy2 = w["kp2"][good["trainIdx"]]
    rc, F, mask, ninl, key = oracle.ransac_fundamental(xy1, xy2, H, TAU, SEED)
    return good, rc, F, mask, ninl, key


@pytest.mark.parametrize("lanes", [1, 3])
def test_batch_matches_oracle_pair_by_pair(oracle, lanes):
    ws = _pairs()
    max1 = max(s[0] for s in SIZES)
    max2 = max(s[1] for s in SIZES)
    b = pm.api.PairBatch(0, lanes, max1, max2, 128)
    jobs = [(w["q"].ctypes.data, w["q"].shape[0], w["t"].ctypes.data, w["t"].shape[0],
             w["kp1"].ctypes.data, w["kp2"].ctypes.data) for w in ws]
    for rep in range(2):                       # the second run reuses every lane's buffers
        res, good, masks = b.run(jobs, RATIO, H, TAU, SEED, want_good=True, want_masks=True)
        for j, w in enumerate(ws):
            g_o, rc_o, F_o, mask_o, ninl_o, key_o = _oracle_pair(oracle, w)
            r = res[j]
            assert r.n_good == g_o.size, (j, r.n_good, g_o.size)
            got = good[j, :r.n_good]
            assert (got["queryIdx"] == g_o["queryIdx"]).all() and (got["trainIdx"] == g_o["trainIdx"]).all()
            assert (got["distance"].view(np.uint32) == g_o["distance"].view(np.uint32)).all()
            if g_o.size < 8:
                assert r.status == pm.api.PM_E_TOO_FEW and r.best_key == 0
                continue
            assert r.status == rc_o == pm.api.PM_OK
            assert r.best_key == key_o and r.n_inliers == ninl_o, (j, r.best_key, key_o)
            assert (np.array(r.F[:]).view(np.uint64) == F_o.reshape(9).view(np.uint64)).all()
            assert (masks[j, :r.n_good] == mask_o).all() and not masks[j, r.n_good:w["q"].shape[0]].any()
    b.close()


def test_batch_two_host_threads_same_results_in_job_order():
    """>= 4 lanes and >= 16 jobs: two host threads enqueue (each its own half of the lanes, every second job).  Results in job
    order, identical to the one-thread, one-lane run."""
    ws = _pairs() * 3                                            # 21 jobs
    max1, max2 = max(s[0] for s in SIZES), max(s[1] for s in SIZES)
    jobs = [(w["q"].ctypes.data, w["q"].shape[0], w["t"].ctypes.data, w["t"].shape[0], w["kp1"].ctypes.data, w["kp2"].ctypes.data) for w in ws]
    b1 = pm.api.PairBatch(0, 1, max1, max2, 128)
    want, g_w, m_w = b1.run(jobs, RATIO, H, TAU, SEED, want_good=True, want_masks=True)
    b1.close()
    for lanes, threads in ((5, 0), (4, 2), (6, 1)):
        b = pm.api.PairBatch(0, lanes, max1, max2, 128)
        b.set_host_threads(threads)
        for rep in range(2):
            got, g_g, m_g = b.run(jobs, RATIO, H, TAU, SEED, want_good=True, want_masks=True)
            for j in range(len(ws)):
                assert (got[j].status, got[j].best_key, got[j].n_good, got[j].n_inliers, bytes(got[j].F)) == \
                       (want[j].status, want[j].best_key, want[j].n_good, want[j].n_inliers, bytes(want[j].F)), (lanes, threads, rep, j)
                ng = want[j].n_good
                assert (g_g[j, :ng] == g_w[j, :ng]).all() and (m_g[j] == m_w[j]).all(), (lanes, threads, rep, j)
        b.close()


def test_batch_pinned_inputs_and_argument_checks():
    w = synth.pair_workload(nq=256, nt=256, dim=128, seed=3, planted=0.5)
    bufs = [w["q"], w["t"], w["kp1"], w["kp2"]]
    for a in bufs:
        pm.api.host_register(a)
    b = pm.api.PairBatch(0, 2, 256, 256, 128)
    job = (w["q"].ctypes.data, 256, w["t"].ctypes.data, 256, w["kp1"].ctypes.data, w["kp2"].ctypes.data)
    res, _, _ = b.run([job] * 5, RATIO, 200, TAU, 1)
    assert len({(r.best_key, r.n_good, r.n_inliers, bytes(r.F)) for r in res}) == 1      # same pair, same answer
    with pytest.raises(pm.PmError):
        b.run([(job[0], 257, job[2], 256, job[4], job[5])], RATIO, 200, TAU, 1)           # exceeds max_n1
    with pytest.raises(pm.PmError):
        b.run([(0, 256, job[2], 256, job[4], job[5])], RATIO, 200, TAU, 1)                # null descriptors
    res, _, _ = b.run([], RATIO, 200, TAU, 1)
    assert len(res) == 0
    b.close()
    for a in bufs:
        pm.api.host_unregister(a)


def test_batch_at_config_c5_size(oracle):
    """BASELINE config C5 at full size: 256 pairs x (4096 x 4096 SIFT-128, 2048 hypotheses), three lanes; eight distinct
    pairs cycled, a sample of the results checked bit for bit against the oracle and every repeat against its twin."""
    n, Hc, distinct, pairs = 4096, 2048, 8, 256
    ws = [synth.pair_workload(n, n, 128, seed=0xC5 + i, kind="sift") for i in range(distinct)]
    b = pm.api.PairBatch(0, 3, n, n, 128)
    jobs = [(ws[j % distinct]["q"].ctypes.data, n, ws[j % distinct]["t"].ctypes.data, n,
             ws[j % distinct]["kp1"].ctypes.data, ws[j % distinct]["kp2"].ctypes.data) for j in range(pairs)]
    res, good, masks = b.run(jobs, RATIO, Hc, TAU, 0x5EED, knn_flags=pm.api.PM_KNN_HINT_INTEGER, want_good=True, want_masks=True)
    b.close()
    for j in range(distinct, pairs):            # same inputs => same answer, whatever lane and position
        a, c = res[j], res[j % distinct]
        assert (a.best_key, a.n_good, a.n_inliers, bytes(a.F)) == (c.best_key, c.n_good, c.n_inliers, bytes(c.F)), j
    for j in (0, 5, 131, 255):
        w = ws[j % distinct]
        g_o = oracle.filter_ratio(oracle.bf_knn_l2(w["q"], w["t"], 2, nthreads=8), RATIO)
        rc_o, F_o, mask_o, ninl_o, key_o = oracle.ransac_fundamental(w["kp1"][g_o["queryIdx"]], w["kp2"][g_o["trainIdx"]],
                                                                     Hc, TAU, 0x5EED, nthreads=8)
        r = res[j]
        assert r.n_good == g_o.size and r.status == rc_o == 0 and r.best_key == key_o and r.n_inliers == ninl_o, j
        assert (good[j, :r.n_good]["trainIdx"] == g_o["trainIdx"]).all() and (good[j, :r.n_good]["queryIdx"] == g_o["queryIdx"]).all()
        assert (np.array(r.F[:]).view(np.uint64) == F_o.reshape(9).view(np.uint64)).all(), j
        assert (masks[j, :r.n_good] == mask_o).all(), j


def _block_of(w, desc_dtype):
    """One host allocation desc1 | desc2 | kp1 | kp2 (256-byte aligned sections) and the job tuple pointing into it."""
    parts = [np.ascontiguousarray(w["q"].astype(desc_dtype)), np.ascontiguousarray(w["t"].astype(desc_dtype)),
             np.ascontiguousarray(w["kp1"], np.float32), np.ascontiguousarray(w["kp2"], np.float32)]
    offs, total = [], 0
    for p in parts:
        total = (total + 255) // 256 * 256
        offs.append(total)
        total += p.nbytes
    blk = np.zeros(total + 512, np.uint8)
    base = (-blk.ctypes.data) % 256                       # 256-byte aligned start inside the allocation
    for p, o in zip(parts, offs):
        blk[base + o:base + o + p.nbytes] = p.view(np.uint8).reshape(-1)
    addr = blk.ctypes.data + base
    return blk, (addr + offs[0], w["q"].shape[0], addr + offs[1], w["t"].shape[0], addr + offs[2], addr + offs[3])


@pytest.mark.parametrize("desc", ["f32", "u8"])
def test_batch_one_block_jobs_and_u8_rows_equal_the_f32_batch(oracle, desc):
    """pm_batch_set_desc_type(1): uint8 descriptor rows; one-block jobs: the four arrays of a pair in one allocation travel in
    one copy.  Same results as the float32 four-copy batch, pair for pair and bit for bit."""
    ws = _pairs()
    max1, max2 = max(s[0] for s in SIZES), max(s[1] for s in SIZES)
    b = pm.api.PairBatch(0, 2, max1, max2, 128)
    jobs = [(w["q"].ctypes.data, w["q"].shape[0], w["t"].ctypes.data, w["t"].shape[0], w["kp1"].ctypes.data, w["kp2"].ctypes.data) for w in ws]
    want, g_w, m_w = b.run(jobs, RATIO, H, TAU, SEED, knn_flags=pm.api.PM_KNN_HINT_U8, want_good=True, want_masks=True)
    held = [_block_of(w, np.float32 if desc == "f32" else np.uint8) for w in ws]
    b.set_desc_u8(desc == "u8")
    got, g_g, m_g = b.run([h[1] for h in held], RATIO, H, TAU, SEED, knn_flags=pm.api.PM_KNN_HINT_U8, want_good=True, want_masks=True)
    b.close()
    for j in range(len(ws)):
        assert (got[j].status, got[j].best_key, got[j].n_good, got[j].n_inliers, bytes(got[j].F)) == \
               (want[j].status, want[j].best_key, want[j].n_good, want[j].n_inliers, bytes(want[j].F)), (desc, j)
        ng = want[j].n_good
        assert (g_g[j, :ng] == g_w[j, :ng]).all() and (m_g[j] == m_w[j]).all()


@pytest.mark.parametrize("nq,nt,dim,k", [(700, 900, 128, 2), (1, 300, 128, 1), (300, 1, 128, 2), (129, 130, 32, 2), (50, 70, 12, 2),
                                         (2048, 2048, 128, 2), (40, 90, 7, 2), (64, 100, 128, 5), (20, 200, 200, 3), (513, 2049, 64, 1)])
def test_knn_l2_u8_rows_equal_the_f32_matcher(ctx, oracle, nq, nt, dim, k):
    """pm_bf_knn_l2_u8: uint8 rows in, the records of pm_bf_knn_l2_f32 on the same values out — the u8 route where it applies
    (dim % 4 == 0, dim <= 128, k <= 2), widened copies through the f32 matcher elsewhere."""
    rng = np.random.default_rng(nq * 31 + nt)
    if dim == 128:
        qf, tf, _ = synth.sift_like(nq, nt, dim, seed=nq + nt)
        q, t = qf.astype(np.uint8), tf.astype(np.uint8)
    else:
        q = rng.integers(0, 256, (nq, dim), dtype=np.uint8)
        t = rng.integers(0, 256, (nt, dim), dtype=np.uint8)
    want = oracle.bf_knn_l2(q.astype(np.float32), t.astype(np.float32), k, nthreads=8)
    from util import assert_matches_equal
    assert_matches_equal(ctx.bf_knn_l2_u8(q, t, k), want, "u8 rows %s" % ((nq, nt, dim, k),))
    # every option that sends a shape to the exact kernel must first widen u8 rows (found by tools/fuzz_campaign.py: k = 3 with
    # PM_OPT_KNN_WIDE = 1 reached the f32 kernel with the null f32 pointers of the u8 entry point)
    try:
        ctx.set_option(pm.api.PM_OPT_KNN_WIDE, 1)
        assert_matches_equal(ctx.bf_knn_l2_u8(q, t, k), want, "u8 rows, exact kernel for k > 2 %s" % ((nq, nt, dim, k),))
    finally:
        ctx.set_option(pm.api.PM_OPT_KNN_WIDE, 0)
