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


def _blocks(n1, n_dev):
    cap = (n1 + n_dev - 1) // n_dev
    return [(min(g * cap, n1), min((g + 1) * cap, n1)) for g in range(n_dev)]


@pytest.mark.parametrize("n_dev", _ndevs())
@pytest.mark.parametrize("lanes", [1, 2, 3])
def test_mgpu_streamed_pairs_equal_the_blocking_call(oracle, n_dev, lanes):
    """pm_mgpu_set_train + pm_mgpu_submit_dev / pm_mgpu_collect (device pointers in, tickets out, `lanes` pairs in flight,
    train side resident): every pair must give the bits of pm_mgpu_match_ransac on the same inputs, whatever the lane
    count and the collection order."""
    nq, nt, dim, H = 1200, 1500, 128, 2000
    mg = pm.api.MultiGpu(n_dev)
    big = synth.pair_workload(5 * nq, nt, dim, seed=60, planted=0.4, kind="sift")       # one train image, five query images
    pairs = []
    for i in range(5):
        rows = np.arange(i, 5 * nq, 5)[:nq - 37 * i]          # (every fifth row: each image keeps its share of planted matches)
        pairs.append({"q": np.ascontiguousarray(big["q"][rows]), "kp1": np.ascontiguousarray(big["kp1"][rows]), "t": big["t"], "kp2": big["kp2"]})
    try:
        want = [mg.match_ransac(w["q"], w["t"], w["kp1"], w["kp2"], 0.8, H, 1.0, 0x5EED, knn_flags=pm.api.PM_KNN_HINT_U8) for w in pairs]
        mg.set_lanes(lanes)
        mg.set_train(pairs[0]["t"], pairs[0]["kp2"])
        hold, tickets, got = [], [], {}

        def collect(i):
            got[i] = mg.collect(tickets[i], n1=pairs[i]["q"].shape[0], want_good=True, want_mask=True)
        for ip, w in enumerate(pairs):
            n1 = w["q"].shape[0]
            dq, dk, rows = [], [], []
            for g, (a, b) in enumerate(_blocks(n1, n_dev)):
                dev = torch.device("cuda", g)
                dq.append(torch.from_numpy(np.ascontiguousarray(w["q"][a:b])).to(dev))
                dk.append(torch.from_numpy(np.ascontiguousarray(w["kp1"][a:b])).to(dev))
                rows.append(b - a)
            for g in range(n_dev):
                torch.cuda.synchronize(g)
            hold.append((dq, dk))
            if ip >= lanes:                                   # a lane holds one pair until it is collected: `lanes` pairs in flight
                if ip == lanes:
                    with pytest.raises(pm.PmError):
                        mg.submit_dev([t.data_ptr() for t in dq], rows, [t.data_ptr() for t in dk], 0.8, H, 1.0, 0x5EED)
                collect(ip - lanes)
            tickets.append(mg.submit_dev([t.data_ptr() for t in dq], rows, [t.data_ptr() for t in dk], 0.8, H, 1.0, 0x5EED,
                                         knn_flags=pm.api.PM_KNN_HINT_U8))
        for i in reversed(range(len(pairs))):                 # the rest, youngest first
            if i not in got:
                collect(i)
        for i in range(len(pairs)):
            res, good, mask = got[i]
            rc, g_w, F_w, m_w, ninl_w, key_w = want[i]
            assert res.status == rc == 0 and res.best_key == key_w and res.n_inliers == ninl_w and res.n_good == g_w.size
            assert (np.array(res.F[:]).view(np.uint64) == F_w.reshape(9).view(np.uint64)).all()
            assert (good["queryIdx"] == g_w["queryIdx"]).all() and (good["trainIdx"] == g_w["trainIdx"]).all()
            assert (good["distance"].view(np.uint32) == g_w["distance"].view(np.uint32)).all() and (mask == m_w).all()
        with pytest.raises(pm.PmError):
            mg.collect(tickets[0])                            # a ticket is collected once
        us = mg.allgather_latency(80, reps=50)
        assert us > 0
    finally:
        mg.close()
    # and the blocking call equals the oracle pipeline (first pair)
    w = pairs[0]
    knn = oracle.bf_knn_l2(w["q"], w["t"], 2, nthreads=8)
    g_o = oracle.filter_ratio(knn, 0.8)
    rc_o, F_o, mask_o, ninl_o, key_o = oracle.ransac_fundamental(oracle.gather_points(w["kp1"], g_o["queryIdx"]),
                                                                 oracle.gather_points(w["kp2"], g_o["trainIdx"]), H, 1.0, 0x5EED, nthreads=8)
    assert want[0][5] == key_o and want[0][4] == ninl_o and (want[0][2].view(np.uint64) == F_o.view(np.uint64)).all()


@pytest.mark.parametrize("n_dev", _ndevs())
def test_mgpu_batch_run_equals_the_single_device_batch(n_dev):
    """BASELINE config C5 behind the ABI: pm_mgpu_batch_run (pair p -> device p mod n_dev, one host thread per device)
    returns what pm_batch_run returns, pair for pair."""
    n, dim, H, P = 700, 128, 512, 7
    ws = [synth.pair_workload(n - 11 * i, n, dim, seed=80 + i, planted=0.4, kind="sift") for i in range(P)]
    keep = [{k: np.ascontiguousarray(w[k]) for k in ("q", "t", "kp1", "kp2")} for w in ws]
    jobs = [(k["q"].ctypes.data, k["q"].shape[0], k["t"].ctypes.data, n, k["kp1"].ctypes.data, k["kp2"].ctypes.data) for k in keep]
    b = pm.api.PairBatch(0, 2, n, n, dim)
    want, g_w, m_w = b.run(jobs, 0.8, H, 1.0, 0x5EED, knn_flags=pm.api.PM_KNN_HINT_U8, want_good=True, want_masks=True)
    b.close()
    mg = pm.api.MultiGpu(n_dev)
    try:
        for rep in range(2):                                  # the per-device batch objects persist across calls
            got, g_g, m_g = mg.batch_run(jobs, 2, n, n, dim, 0.8, H, 1.0, 0x5EED, knn_flags=pm.api.PM_KNN_HINT_U8, want_good=True,
                                         want_masks=True)
            for i in range(P):
                assert got[i].status == want[i].status and got[i].best_key == want[i].best_key and got[i].n_good == want[i].n_good
                assert got[i].n_inliers == want[i].n_inliers and list(got[i].F) == list(want[i].F)
                ng = want[i].n_good
                assert (g_g[i, :ng] == g_w[i, :ng]).all() and (m_g[i, :ng] == m_w[i, :ng]).all()
    finally:
        mg.close()
