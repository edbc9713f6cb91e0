"""GPU: the device-resident pipeline (matcher -> ratio filter + gather -> RANSAC with the count
on the device -> finalisation) against the host-side stages and the oracle."""
import numpy as np
import pytest

import points_matching_amd as pm
from points_matching_amd import synth
from util import assert_matches_equal

pytestmark = pytest.mark.gpu


def test_device_pipeline_matches_host_stages(ctx, oracle):
    import torch
    dev = torch.device("cuda", 0)
    w = synth.pair_workload(nq=1500, nt=1300, dim=128, seed=21, planted=0.4)
    nq, nt, K = 1500, 1300, 2
    s = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(s):
        ctx.set_stream(s.cuda_stream)
        d_q = torch.from_numpy(w["q"]).to(dev)
        d_t = torch.from_numpy(w["t"]).to(dev)
        d_kp1 = torch.from_numpy(w["kp1"]).to(dev)
        d_kp2 = torch.from_numpy(w["kp2"]).to(dev)
        d_knn = torch.empty((nq, K, 4), dtype=torch.int32, device=dev)
        d_good = torch.zeros((nq, 4), dtype=torch.int32, device=dev)
        d_xy1 = torch.zeros((nq, 2), dtype=torch.float32, device=dev)
        d_xy2 = torch.zeros((nq, 2), dtype=torch.float32, device=dev)
        d_n = torch.zeros(1, dtype=torch.int32, device=dev)
        d_key = torch.zeros(1, dtype=torch.int64, device=dev)
        d_F = torch.zeros(9, dtype=torch.float64, device=dev)
        d_mask = torch.full((nq,), 7, dtype=torch.uint8, device=dev)
        d_ninl = torch.zeros(1, dtype=torch.int32, device=dev)
        s.synchronize()
        ctx.bf_knn_l2_dev(d_q.data_ptr(), nq, d_t.data_ptr(), nt, 128, K, d_knn.data_ptr())
        ctx.filter_ratio_gather_dev(d_knn.data_ptr(), nq, K, 0.8, d_kp1.data_ptr(), d_kp2.data_ptr(),
                                    d_good.data_ptr(), d_xy1.data_ptr(), d_xy2.data_ptr(), d_n.data_ptr())
        ctx.ransac_score_devn(d_xy1.data_ptr(), d_xy2.data_ptr(), nq, d_n.data_ptr(), 0, 800, 1.0, 11,
                              d_key.data_ptr())
        ctx.ransac_model_from_key_dev(d_xy1.data_ptr(), d_xy2.data_ptr(), nq, d_n.data_ptr(), 1.0, 11,
                                      d_key.data_ptr(), d_F.data_ptr(), d_mask.data_ptr(), d_ninl.data_ptr())
        ctx.synchronize()
        ctx.set_stream(0)
    # the fused single-shard run must publish the same key / F / mask / count
    d_key2 = torch.zeros(1, dtype=torch.int64, device=dev)
    d_F2 = torch.zeros(9, dtype=torch.float64, device=dev)
    d_mask2 = torch.full((nq,), 7, dtype=torch.uint8, device=dev)
    d_ninl2 = torch.full((1,), 99, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    ctx.ransac_run_dev(d_xy1.data_ptr(), d_xy2.data_ptr(), nq, d_n.data_ptr(), 0, 800, 1.0, 11, d_key2.data_ptr(),
                       d_F2.data_ptr(), d_mask2.data_ptr(), d_ninl2.data_ptr())
    ctx.synchronize()
    assert int(d_key2.item()) == int(d_key.item()) and int(d_ninl2.item()) == int(d_ninl.item())
    assert torch.equal(d_F2, d_F) and torch.equal(d_mask2, d_mask)
    knn = d_knn.cpu().numpy().view(pm.MATCH_DTYPE).reshape(nq, K)
    want_knn = oracle.bf_knn_l2(w["q"], w["t"], K)
    assert_matches_equal(knn, want_knn, "dev knn")
    good_o = oracle.filter_ratio(want_knn, 0.8)
    n = int(d_n.item())
    assert n == good_o.size
    assert_matches_equal(d_good.cpu().numpy().view(pm.MATCH_DTYPE).reshape(-1)[:n], good_o, "dev ratio")
    xy1 = oracle.gather_points(w["kp1"], good_o["queryIdx"])
    xy2 = oracle.gather_points(w["kp2"], good_o["trainIdx"])
    assert (d_xy1.cpu().numpy()[:n] == xy1).all() and (d_xy2.cpu().numpy()[:n] == xy2).all()
    rc, F, mask, ninl, key = oracle.ransac_fundamental(xy1, xy2, 800, 1.0, 11)
    assert int(d_key.item()) == key and int(d_ninl.item()) == ninl
    assert (d_mask.cpu().numpy()[:n] == mask).all() and not d_mask.cpu().numpy()[n:].any()
    assert (d_F.cpu().numpy().view(np.uint64) == F.reshape(9).view(np.uint64)).all()


def test_concat_points(ctx):
    import torch
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(0)
    parts, stride = 3, 100
    a = rng.random((parts, stride, 2)).astype(np.float32)
    b = rng.random((parts, stride, 2)).astype(np.float32)
    counts = np.array([40, 0, 100], np.int32)
    d_a, d_b, d_c = (torch.from_numpy(x).to(dev) for x in (a, b, counts))
    o1 = torch.zeros((parts * stride, 2), dtype=torch.float32, device=dev)
    o2 = torch.zeros_like(o1)
    d_n = torch.zeros(1, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    ctx.concat_points_dev(d_a.data_ptr(), d_b.data_ptr(), d_c.data_ptr(), parts, stride, o1.data_ptr(),
                          o2.data_ptr(), d_n.data_ptr())
    ctx.synchronize()
    assert int(d_n.item()) == 140
    want1 = np.concatenate([a[p, :counts[p]] for p in range(parts)])
    want2 = np.concatenate([b[p, :counts[p]] for p in range(parts)])
    assert (o1.cpu().numpy()[:140] == want1).all() and (o2.cpu().numpy()[:140] == want2).all()


def test_knn_stats_report_refinement(ctx):
    q, t, _ = synth.surf_like(1024, 1024, 128, seed=2)
    ctx.knn_diag_enable(True)
    ctx.bf_knn_l2(q, t, 2)
    st = ctx.knn_stats()
    ctx.knn_diag_enable(False)
    assert st["nonfinite"] == 0 and st["rescans"] < 32      # the MFMA route, not the re-scan, did the work


@pytest.mark.parametrize("n,scale", [(1, 0.3), (255, 0.5), (256, 300.0), (257, 0.9), (5000, 0.7), (4097, 250.0)])
def test_device_midpoint_filter_matches_the_reference_rule(ctx, n, scale):
    """pm_filter_midpoint_gather_dev == the host restatement of main.cpp:49-69 (incl. the min = 1,
    max = 0 start values: with SIFT-scale distances, all > 1, minMatch stays 1) + main.cpp:89-91."""
    import torch
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(n)
    m = np.zeros(n, pm.MATCH_DTYPE)
    m["queryIdx"] = np.arange(n)
    m["trainIdx"] = rng.integers(0, 777, n)
    m["distance"] = (rng.random(n) * scale + (2.0 if scale > 1 else 0.0)).astype(np.float32)
    if n > 300:
        m["distance"][7] = np.nan                          # a NaN never wins a comparison (main.cpp:54-55, :65)
        m["distance"][100:110] = m["distance"][100]        # ties
    kp1 = rng.random((n, 2)).astype(np.float32) * 900
    kp2 = rng.random((777, 2)).astype(np.float32) * 600
    good_h, lo, hi = pm.api.filter_midpoint(m)
    d_m = torch.from_numpy(m.view(np.int32).reshape(n, 4)).to(dev)
    d_kp1, d_kp2 = torch.from_numpy(kp1).to(dev), torch.from_numpy(kp2).to(dev)
    d_good = torch.zeros((n, 4), dtype=torch.int32, device=dev)
    d_xy1 = torch.zeros((n, 2), dtype=torch.float32, device=dev)
    d_xy2 = torch.zeros((n, 2), dtype=torch.float32, device=dev)
    d_n = torch.zeros(1, dtype=torch.int32, device=dev)
    d_mm = torch.zeros(2, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    for rep in range(2):                                   # the second call reuses the epoch-tagged words
        ctx.filter_midpoint_gather_dev(d_m.data_ptr(), n, 1, d_kp1.data_ptr(), d_kp2.data_ptr(), d_good.data_ptr(),
                                       d_xy1.data_ptr(), d_xy2.data_ptr(), d_n.data_ptr(), d_mm.data_ptr())
        ctx.synchronize()
        k = int(d_n.item())
        assert k == good_h.size
        got = d_good.cpu().numpy().view(pm.MATCH_DTYPE).reshape(-1)[:k]
        assert (got["queryIdx"] == good_h["queryIdx"]).all() and (got["trainIdx"] == good_h["trainIdx"]).all()
        assert (got["distance"].view(np.uint32) == good_h["distance"].view(np.uint32)).all()
        assert d_mm.cpu().numpy().tolist() == [lo, hi]
        assert np.array_equal(d_xy1.cpu().numpy()[:k], kp1[good_h["queryIdx"]])
        assert np.array_equal(d_xy2.cpu().numpy()[:k], kp2[good_h["trainIdx"]])


# ---- matcher + ratio filter + gather in one call (pm_bf_knn_l2_ratio_dev): the filter rides the refinement launch ----
@pytest.mark.parametrize("nq,nt,dim,kind", [(8192, 8192, 128, "sift"), (1000, 777, 128, "sift"), (33, 500, 64, "surf"),
                                            (4, 9, 128, "sift"), (2049, 300, 128, "surf"), (300, 1, 128, "sift"),
                                            (700, 650, 30, "surf")])
def test_fused_matcher_filter_equals_the_two_call_form(ctx, oracle, nq, nt, dim, kind):
    import torch
    dev = torch.device("cuda", 0)
    w = synth.pair_workload(nq, nt, dim, seed=nq + nt, planted=0.4, kind=kind)
    flags = (pm.api.PM_KNN_HINT_U8 if (nq + nt) % 2 else pm.api.PM_KNN_HINT_INTEGER) if kind == "sift" else 0
    d_q, d_t = torch.from_numpy(w["q"]).to(dev), torch.from_numpy(w["t"]).to(dev)
    d_kp1, d_kp2 = torch.from_numpy(w["kp1"]).to(dev), torch.from_numpy(w["kp2"]).to(dev)
    want = oracle.filter_ratio(oracle.bf_knn_l2(w["q"], w["t"], 2, nthreads=8), 0.8)
    for mode, with_knn in ((0, True), (0, False), (1, True), (2, True)):
        if dim % 4 and not with_knn:
            continue                                   # the exact kernel needs the record buffer
        d_knn = torch.zeros((nq, 2, 4), dtype=torch.int32, device=dev)
        d_good = torch.full((nq, 4), -7, dtype=torch.int32, device=dev)
        d_xy1 = torch.full((nq, 2), -1.0, dtype=torch.float32, device=dev)
        d_xy2 = torch.full((nq, 2), -1.0, dtype=torch.float32, device=dev)
        d_n = torch.full((1,), -1, dtype=torch.int32, device=dev)
        ctx.set_option(pm.api.PM_OPT_FILTER_FUSION, mode)
        try:
            for _ in range(2):                         # twice: arrival words must be back at zero, epochs move on
                ctx.bf_knn_l2_ratio_dev(d_q.data_ptr(), nq, d_t.data_ptr(), nt, dim, flags, 0.8, d_kp1.data_ptr(), d_kp2.data_ptr(),
                                        d_knn.data_ptr() if with_knn else 0, d_good.data_ptr(), d_xy1.data_ptr(),
                                        d_xy2.data_ptr(), d_n.data_ptr())
            ctx.synchronize()
        finally:
            ctx.set_option(pm.api.PM_OPT_FILTER_FUSION, 0)
        n = int(d_n.item())
        assert n == want.size, (mode, with_knn)
        assert ctx.filter_fusion_gave_up() == 0             # no look-back poll of the fused compaction ran out
        got = d_good.cpu().numpy().view(pm.MATCH_DTYPE).reshape(-1)[:n]
        assert_matches_equal(got, want, "fused good list %s" % ((mode, with_knn),))
        assert (d_xy1.cpu().numpy()[:n] == w["kp1"][want["queryIdx"]]).all()
        assert (d_xy2.cpu().numpy()[:n] == w["kp2"][want["trainIdx"]]).all()
        if with_knn:
            knn = d_knn.cpu().numpy().view(pm.MATCH_DTYPE).reshape(nq, 2)
            assert_matches_equal(knn, oracle.bf_knn_l2(w["q"], w["t"], 2, nthreads=8), "records")


@pytest.mark.filterwarnings("ignore:The CUDA Graph is empty")
def test_matcher_and_filter_refuse_a_capturing_stream():
    """The matcher and the compaction tag their side-band words with a per-call epoch the host increments; a captured graph
    would freeze it and a replay could read the previous replay's look-back counts.  So those calls fail loudly on a
    capturing stream (PM_E_UNSUPPORTED) and the context keeps working afterwards; the RANSAC call, which carries no such
    argument, records and replays (hipGraph note)."""
    import gc
    import torch
    n, H = 700, 600
    dev = torch.device("cuda", 0)
    st = torch.cuda.Stream(device=dev)
    prev = torch.cuda.current_stream(dev)
    torch.cuda.set_stream(st)
    c = pm.Context(0)
    c.set_stream(st.cuda_stream)
    try:
        w = synth.pair_workload(n, n, 128, seed=11, kind="sift")
        d_q, d_t, d_kp1, d_kp2 = [torch.from_numpy(np.ascontiguousarray(w[k])).to(dev) for k in ("q", "t", "kp1", "kp2")]
        knn = torch.empty((n, 2, 4), dtype=torch.int32, device=dev)
        good = torch.empty((n, 4), dtype=torch.int32, device=dev)
        cnt = torch.zeros(4, dtype=torch.int32, device=dev)
        xy1 = torch.empty((n, 2), dtype=torch.float32, device=dev)
        xy2 = torch.empty((n, 2), dtype=torch.float32, device=dev)
        key = torch.zeros(1, dtype=torch.int64, device=dev)
        F = torch.zeros(9, dtype=torch.float64, device=dev)
        mask = torch.zeros(n, dtype=torch.uint8, device=dev)
        ninl = torch.zeros(1, dtype=torch.int32, device=dev)

        def match(fusion):
            c.set_option(pm.api.PM_OPT_FILTER_FUSION, fusion)
            c.bf_knn_l2_ratio_dev(d_q.data_ptr(), n, d_t.data_ptr(), n, 128, pm.api.PM_KNN_HINT_U8, 0.8, d_kp1.data_ptr(),
                                  d_kp2.data_ptr(), knn.data_ptr(), good.data_ptr(), xy1.data_ptr(), xy2.data_ptr(), cnt.data_ptr())

        def ransac():
            c.ransac_run_dev(xy1.data_ptr(), xy2.data_ptr(), n, cnt.data_ptr(), 0, H, 1.0, 7, key.data_ptr(), F.data_ptr(),
                             mask.data_ptr(), ninl.data_ptr())

        def result():
            torch.cuda.synchronize()
            m = int(cnt[0])
            return (m, int(key[0]), int(ninl[0]), F.cpu().numpy().tobytes(), good.cpu().numpy()[:m].tobytes(),
                    mask.cpu().numpy()[:m].tobytes())

        match(0)
        ransac()
        base = result()
        assert base[0] > 30
        for fusion in (1, 2):
            gc.collect()                 # no finaliser of an earlier test's context (hipFree) inside the capture
            g = torch.cuda.CUDAGraph()
            with pytest.raises(pm.api.PmError) as err:
                with torch.cuda.graph(g, stream=st, capture_error_mode="relaxed"):
                    match(fusion)
            assert err.value.status == pm.api.PM_E_UNSUPPORTED and "capturing" in str(err.value)
            del g
            torch.cuda.set_stream(st)
        match(0)
        key.zero_(); F.zero_()
        gc.collect()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st, capture_error_mode="relaxed"):
            ransac()
        torch.cuda.set_stream(st)
        g.replay()
        assert result() == base
        del g
    finally:
        torch.cuda.synchronize()
        torch.cuda.set_stream(prev)
        c.close()
