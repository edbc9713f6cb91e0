"""GPU parity of the round-3 coarse routes of pm_bf_knn_l2_f32 (main.cpp:46, BF-L2 matcher of main.cpp:43):

  PM_KNN_HINT_U8       u8-valued descriptors (OpenCV SIFT) centred to x - 128 and ranked on the i8 matrix cores, the
                       per-row term starting the accumulators from a per-tile LDS array (route 3);
  PM_KNN_HINT_INTEGER  the f16 pass in its seeded form (8 k-chunks, PM_OPT_KNN_SEEDED = 2) against the default form with
                       the row term as a ninth k-chunk (PM_OPT_KNN_SEEDED = 0 / 1).

Every route must return the oracle's bits (docs/SPEC.md S1/S3); a wrong hint may only cost time."""
import numpy as np
import pytest

import points_matching_amd as pm
from points_matching_amd import synth
from points_matching_amd.api import (PM_KNN_HINT_INTEGER, PM_KNN_HINT_U8, PM_OPT_KNN_F16_WAVES, PM_OPT_KNN_RING, PM_OPT_KNN_RING_PROLOGUE,
                                      PM_OPT_KNN_SEEDED,                                      PM_OPT_KNN_U8_GROUP, PM_OPT_KNN_U8_REFINE)
from util import assert_matches_equal

pytestmark = pytest.mark.gpu


def _run(ctx, q, t, k, flags):
    ctx.knn_diag_enable(True)
    try:
        got = ctx.bf_knn_l2(q, t, k, flags)
        st = ctx.knn_stats()
    finally:
        ctx.knn_diag_enable(False)
    return got, st


@pytest.mark.parametrize("nq,nt,dim", [(256, 256, 128), (100, 77, 128), (1, 300, 128), (300, 1, 128), (65, 64, 64),
                                       (129, 130, 32), (33, 200, 96), (50, 70, 12), (17, 5, 8), (700, 1000, 128),
                                       (513, 2049, 128), (2048, 2048, 128)])
@pytest.mark.parametrize("k", [1, 2])
def test_u8_route_matches_oracle(ctx, oracle, nq, nt, dim, k):
    q, t, _ = synth.sift_like(nq, nt, dim, seed=nq * 7 + nt)
    want = oracle.bf_knn_l2(q, t, k, nthreads=8)
    got, st = _run(ctx, q, t, k, PM_KNN_HINT_U8)
    assert_matches_equal(got, want, "u8 route %s" % ((nq, nt, dim, k),))
    assert st["route"] == 3 and st["nonfinite"] == 0 and (st["rescans"] <= 1 or nt < 8), st


def test_u8_route_full_byte_range_and_extremes(ctx, oracle):
    """Uniform bytes over the whole 0..255 range (norms far apart), all-zero and all-255 rows (the extremes of the
    centred dot product and of the seed), duplicated rows (exact ties, lowest index first)."""
    rng = np.random.default_rng(8)
    q = rng.integers(0, 256, (300, 128)).astype(np.float32)
    t = rng.integers(0, 256, (1500, 128)).astype(np.float32)
    t[3] = 0.0; t[4] = 255.0; t[5] = 255.0; t[900] = 0.0
    q[0] = 0.0; q[1] = 255.0; q[2] = t[77]; t[78] = t[77]; t[1400] = t[77]
    t[128:141] = t[20]; q[6] = t[20]                      # a run longer than a candidate list: split re-scan
    for k in (1, 2):
        want = oracle.bf_knn_l2(q, t, k, nthreads=8)
        got, st = _run(ctx, q, t, k, PM_KNN_HINT_U8)
        assert_matches_equal(got, want, "full range k=%d" % k)
        assert st["route"] == 3
    assert want["trainIdx"][2, 0] == 77 and want["trainIdx"][2, 1] == 78


def test_u8_route_odd_norms_half_unit(ctx, oracle):
    """The seed is -(||t'||^2 >> 1): rows whose centred norm is odd rank half a unit high.  Neighbours whose squared
    distances differ by exactly 1 must still come out in canonical order."""
    rng = np.random.default_rng(19)
    t = np.full((600, 128), 128.0, np.float32)
    base = rng.integers(100, 156, 128).astype(np.float32)
    q = np.tile(base, (64, 1))
    for j in range(600):                                   # train row j: base with (j % 7) coordinates moved by one
        t[j] = base
        idx = rng.choice(128, size=j % 7, replace=False)
        t[j, idx] += 1.0
    want = oracle.bf_knn_l2(q, t, 2, nthreads=8)
    got, st = _run(ctx, q, t, 2, PM_KNN_HINT_U8)
    assert_matches_equal(got, want, "unit-spaced squared distances")
    assert st["route"] == 3


def test_u8_hint_wrong_only_costs_time(ctx, oracle):
    """Values outside [0, 255], fractions, negatives, non-finite values: the premise is verified on the device and the
    refinement scans exactly."""
    q, t, _ = synth.sift_like(200, 700, 128, seed=4)
    cases = []
    tt = t.copy(); tt[17, 5] = 256.0; cases.append(("256", q, tt))
    tt = t.copy(); tt[17, 5] = 0.5; cases.append(("fraction", q, tt))
    qq = q.copy(); qq[3, 9] = -1.0; cases.append(("negative query", qq, t))
    qs, ts, _ = synth.surf_like(200, 700, 128, seed=5); cases.append(("surf", qs, ts))
    for name, a, b in cases:
        got, st = _run(ctx, a, b, 2, PM_KNN_HINT_U8)
        assert_matches_equal(got, oracle.bf_knn_l2(a, b, 2, nthreads=8), name)
        assert st["nonfinite"] == 1, (name, st)             # the exact-scan branch
    tt = t.copy(); tt[9, 0] = np.inf; tt[7, 3] = np.nan
    got = ctx.bf_knn_l2(q, tt, 2, PM_KNN_HINT_U8)
    want = oracle.bf_knn_l2(q, tt, 2, nthreads=8)
    assert (got["trainIdx"] == want["trainIdx"]).all()
    fin = np.isfinite(want["distance"])
    assert (got["distance"][fin].view(np.uint32) == want["distance"][fin].view(np.uint32)).all()


@pytest.mark.parametrize("nq,nt", [(700, 1000), (8192, 8192), (300, 20000)])
def test_seeded_forms_and_wave_layouts_same_result(ctx, oracle, nq, nt):
    """Hint routes x (seeded, seed chunk) x wave layouts (8 x 32 queries, 4 x 64, two row groups): same candidates,
    same result, no silent repair by the refinement."""
    w = synth.pair_workload(nq, nt, 128, seed=nq ^ nt, planted=0.4, kind="sift")
    base, st0 = _run(ctx, w["q"], w["t"], 2, PM_KNN_HINT_INTEGER)
    try:
        ctx.set_option(PM_OPT_KNN_RING, 1)                   # the double-buffered form shares the f16 kernel's wave layouts
        for seeded in (1, 2):
            ctx.set_option(PM_OPT_KNN_SEEDED, seeded)
            for waves in (0, 1, 2, 3):
                ctx.set_option(PM_OPT_KNN_F16_WAVES, waves)
                for flags in (PM_KNN_HINT_INTEGER, PM_KNN_HINT_U8):
                    got, st = _run(ctx, w["q"], w["t"], 2, flags)
                    assert_matches_equal(got, base, "seeded %d waves %d flags %d" % (seeded, waves, flags))
                    assert st["rescans"] <= max(1, st0["rescans"]) and st["nonfinite"] == 0, (seeded, waves, flags, st)
                    assert st["route"] == (3 if flags == PM_KNN_HINT_U8 and seeded != 1 else 0), st
        ctx.set_option(PM_OPT_KNN_SEEDED, 0)
        ctx.set_option(PM_OPT_KNN_F16_WAVES, 0)
        got, st = _run(ctx, w["q"], w["t"], 2, PM_KNN_HINT_U8)                 # all defaults
        assert_matches_equal(got, base, "u8 defaults")
        assert st["route"] == 3
    finally:
        ctx.set_option(PM_OPT_KNN_SEEDED, 0)
        ctx.set_option(PM_OPT_KNN_F16_WAVES, 0)
        ctx.set_option(PM_OPT_KNN_RING, 0)
    assert_matches_equal(base[:128], oracle.bf_knn_l2(w["q"][:128], w["t"], 2, nthreads=8), "vs oracle")


@pytest.mark.parametrize("nq,nt", [(700, 1000), (8192, 8192), (300, 20000), (257, 1153), (2048, 2048), (1500, 40000),
                                   (16000, 16500), (8192, 32768)])       # the last two: long sweeps (the split-per-wave form is live)
def test_u8_group_sizes_ring_and_refinements_same_result(ctx, oracle, nq, nt):
    """u8 route: rows per candidate group (4 / 8 / 16) x staging (two buffers / ring of eight with counted waits) x wave
    layout x refinement (integer on the byte copies / canonical f32): one result."""
    w = synth.pair_workload(nq, nt, 128, seed=nq + 3 * nt, planted=0.4, kind="sift")
    base, st0 = _run(ctx, w["q"], w["t"], 2, PM_KNN_HINT_INTEGER)
    try:
        for group in (1, 2, 3):
            ctx.set_option(PM_OPT_KNN_U8_GROUP, group)
            for ring in ((1, 2, 3, 4, 5, 6) if group == 2 else (1,)):   # two buffers / ring + barrier per tile / ring + split-phase counters / operands from global memory
                ctx.set_option(PM_OPT_KNN_RING, ring)
                for waves in ((0,) if ring >= 4 else (0, 2, 3)):   # ring: 8 x 32 queries, 4 x 64, 16 x 32 (512-query workgroups)
                    ctx.set_option(PM_OPT_KNN_F16_WAVES, waves)
                    for pro in ((0, 3, 8) if ring >= 2 and waves != 2 else (0,)):       # tiles requested before the sweep
                        ctx.set_option(PM_OPT_KNN_RING_PROLOGUE, pro)
                        for refine in ((1, 2) if group == 1 else (2,)):
                            ctx.set_option(PM_OPT_KNN_U8_REFINE, refine)
                            for k in (1, 2):
                                got, st = _run(ctx, w["q"], w["t"], k, PM_KNN_HINT_U8)
                                what = "group %d ring %d waves %d prologue %d refine %d k %d" % (group, ring, waves, pro, refine, k)
                                assert_matches_equal(got, base[:, :k], what)
                                assert st["route"] == 3 and st["nonfinite"] == 0 and st["rescans"] <= max(1, st0["rescans"]), (what, st)
    finally:
        for o in (PM_OPT_KNN_U8_GROUP, PM_OPT_KNN_RING, PM_OPT_KNN_F16_WAVES, PM_OPT_KNN_U8_REFINE, PM_OPT_KNN_RING_PROLOGUE):
            ctx.set_option(o, 0)
    assert_matches_equal(base[:128], oracle.bf_knn_l2(w["q"][:128], w["t"], 2, nthreads=8), "vs oracle")


def test_f16s_integer_range_edge(ctx, oracle):
    """Signed integers up to |x| = 361 (the f16-exact bound) through the seeded f16 pass."""
    rng = np.random.default_rng(77)
    q = rng.integers(-361, 362, (200, 128)).astype(np.float32)
    t = rng.integers(-361, 362, (1300, 128)).astype(np.float32)
    want = oracle.bf_knn_l2(q, t, 2, nthreads=8)
    try:
        for seeded in (2, 1):
            ctx.set_option(PM_OPT_KNN_SEEDED, seeded)
            got, st = _run(ctx, q, t, 2, PM_KNN_HINT_INTEGER)
            assert_matches_equal(got, want, "f16 pass (seeded option %d), signed integers" % seeded)
            assert st["route"] == 0 and st["nonfinite"] == 0
    finally:
        ctx.set_option(PM_OPT_KNN_SEEDED, 0)
    got, st = _run(ctx, q, t, 2, PM_KNN_HINT_U8)            # negatives break the u8 premise
    assert_matches_equal(got, want, "u8 hint on signed integers")
    assert st["nonfinite"] == 1


def test_u8_route_c3_full_and_32k_slice(ctx, oracle):
    """BASELINE config C3 at full size through the u8 route, and a 32k x 32k run (16-tile splits: the 9 id bits of an
    integer candidate are all in use) checked on a 1k-query slice."""
    import torch
    q, t, truth = synth.sift_like(8192, 8192, 128, seed=0xC3)
    want = oracle.bf_knn_l2(q, t, 2, nthreads=8)
    got, st = _run(ctx, q, t, 2, PM_KNN_HINT_U8)
    assert_matches_equal(got, want, "C3 u8 route")
    assert st["route"] == 3 and st["rescans"] <= 1
    n = 32768
    q, t, truth = synth.sift_like(n, n, 128, seed=0x32)
    dev = torch.device("cuda", 0)
    d_q, d_t = torch.from_numpy(q).to(dev), torch.from_numpy(t).to(dev)
    d_out = torch.empty((n, 2, 4), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    ctx.knn_diag_enable(True)
    ctx.bf_knn_l2_dev(d_q.data_ptr(), n, d_t.data_ptr(), n, 128, 2, d_out.data_ptr(), PM_KNN_HINT_U8)
    st = ctx.knn_stats()
    ctx.knn_diag_enable(False)
    got = d_out.cpu().numpy().view(pm.MATCH_DTYPE).reshape(n, 2)
    sl = slice(5000, 6024)
    want = oracle.bf_knn_l2(q[sl], t, 2, nthreads=8)
    want["queryIdx"] += 5000
    assert_matches_equal(got[sl], want, "32k slice, u8 route")
    assert st["route"] == 3 and st["rescans"] <= 2 and st["nonfinite"] == 0, st
    planted = truth >= 0
    assert (got["trainIdx"][planted, 0] == truth[planted]).mean() > 0.99


def test_low_dimensional_near_ties_every_coarse_form(ctx, oracle):
    """20-dimensional u8 rows against 33 645 train rows: hundreds of near ties per query, so a coarse pass that ranks on stale
    operands loses true neighbours (at 128 dimensions the planted matches are too far ahead for that to show).  This is the
    case tools/fuzz_campaign.py (seed 41) caught the first version of coarse form 5 with: a counted vmcnt wait over a mix of
    LDS-DMA pieces and register loads returned early — 8 to 70 wrong records per launch, differently on every run."""
    q, t, _ = synth.sift_like(514, 33645, 20, seed=41)
    want = oracle.bf_knn_l2(q, t, 2, nthreads=8)
    q8, t8 = q.astype(np.uint8), t.astype(np.uint8)
    try:
        for ring in (5, 6, 2, 3, 4, 1):
            ctx.set_option(PM_OPT_KNN_RING, ring)
            for rep in range(4):
                assert_matches_equal(ctx.bf_knn_l2_u8(q8, t8, 2), want, "u8 rows, coarse form %d, run %d" % (ring, rep))
            assert_matches_equal(ctx.bf_knn_l2(q, t, 2, PM_KNN_HINT_U8), want, "u8 hint, coarse form %d" % ring)
    finally:
        ctx.set_option(PM_OPT_KNN_RING, 0)
