"""GPU parity: pm_bf_knn_l2_f32 (MFMA coarse pass + canonical refinement, and the exact kernel)
against the CPU oracle — bit-exact trainIdx and distance bits (docs/SPEC.md S1/S3).
Slot in the reference: matcher.match(...) at main.cpp:46 with the BF-L2 matcher of main.cpp:43."""
import numpy as np
import pytest

import points_matching_amd as pm
from points_matching_amd import synth
from points_matching_amd.api import PM_KNN_FORCE_EXACT, PM_KNN_FORCE_F32, PM_KNN_HINT_INTEGER
from util import assert_matches_equal

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("gen", ["sift", "surf"])
@pytest.mark.parametrize("k", [1, 2])
def test_knn_small_both_paths(ctx, oracle, gen, k):
    q, t, truth = (synth.sift_like if gen == "sift" else synth.surf_like)(256, 256, 128, seed=11)
    want = oracle.bf_knn_l2(q, t, k)
    assert_matches_equal(ctx.bf_knn_l2(q, t, k), want, "auto route")
    assert_matches_equal(ctx.bf_knn_l2(q, t, k, PM_KNN_FORCE_F32), want, "f32-MFMA route")
    # integer hint: the f16 route for sift data; for surf data the hint is WRONG and must only cost time
    assert_matches_equal(ctx.bf_knn_l2(q, t, k, PM_KNN_HINT_INTEGER), want, "f16 route / wrong hint")
    assert_matches_equal(ctx.bf_knn_l2(q, t, k, PM_KNN_FORCE_EXACT), want, "exact")
    planted = truth >= 0
    assert (want["trainIdx"][planted, 0] == truth[planted]).all()


@pytest.mark.parametrize("nq,nt,dim", [(100, 77, 128), (1, 300, 128), (300, 1, 128), (65, 64, 64),
                                       (129, 130, 32), (33, 200, 96), (50, 70, 12), (17, 5, 8)])
def test_knn_ragged_shapes(ctx, oracle, nq, nt, dim):
    q, t, _ = synth.surf_like(nq, nt, dim, seed=nq * 1000 + nt)
    for k in (1, 2):
        want = oracle.bf_knn_l2(q, t, k)
        assert_matches_equal(ctx.bf_knn_l2(q, t, k), want, "fast %s" % ((nq, nt, dim, k),))
        assert_matches_equal(ctx.bf_knn_l2(q, t, k, PM_KNN_FORCE_EXACT), want, "exact")
    qi, ti, _ = synth.sift_like(nq, nt, dim, seed=nq * 1000 + nt + 1)       # integer data: f16 route
    for k in (1, 2):
        want = oracle.bf_knn_l2(qi, ti, k)
        for flags in (0, PM_KNN_HINT_INTEGER, PM_KNN_FORCE_F32):
            assert_matches_equal(ctx.bf_knn_l2(qi, ti, k, flags), want, "int %s" % ((nq, nt, dim, k, flags),))


@pytest.mark.parametrize("nq,nt,dim,k", [(40, 90, 7, 2), (40, 90, 200, 1), (70, 300, 130, 3),
                                         (64, 100, 128, 5), (30, 40, 128, 16), (10, 3, 64, 4),
                                         (20, 200, 300, 9)])
def test_knn_general_kernel(ctx, oracle, nq, nt, dim, k):
    """dims that are not a multiple of 4 / above 128 and k > 2 take the exact kernel."""
    q, t, _ = synth.surf_like(nq, nt, dim, seed=dim * 7 + k)
    assert_matches_equal(ctx.bf_knn_l2(q, t, k), oracle.bf_knn_l2(q, t, k), str((nq, nt, dim, k)))


def test_knn_duplicate_rows_lowest_index_wins(ctx, oracle):
    q, t, _ = synth.sift_like(64, 256, 128, seed=3)
    t[200] = t[10]
    t[77] = t[10]
    q[5] = t[10]
    # a run of identical rows longer than the coarse candidate list forces the re-scan branch
    t[128:140] = t[20]
    q[6] = t[20]
    want = oracle.bf_knn_l2(q, t, 2)
    assert want["trainIdx"][5, 0] == 10 and want["trainIdx"][5, 1] == 77
    assert want["trainIdx"][6, 0] == 20 and want["trainIdx"][6, 1] == 128
    for flags in (0, PM_KNN_HINT_INTEGER, PM_KNN_FORCE_F32, PM_KNN_FORCE_EXACT):
        assert_matches_equal(ctx.bf_knn_l2(q, t, 2, flags), want, "dups flags=%d" % flags)


def test_knn_near_ties_general_floats(ctx, oracle):
    """Train rows that differ from each other by a few ulps: the coarse MFMA ranking cannot
    separate them, the canonical refinement must."""
    rng = np.random.default_rng(5)
    q, t, _ = synth.surf_like(128, 512, 128, seed=9)
    for i in range(0, 128, 4):
        base = q[i].copy()
        for r in range(6):
            row = base.copy()
            row[rng.integers(0, 128, 3)] *= np.float32(1 + (r + 1) * 1.2e-7)
            t[(i * 3 + r * 17) % 512] = row
    want = oracle.bf_knn_l2(q, t, 2)
    assert_matches_equal(ctx.bf_knn_l2(q, t, 2), want, "near ties")


def test_knn_non_finite_inputs_take_the_exact_route(ctx, oracle):
    q, t, _ = synth.surf_like(40, 100, 128, seed=21)
    t[7, 3] = np.nan
    t[9, 0] = np.inf
    q[4, 1] = np.nan
    want = oracle.bf_knn_l2(q, t, 2)
    got = ctx.bf_knn_l2(q, t, 2)
    # NaN payload bits are not part of the contract: compare indices, and distances where finite
    assert (got["trainIdx"] == want["trainIdx"]).all()
    fin = np.isfinite(want["distance"])
    assert (got["distance"][fin].view(np.uint32) == want["distance"][fin].view(np.uint32)).all()
    assert (np.isnan(got["distance"]) == np.isnan(want["distance"])).all()


def test_knn_empty_and_invalid(ctx):
    q = np.zeros((0, 128), np.float32)
    t = np.zeros((10, 128), np.float32)
    assert ctx.bf_knn_l2(q, t, 2).shape == (0, 2)
    out = ctx.bf_knn_l2(np.zeros((3, 128), np.float32), np.zeros((0, 128), np.float32), 2)
    assert (out["trainIdx"] == -1).all() and np.isinf(out["distance"]).all()
    assert (out["queryIdx"][:, 0] == np.arange(3)).all()
    with pytest.raises(pm.PmError):
        ctx.bf_knn_l2(t, t, 0)
    with pytest.raises(pm.PmError):
        ctx.bf_knn_l2(t, t, 17)


def test_knn_c2_2k_full_parity(ctx, oracle):
    """BASELINE config C2: 2k x 2k SIFT-128, k=2 + ratio test."""
    q, t, truth = synth.sift_like(2048, 2048, 128, seed=0xC2)
    want = oracle.bf_knn_l2(q, t, 2, nthreads=8)
    got = ctx.bf_knn_l2(q, t, 2)
    assert_matches_equal(got, want, "C2 sift")
    good = pm.api.filter_ratio(got, 0.8)
    good_o = oracle.filter_ratio(want, 0.8)
    assert_matches_equal(good, good_o, "C2 ratio")
    planted = np.nonzero(truth >= 0)[0]
    kept = set(good["queryIdx"].tolist())
    assert len(kept & set(planted.tolist())) > 0.9 * planted.size
    q, t, _ = synth.surf_like(2048, 2048, 128, seed=0xC2)
    assert_matches_equal(ctx.bf_knn_l2(q, t, 2), oracle.bf_knn_l2(q, t, 2, nthreads=8), "C2 surf")


def test_knn_c3_8k_full_size(ctx, oracle):
    """BASELINE config C3 matcher: 8k x 8k x 128.  Full oracle parity (the vectorised oracle
    needs a few seconds) plus size-independent properties."""
    q, t, truth = synth.sift_like(8192, 8192, 128, seed=0xC3)
    got = ctx.bf_knn_l2(q, t, 2)
    want = oracle.bf_knn_l2(q, t, 2, nthreads=8)
    assert_matches_equal(got, want, "C3")
    assert_matches_equal(ctx.bf_knn_l2(q, t, 2, PM_KNN_FORCE_F32), want, "C3 f32 route")
    assert_matches_equal(ctx.bf_knn_l2(q, t, 2, PM_KNN_HINT_INTEGER), want, "C3 f16 route")
    # properties: rows sorted, indices in range and distinct, planted pairs found, and the
    # reported distance is the canonical distance of the reported pair
    assert (got["distance"][:, 0] <= got["distance"][:, 1]).all()
    assert (got["trainIdx"] >= 0).all() and (got["trainIdx"] < 8192).all()
    assert (got["trainIdx"][:, 0] != got["trainIdx"][:, 1]).all()
    planted = truth >= 0
    assert (got["trainIdx"][planted, 0] == truth[planted]).mean() > 0.999
    for i in (0, 17, 4095, 8191):
        d = np.sqrt(oracle.l2sqr(q[i], t[got["trainIdx"][i, 0]]))
        assert np.float32(d).view(np.uint32) == got["distance"][i, 0].view(np.uint32)
    # permuting the train rows permutes the indices
    perm = np.random.default_rng(1).permutation(8192)
    got_p = ctx.bf_knn_l2(q[:512], t[perm], 1)
    assert (perm[got_p["trainIdx"][:, 0]] == got["trainIdx"][:512, 0]).mean() > 0.999


def test_knn_integer_eligibility_edge(ctx, oracle):
    """|x| <= 361 is f16-exact territory; one larger value must push the call to the f32 route
    (auto) or the exact re-scan (wrong hint) with identical results."""
    rng = np.random.default_rng(77)
    q = rng.integers(-361, 362, (200, 128)).astype(np.float32)
    t = rng.integers(-361, 362, (300, 128)).astype(np.float32)
    want = oracle.bf_knn_l2(q, t, 2)
    for flags in (0, PM_KNN_HINT_INTEGER, PM_KNN_FORCE_F32):
        assert_matches_equal(ctx.bf_knn_l2(q, t, 2, flags), want, "edge flags=%d" % flags)
    t[17, 5] = 400.0
    q[3, 9] = 0.5
    want = oracle.bf_knn_l2(q, t, 2)
    for flags in (0, PM_KNN_HINT_INTEGER, PM_KNN_FORCE_F32):
        assert_matches_equal(ctx.bf_knn_l2(q, t, 2, flags), want, "beyond flags=%d" % flags)


def test_knn_randomised_shapes_values_and_routes(ctx, oracle):
    """Seeded sweep over shapes, dims, k, value families and coarse routes: every combination must
    give the oracle's bits (indices and distance patterns)."""
    rng = np.random.default_rng(0xC0FFEE)
    flags_all = [0, pm.api.PM_KNN_FORCE_F32, pm.api.PM_KNN_HINT_INTEGER, pm.api.PM_KNN_FORCE_EXACT]
    for case in range(40):
        nq = int(rng.integers(1, 700))
        nt = int(rng.integers(1, 900))
        dim = int(rng.choice([4, 8, 12, 20, 32, 36, 64, 100, 128]))
        k = int(rng.choice([1, 2, 2, 3]))
        fam = case % 4
        if fam == 0:                                   # integer-valued (SIFT-like), some exceeding the f16 bound
            q = rng.integers(0, 256, (nq, dim)).astype(np.float32)
            t = rng.integers(0, 256, (nt, dim)).astype(np.float32)
            if case % 8 == 4:
                t[rng.integers(0, nt), rng.integers(0, dim)] = 1000.0
        elif fam == 1:                                 # general floats of mixed scale
            q = (rng.standard_normal((nq, dim)) * 10.0 ** rng.integers(-3, 4)).astype(np.float32)
            t = (rng.standard_normal((nt, dim)) * 10.0 ** rng.integers(-3, 4)).astype(np.float32)
        elif fam == 2:                                 # many exact duplicates and near ties
            base = rng.integers(0, 4, (5, dim)).astype(np.float32)
            q = base[rng.integers(0, 5, nq)]
            t = base[rng.integers(0, 5, nt)]
            t[::3] += np.float32(2.0 ** -20)
        else:                                          # unit-norm (SURF-like) with planted neighbours
            t = rng.standard_normal((nt, dim)).astype(np.float32)
            t /= np.linalg.norm(t, axis=1, keepdims=True) + 1e-12
            q = t[rng.integers(0, nt, nq)] + (rng.standard_normal((nq, dim)) * 0.01).astype(np.float32)
        flags = flags_all[int(rng.integers(0, 4))]
        got = ctx.bf_knn_l2(q, t, k, flags)
        want = oracle.bf_knn_l2(q, t, k, nthreads=4)
        assert_matches_equal(got, want, "case %d: nq=%d nt=%d dim=%d k=%d fam=%d flags=%d" % (case, nq, nt, dim, k, fam, flags))


@pytest.mark.parametrize("kind", ["sift", "surf"])
def test_knn_32k_long_sweeps_slice_and_no_list_overflow(ctx, oracle, kind):
    """32k x 32k: many query blocks -> few, long splits.  A 1k-query slice against the oracle, and the
    refinement must not fall back to split re-scans (the id bits embedded in the candidates widen the
    window; splits are capped at 2048 rows for that reason)."""
    import torch
    n = 32768
    q, t, truth = (synth.sift_like if kind == "sift" else synth.surf_like)(n, n, 128, seed=0x32)
    dev = torch.device("cuda", 0)
    d_q, d_t = torch.from_numpy(q).to(dev), torch.from_numpy(t).to(dev)
    d_out = torch.empty((n, 2, 4), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    ctx.knn_diag_enable(True)
    ctx.bf_knn_l2_dev(d_q.data_ptr(), n, d_t.data_ptr(), n, 128, 2, d_out.data_ptr(), 0)
    st = ctx.knn_stats()
    ctx.knn_diag_enable(False)
    got = d_out.cpu().numpy().view(pm.MATCH_DTYPE).reshape(n, 2)
    sl = slice(5000, 6024)
    want = oracle.bf_knn_l2(q[sl], t, 2, nthreads=8)
    want["queryIdx"] += 5000
    assert_matches_equal(got[sl], want, "32k slice " + kind)
    assert st["rescans"] <= 2 and st["nonfinite"] == 0, st
    planted = truth >= 0
    assert (got["trainIdx"][planted, 0] == truth[planted]).mean() > 0.99


@pytest.mark.parametrize("nq,nt", [(700, 1000), (8192, 8192), (300, 20000)])
def test_lds_dma_staging_same_result(ctx, oracle, nq, nt):
    """PM_OPT_KNN_STAGING: the train tiles of the f16 coarse kernel go global -> LDS by LDS-DMA (2, the default: padded
    rows through the per-lane source address) or through registers (1); same candidates, same result."""
    w = synth.pair_workload(nq, nt, 128, seed=nq ^ nt, planted=0.4, kind="sift")
    ctx.knn_diag_enable(True)
    base = ctx.bf_knn_l2(w["q"], w["t"], 2, pm.api.PM_KNN_HINT_INTEGER)
    base_rescans = ctx.knn_stats()["rescans"]
    assert base_rescans <= 1
    try:
        for staging in (1, 2):
            ctx.set_option(pm.api.PM_OPT_KNN_STAGING, staging)
            for waves in (0, 1, 2, 3):           # 3: two row groups of 4 waves x 64 queries, lists merged in LDS
                ctx.set_option(pm.api.PM_OPT_KNN_F16_WAVES, waves)
                ctx.knn_diag_enable(True)
                got = ctx.bf_knn_l2(w["q"], w["t"], 2, pm.api.PM_KNN_HINT_INTEGER)
                st = ctx.knn_stats()
                assert_matches_equal(got, base, "staging %d, waves %d" % (staging, waves))
                # a wrong candidate id or a lost list would show as a re-scan (the refinement repairs it silently)
                assert st["rescans"] == base_rescans and st["nonfinite"] == 0, (staging, waves, st)
        for order in (1, 2):                     # workgroup order: launch order / one 2-D grid tile per XCD
            ctx.set_option(pm.api.PM_OPT_KNN_XCD_TILE, order)
            for flags in (pm.api.PM_KNN_HINT_INTEGER, pm.api.PM_KNN_FORCE_F32):
                got = ctx.bf_knn_l2(w["q"], w["t"], 2, flags)
                assert_matches_equal(got, base, "xcd order %d flags %d" % (order, flags))
                assert ctx.knn_stats()["rescans"] <= 1
    finally:
        ctx.knn_diag_enable(False)
        ctx.set_option(pm.api.PM_OPT_KNN_STAGING, 0)
        ctx.set_option(pm.api.PM_OPT_KNN_F16_WAVES, 0)
        ctx.set_option(pm.api.PM_OPT_KNN_XCD_TILE, 0)
    assert_matches_equal(base[:128], oracle.bf_knn_l2(w["q"][:128], w["t"], 2, nthreads=8), "vs oracle")


def test_knn_l2_randomised(ctx, oracle):
    """Seeded sweep over shapes, descriptor kinds, route flags and adversarial content: duplicated rows (exact ties),
    rows that break the integer hint (fractions, negatives, values beyond the f16-exact range), widely scaled rows.
    Every route must return the canonical result bit for bit."""
    rng = np.random.default_rng(0xBEEF)
    for case in range(28):
        dim = int(rng.choice([4, 8, 16, 32, 64, 96, 128, 128, 128]))
        nq = int(rng.integers(1, 2500))
        nt = int(rng.integers(1, 6000))
        k = int(rng.choice([1, 2]))
        kind = str(rng.choice(["sift", "surf"]))
        q, t, _ = (synth.sift_like if kind == "sift" else synth.surf_like)(nq, nt, dim, seed=500 + case)
        twist = int(rng.integers(0, 5))
        if twist == 1 and nt > 8:                       # runs of identical train rows, one of them a query's exact match
            for _ in range(4):
                a = int(rng.integers(0, nt)); ln = int(rng.integers(2, min(12, nt)))
                b = int(rng.integers(0, nt - ln + 1))
                t[b:b + ln] = t[a]
                q[int(rng.integers(0, nq))] = t[a]
        elif twist == 2:                                # break the integer premise in a few places
            t[int(rng.integers(0, nt)), int(rng.integers(0, dim))] += 0.5
            q[int(rng.integers(0, nq)), int(rng.integers(0, dim))] = -3.0
        elif twist == 3:                                # beyond the range where f16 products stay exact
            t[int(rng.integers(0, nt))] *= 7.0
            q[int(rng.integers(0, nq))] *= 1000.0
        elif twist == 4:                                # mixed scales
            t[::3] *= np.float32(1e-3)
            q[::2] *= np.float32(37.0)
        want = oracle.bf_knn_l2(q, t, k, nthreads=8)
        for flags in (0, PM_KNN_HINT_INTEGER, PM_KNN_FORCE_F32, PM_KNN_FORCE_EXACT):
            assert_matches_equal(ctx.bf_knn_l2(q, t, k, flags), want,
                                 "case %d: %s %dx%dx%d k=%d twist=%d flags=%d" % (case, kind, nq, nt, dim, k, twist, flags))


def _route_of(ctx, q, t, k=2, flags=0):
    ctx.knn_diag_enable(True)
    try:
        got = ctx.bf_knn_l2(q, t, k, flags)
        st = ctx.knn_stats()
    finally:
        ctx.knn_diag_enable(False)
    return got, st


@pytest.mark.parametrize("nq,nt,dim", [(8192, 8192, 128), (1000, 3000, 64), (257, 129, 128), (50, 5000, 32)])
def test_general_floats_take_the_rounded_f16_pass(ctx, oracle, nq, nt, dim):
    """Automatic route, SURF-like unit-norm floats (the reference's own descriptors, main.cpp:37-40): the coarse pass
    runs on f16-rounded scaled copies (route 1) and the refinement's wider window keeps the result canonical; the
    f32-input pass (PM_OPT_KNN_GENERAL_F16 = 1, route 2) gives the same bits.  Integer data stay on route 0."""
    q, t, _ = synth.surf_like(nq, nt, dim, seed=nq + nt)
    want = oracle.bf_knn_l2(q, t, 2, nthreads=8) if nq <= 1000 else None
    got, st = _route_of(ctx, q, t)
    assert st["route"] == 1 and st["nonfinite"] == 0 and st["rescans"] <= max(2, nq // 500), st
    try:
        ctx.set_option(pm.api.PM_OPT_KNN_GENERAL_F16, 1)
        got32, st32 = _route_of(ctx, q, t)
    finally:
        ctx.set_option(pm.api.PM_OPT_KNN_GENERAL_F16, 0)
    assert st32["route"] == 2
    assert_matches_equal(got, got32, "rounded-f16 pass vs f32-input pass")
    if want is not None:
        assert_matches_equal(got, want, "vs oracle")
    else:
        sl = slice(3000, 3512)
        w = oracle.bf_knn_l2(q[sl], t, 2, nthreads=8)
        w["queryIdx"] += 3000
        assert_matches_equal(got[sl], w, "slice vs oracle")
    qi, ti, _ = synth.sift_like(min(nq, 512), min(nt, 2048), dim, seed=3)
    _, sti = _route_of(ctx, qi, ti)
    assert sti["route"] == 0


def test_general_floats_scales(ctx, oracle):
    """The rounded-copy route rescales by powers of two — the train matrix by one factor, every query row by its own —
    so it covers any pair of scales: tiny queries lose bits (wider window, at worst re-scans), queries more than 8x
    larger than every train row are not ranked at all (exact scan of that query).  Whatever it does, the result is
    the canonical one; a non-finite input sends every query to the exact scan."""
    q, t, _ = synth.surf_like(300, 2000, 128, seed=77)
    cases = [("tiny", q * np.float32(1e-12), t * np.float32(1e-12)),
             ("huge", q * np.float32(3e15), t * np.float32(3e15)),
             ("train x64", q, t * np.float32(64.0)),
             ("train x1e4 (queries 2^-13 of the train scale)", q, t * np.float32(1e4)),
             ("query x6", q * np.float32(6.0), t)]
    t_out = t.copy(); t_out[17] *= np.float32(50.0)              # one large row sets the train scale
    cases.append(("one large train row", q, t_out))
    t_out = t.copy(); t_out[17] *= np.float32(5000.0)
    cases.append(("one giant train row", q, t_out))
    q_mix = q.copy(); q_mix[::2] *= np.float32(1e-4)
    cases.append(("queries of mixed scale", q_mix, t))
    for name, qq, tt in cases:
        got, st = _route_of(ctx, qq, tt)
        assert st["route"] == 1 and st["nonfinite"] == 0, (name, st)
        assert_matches_equal(got, oracle.bf_knn_l2(qq, tt, 2, nthreads=8), name)
    # queries far above the train scale have no f16 image: each of them is scanned exactly
    qq = q.copy(); qq[:40] *= np.float32(1000.0)
    got, st = _route_of(ctx, qq, t)
    assert st["route"] == 1 and st["rescans"] >= 40, st
    assert_matches_equal(got, oracle.bf_knn_l2(qq, t, 2, nthreads=8), "40 queries x1000")
    t_nan = t.copy(); t_nan[5, 3] = np.inf
    got, st = _route_of(ctx, q, t_nan)
    assert st["nonfinite"] == 1
    assert_matches_equal(got, oracle.bf_knn_l2(q, t_nan, 2, nthreads=8), "non-finite input")
    try:                                                          # the f32-input pass stays selectable
        ctx.set_option(pm.api.PM_OPT_KNN_GENERAL_F16, 1)
        got, st = _route_of(ctx, q, t * np.float32(1e4))
        assert st["route"] == 2
        assert_matches_equal(got, oracle.bf_knn_l2(q, t * np.float32(1e4), 2, nthreads=8), "f32-input pass")
    finally:
        ctx.set_option(pm.api.PM_OPT_KNN_GENERAL_F16, 0)


@pytest.mark.parametrize("nq,nt,dim,k", [(512, 600, 64, 2), (300, 2000, 128, 2), (8192, 8192, 128, 2), (129, 130, 130, 3), (64, 100, 200, 1)])
def test_unit_norm_hint_one_prep_launch_same_result(ctx, oracle, nq, nt, dim, k):
    """PM_KNN_HINT_UNIT_NORM: unit-norm general floats (SURF: main.cpp:37-40) through the f16 matrix pass with ONE prep launch.
    Same records as the automatic route and the oracle; smaller-than-stated norms only widen the window; a train row beyond
    the bound is detected on the device and every query is scanned exactly."""
    q, t, _ = synth.surf_like(nq, nt, dim, seed=nq + nt + dim)
    want = oracle.bf_knn_l2(q, t, k, nthreads=8)
    H = pm.api.PM_KNN_HINT_UNIT_NORM
    ctx.knn_diag_enable(True)
    try:
        ctx.timing_enable(True); ctx.timing_reset()
        got = ctx.bf_knn_l2(q, t, k, H)
        st = ctx.knn_stats()
        launches = ctx.timing_get("knn_l2_prep")[1]
        ctx.timing_enable(False)
        assert_matches_equal(got, want, "unit-norm hint")
        assert st["route"] == 1 and st["nonfinite"] == 0 and launches == 1, (st, launches)
        for name, qq, tt in (("train rows at 0.3", q, t * np.float32(0.3)), ("queries x5", q * np.float32(5.0), t),
                             ("tiny train rows", q, t * np.float32(1e-3)), ("a zero row each", _with_zero_row(q), _with_zero_row(t))):
            w2 = oracle.bf_knn_l2(qq, tt, k, nthreads=8)
            assert_matches_equal(ctx.bf_knn_l2(qq, tt, k, H), w2, name)
            assert ctx.knn_stats()["nonfinite"] == 0, name
        tt = t.copy(); tt[nt // 2] *= np.float32(1.5)                   # the hint is wrong for one row: exact scan, same answer
        assert_matches_equal(ctx.bf_knn_l2(q, tt, k, H), oracle.bf_knn_l2(q, tt, k, nthreads=8), "one train row beyond the bound")
        assert ctx.knn_stats()["nonfinite"] == 1
        qi, ti, _ = synth.sift_like(min(nq, 400), min(nt, 900), dim if dim % 4 == 0 else 128, seed=3)
        assert_matches_equal(ctx.bf_knn_l2(qi, ti, k, H), oracle.bf_knn_l2(qi, ti, k, nthreads=8), "integer data under the hint (wrong: norms >> 1)")
    finally:
        ctx.knn_diag_enable(False)


def _with_zero_row(x):
    y = x.copy()
    y[len(y) // 3] = 0.0
    return y


def test_general_floats_near_ties_below_f16_resolution(ctx, oracle):
    """Train rows that differ by less than an f16 ulp: the rounded copies are IDENTICAL, the window must still deliver
    every one of them to the refinement (runs longer than a candidate list force the re-scan branch)."""
    rng = np.random.default_rng(21)
    q, t, _ = synth.surf_like(128, 4096, 128, seed=31)
    for i in range(0, 128, 3):
        base = q[i].copy()
        n_dup = int(rng.integers(2, 24))
        rows = rng.choice(4096, size=n_dup, replace=False)
        for r in rows:
            t[r] = base * (1.0 + np.float32(rng.uniform(-3e-5, 3e-5, size=128)))
    got, st = _route_of(ctx, q, t)
    assert st["route"] == 1
    assert_matches_equal(got, oracle.bf_knn_l2(q, t, 2, nthreads=8), "near ties")
