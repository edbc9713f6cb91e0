"""GPU parity of the shapes the matrix-core routes of pm_bf_knn_l2_f32 took on in round 3 (main.cpp:46 hands the matcher
whatever the DescriptorExtractor produced): dim % 4 != 0, 128 < dim <= 256, rows that are not 16-byte aligned, k = 3 and 4,
on u8-valued, integer and general-float data — against the oracle, and against the exact VALU kernel (PM_OPT_KNN_WIDE = 1:
round 2's route for these shapes)."""
import numpy as np
import pytest
import torch

import points_matching_amd as pm
from points_matching_amd import synth
from points_matching_amd.api import PM_KNN_FORCE_EXACT, PM_KNN_HINT_INTEGER, PM_KNN_HINT_U8, PM_OPT_KNN_WIDE
from util import assert_matches_equal

pytestmark = pytest.mark.gpu


def _data(kind, nq, nt, dim, seed):
    rng = np.random.default_rng(seed)
    if kind == "u8":
        q = rng.integers(0, 256, (nq, dim)).astype(np.float32)
        t = rng.integers(0, 256, (nt, dim)).astype(np.float32)
        n_pl = nq // 2                                              # planted near-duplicates: a real nearest neighbour
        t_src = rng.integers(0, nt, n_pl)
        q[:n_pl] = np.clip(t[t_src] + rng.integers(-6, 7, (n_pl, dim)), 0, 255)
        return q, t
    q, t, _ = synth.surf_like(nq, nt, dim, seed=seed)
    return q, t


@pytest.mark.parametrize("nq,nt,dim", [(700, 900, 130), (512, 2048, 256), (300, 1500, 200), (129, 257, 7), (1000, 1000, 129),
                                       (64, 5000, 255), (2048, 2048, 192), (33, 40, 3)])
@pytest.mark.parametrize("kind", ["u8", "surf"])
def test_wide_and_unaligned_dims_take_the_f16_pass(ctx, oracle, nq, nt, dim, kind):
    q, t = _data(kind, nq, nt, dim, nq + nt + dim)
    for k in (1, 2):
        want = oracle.bf_knn_l2(q, t, k, nthreads=8)
        for flags in ((0, PM_KNN_HINT_INTEGER, PM_KNN_HINT_U8) if kind == "u8" else (0,)):
            ctx.knn_diag_enable(True)
            got = ctx.bf_knn_l2(q, t, k, flags)
            st = ctx.knn_stats()
            ctx.knn_diag_enable(False)
            assert_matches_equal(got, want, "dim %d k %d %s flags %d" % (dim, k, kind, flags))
            assert st["route"] == (0 if kind == "u8" else 1) and st["nonfinite"] == 0, st      # the f16 pass (exact / rounded copies)
            assert st["rescans"] <= max(2, nq // 200), st
    try:
        ctx.set_option(PM_OPT_KNN_WIDE, 1)
        assert_matches_equal(ctx.bf_knn_l2(q, t, 2), oracle.bf_knn_l2(q, t, 2, nthreads=8), "exact kernel route")
    finally:
        ctx.set_option(PM_OPT_KNN_WIDE, 0)


@pytest.mark.parametrize("nq,nt,dim", [(700, 900, 128), (1024, 4096, 64), (300, 1500, 200), (129, 257, 30), (2048, 2048, 128)])
@pytest.mark.parametrize("kind", ["u8", "surf"])
def test_k3_and_k4_on_the_matrix_routes(ctx, oracle, nq, nt, dim, kind):
    q, t = _data(kind, nq, nt, dim, 7 * nq + nt)
    t[5] = t[9]; t[77] = t[9]; q[3] = t[9]                           # exact ties among the first neighbours
    for k in (3, 4):
        want = oracle.bf_knn_l2(q, t, k, nthreads=8)
        for flags in ((0, PM_KNN_HINT_INTEGER, PM_KNN_HINT_U8) if kind == "u8" else (0,)):
            got = ctx.bf_knn_l2(q, t, k, flags)
            assert_matches_equal(got, want, "k %d dim %d %s flags %d" % (k, dim, kind, flags))
        assert_matches_equal(ctx.bf_knn_l2(q, t, k, PM_KNN_FORCE_EXACT), want, "exact")
    if kind == "u8" and dim % 4 == 0 and dim <= 128:
        got = ctx.bf_knn_l2_u8(q.astype(np.uint8), t.astype(np.uint8), 4)
        assert_matches_equal(got, oracle.bf_knn_l2(q, t, 4, nthreads=8), "u8 rows, k = 4")


def test_rows_that_are_not_16_byte_aligned(ctx, oracle):
    """Device pointers offset by 4 bytes (a slice of a larger allocation): element loads, same records."""
    q, t = _data("u8", 600, 1100, 128, 5)
    dev = torch.device("cuda", 0)
    big_q = torch.zeros(q.size + 1, dtype=torch.float32, device=dev)
    big_t = torch.zeros(t.size + 1, dtype=torch.float32, device=dev)
    big_q[1:] = torch.from_numpy(q.reshape(-1)).to(dev)
    big_t[1:] = torch.from_numpy(t.reshape(-1)).to(dev)
    out = torch.zeros((600, 2, 4), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    want = oracle.bf_knn_l2(q, t, 2, nthreads=8)
    for flags in (0, PM_KNN_HINT_INTEGER, PM_KNN_HINT_U8):
        ctx.bf_knn_l2_dev(big_q.data_ptr() + 4, 600, big_t.data_ptr() + 4, 1100, 128, 2, out.data_ptr(), flags)
        ctx.synchronize()
        assert_matches_equal(out.cpu().numpy().view(pm.MATCH_DTYPE).reshape(600, 2), want, "unaligned rows, flags %d" % flags)
