"""GPU parity of the one-launch RANSAC-F kernel (csrc/ransac_fused.hip) in its three uses — device-resident local
run, sharded run over a correspondence VIEW (padded parts + device counts) and the finish after the exchange — against
the CPU oracle on the concatenated correspondences.  Slot in the reference: cv::findFundamentalMat, main.cpp:95-98;
the view is what the all-gathered survivor blocks of a query-row-sharded matcher look like."""
import numpy as np
import pytest
import torch

import points_matching_amd as pm
from points_matching_amd import synth
from points_matching_amd.api import PM_ERR_SAMPSON, PM_ERR_SYM_EPIPOLAR, PointsView, RANSAC_RECORD_DTYPE

pytestmark = pytest.mark.gpu


def _view(g1, g2, gn, cap):
    parts = g1.shape[0]
    return PointsView(g1.data_ptr(), g2.data_ptr(), gn.data_ptr() if gn is not None else None, parts, cap,
                      g1.stride(0) if parts > 1 else 0, gn.stride(0) if gn is not None else 1, 0)


def _parts(x1, x2, counts, cap, dev):
    """Scatter the concatenated correspondences into padded parts (garbage behind each count)."""
    parts = len(counts)
    rng = np.random.default_rng(1)
    g1 = rng.uniform(0, 900, (parts, cap, 2)).astype(np.float32)
    g2 = rng.uniform(0, 600, (parts, cap, 2)).astype(np.float32)
    o = 0
    for p, c in enumerate(counts):
        g1[p, :c] = x1[o:o + c]
        g2[p, :c] = x2[o:o + c]
        o += c
    assert o == x1.shape[0]
    return (torch.from_numpy(g1).to(dev), torch.from_numpy(g2).to(dev),
            torch.tensor(counts, dtype=torch.int32, device=dev))


@pytest.mark.parametrize("counts,cap", [([700], 700), ([300, 0, 511, 64], 512), ([1, 7, 0, 0, 3], 16), ([2500, 2300], 2560),
                                        ([5000, 5100], 5120), ([3], 8)])
@pytest.mark.parametrize("kind", [PM_ERR_SAMPSON, PM_ERR_SYM_EPIPOLAR])
def test_sharded_run_over_a_view_equals_the_oracle(ctx, oracle, counts, cap, kind):
    dev = torch.device("cuda", 0)
    n, H, shards, seed, thr = sum(counts), 1200, 3, 0x5EED + len(counts), 1.0
    x1, x2, _, _ = synth.two_view(max(n, 1), seed=17 * n + cap, outlier_frac=0.3, noise_px=0.5)
    x1, x2 = x1[:n], x2[:n]
    g1, g2, gn = _parts(x1, x2, counts, cap, dev)
    view = _view(g1, g2, gn, cap)
    recs = torch.zeros(shards * 10, dtype=torch.float64, device=dev)        # 80-byte records
    for r in range(shards):
        ctx.set_option(pm.api.PM_OPT_RANSAC_FORM, 1 if r == 1 else 0)       # (the middle shard on the round-2 register form)
        ctx.ransac_shard_parts_dev(view, r * H // shards, (r + 1) * H // shards, thr, seed, recs.data_ptr() + 80 * r, kind)
    ctx.set_option(pm.api.PM_OPT_RANSAC_FORM, 0)
    mask_len = len(counts) * cap + 5
    d_key = torch.zeros(1, dtype=torch.int64, device=dev)
    d_F = torch.zeros(9, dtype=torch.float64, device=dev)
    d_mask = torch.full((mask_len,), 7, dtype=torch.uint8, device=dev)
    d_ninl = torch.full((1,), -1, dtype=torch.int32, device=dev)
    d_ntot = torch.full((1,), -1, dtype=torch.int32, device=dev)
    ctx.ransac_finish_parts_dev(view, thr, recs.data_ptr(), shards, d_key.data_ptr(), d_F.data_ptr(), d_mask.data_ptr(),
                                mask_len, d_ninl.data_ptr(), d_ntot.data_ptr(), kind)
    ctx.synchronize()
    rc, F_o, mask_o, ninl_o, key_o = oracle.ransac_fundamental(x1, x2, H, thr, seed, kind, nthreads=8)
    rec = recs.cpu().numpy().view(RANSAC_RECORD_DTYPE)
    assert int(d_ntot.item()) == n
    if n < 8 or rc != 0:
        assert int(d_key.item()) == 0 and not d_mask.cpu().numpy().any() and int(d_ninl.item()) == 0
        assert not rec["key"].any()
        return
    # every shard's record is the oracle's answer for that id range (key AND model bits)
    for r in range(shards):
        _, F_r, _, _, key_r = oracle.ransac_fundamental(x1, x2, (r + 1) * H // shards, thr, seed, kind, hyp_begin=r * H // shards)
        assert int(rec["key"][r]) == key_r, r
        assert (rec["F"][r].view(np.uint64) == F_r.reshape(9).view(np.uint64)).all(), r
    assert int(d_key.cpu().numpy().view(np.uint64)[0]) == key_o
    assert (d_F.cpu().numpy().view(np.uint64) == F_o.reshape(9).view(np.uint64)).all()
    m = d_mask.cpu().numpy()
    assert (m[:n] == mask_o).all() and not m[n:].any()
    assert int(d_ninl.item()) == ninl_o


@pytest.mark.parametrize("n,cap,H", [(2275, 8192, 10000), (573, 2048, 10000), (40, 4096, 300), (9000, 10240, 700), (7, 64, 50)])
def test_device_resident_local_run_with_a_device_count(ctx, oracle, n, cap, H):
    """pm_ransac_run_dev (what bench.py and the pair batch call): count on the device, buffers larger than the count."""
    dev = torch.device("cuda", 0)
    x1, x2, _, _ = synth.two_view(max(n, 8), seed=n + H, outlier_frac=0.3, noise_px=0.5)
    x1, x2 = x1[:n], x2[:n]
    b1 = np.full((cap, 2), 123.0, np.float32); b1[:n] = x1
    b2 = np.full((cap, 2), 321.0, np.float32); b2[:n] = x2
    d1, d2 = torch.from_numpy(b1).to(dev), torch.from_numpy(b2).to(dev)
    dn = torch.tensor([n], dtype=torch.int32, device=dev)
    d_key = torch.zeros(1, dtype=torch.int64, device=dev)
    d_F = torch.ones(9, dtype=torch.float64, device=dev)
    d_mask = torch.full((cap,), 9, dtype=torch.uint8, device=dev)
    d_ninl = torch.full((1,), -1, dtype=torch.int32, device=dev)
    for path, form in ((2, 2), (2, 1), (1, 0)):   # one-launch kernel: LDS form (default), register form; per-lane kernels
        ctx.set_option(pm.api.PM_OPT_RANSAC_PATH, path)
        ctx.set_option(pm.api.PM_OPT_RANSAC_FORM, form)
        try:
            for _ in range(2):          # twice: the arrival ticket must be back at zero after a launch
                ctx.ransac_run_dev(d1.data_ptr(), d2.data_ptr(), cap, dn.data_ptr(), 0, H, 1.0, 0x5EED, d_key.data_ptr(),
                                   d_F.data_ptr(), d_mask.data_ptr(), d_ninl.data_ptr())
            ctx.synchronize()
        finally:
            ctx.set_option(pm.api.PM_OPT_RANSAC_PATH, 0)
            ctx.set_option(pm.api.PM_OPT_RANSAC_FORM, 0)
        rc, F_o, mask_o, ninl_o, key_o = oracle.ransac_fundamental(x1, x2, H, 1.0, 0x5EED, nthreads=8)
        m = d_mask.cpu().numpy()
        assert int(d_key.cpu().numpy().view(np.uint64)[0]) == key_o, path
        assert (d_F.cpu().numpy().view(np.uint64) == F_o.reshape(9).view(np.uint64)).all(), path
        assert (m[:n] == mask_o).all() and not m[n:].any(), path
        assert int(d_ninl.item()) == ninl_o, path


def test_ids_per_workgroup_option_same_bits(ctx, oracle):
    """PM_OPT_RANSAC_WG_IDS (the throughput form a pm_batch lane uses: full solver waves, few workgroups): same result."""
    n, H = 1500, 2048
    x1, x2, _, _ = synth.two_view(n, seed=5, outlier_frac=0.35, noise_px=0.6)
    want = oracle.ransac_fundamental(x1, x2, H, 1.0, 0xC5, nthreads=8)
    try:
        for form in (2, 1):
            ctx.set_option(pm.api.PM_OPT_RANSAC_FORM, form)
            for ids in (0, 1, 7, 64, 100, 128):
                ctx.set_option(pm.api.PM_OPT_RANSAC_WG_IDS, ids)
                got = ctx.ransac_fundamental(x1, x2, H, 1.0, 0xC5)
                assert got[4] == want[4] and got[3] == want[3] and (got[2] == want[2]).all(), (form, ids)
                assert (got[1].view(np.uint64) == want[1].view(np.uint64)).all(), (form, ids)
    finally:
        ctx.set_option(pm.api.PM_OPT_RANSAC_FORM, 0)
        ctx.set_option(pm.api.PM_OPT_RANSAC_WG_IDS, 0)


def test_many_hypotheses_and_every_register_depth(ctx, oracle):
    """100 000 ids (four rounds of workgroups, ids beyond 2^31) on each point-slot depth of the kernel."""
    for n in (900, 2400, 5000, 9175):
        x1, x2, _, _ = synth.two_view(n, seed=n, outlier_frac=0.4, noise_px=0.7)
        hb = 2 ** 31 - 50000
        got = ctx.ransac_fundamental(x1, x2, hb + 100000, 1.0, 0xC4, hyp_begin=hb)
        want = oracle.ransac_fundamental(x1, x2, hb + 100000, 1.0, 0xC4, hyp_begin=hb, nthreads=16)
        assert got[4] == want[4] and got[3] == want[3], n
        assert (got[2] == want[2]).all(), n
        assert (got[1].view(np.uint64) == want[1].view(np.uint64)).all(), n


def test_automatic_path_choice_crosses_to_the_per_lane_kernels(ctx, oracle):
    """ids x correspondences >= 2^30: the automatic rule hands the run to the hypothesis-per-lane kernels, whose own
    automatic choice then takes the scalar-operand scorer (BASELINE config C4's shape on one GPU); just below the rule
    the one-launch kernel runs.  Same bits either way."""
    assert ctx.get_option(pm.api.PM_OPT_RANSAC_PATH) == 0 and ctx.get_option(pm.api.PM_OPT_SCORE_OPERANDS) == 0
    for n, H in ((11000, 100000), (10000, 100000)):
        x1, x2, _, _ = synth.two_view(n, seed=n, outlier_frac=0.35, noise_px=0.6)
        ctx.timing_enable(True)
        ctx.timing_reset()
        got = ctx.ransac_fundamental(x1, x2, H, 1.0, 0xC4)
        used_fused = ctx.timing_get("ransac_fused")[1] > 0
        used_lane = ctx.timing_get("ransac_score")[1] > 0
        ctx.timing_enable(False)
        assert used_fused == (n * H < 2 ** 30) and used_lane == (n * H >= 2 ** 30), (n, H, used_fused, used_lane)
        want = oracle.ransac_fundamental(x1, x2, H, 1.0, 0xC4, nthreads=16)
        assert got[4] == want[4] and got[3] == want[3] and (got[2] == want[2]).all(), n
        assert (got[1].view(np.uint64) == want[1].view(np.uint64)).all(), n


def test_argument_checks_of_the_view_entry_points(ctx):
    dev = torch.device("cuda", 0)
    g = torch.zeros((2, 64, 2), dtype=torch.float32, device=dev)
    cnt = torch.zeros(2, dtype=torch.int32, device=dev)
    rec = torch.zeros(10, dtype=torch.float64, device=dev)
    msk = torch.zeros(128, dtype=torch.uint8, device=dev)
    good = PointsView(g.data_ptr(), g.data_ptr(), cnt.data_ptr(), 2, 64, 128, 1, 0)
    with pytest.raises(pm.PmError):                      # more than PM_MAX_PARTS parts
        ctx.ransac_shard_parts_dev(PointsView(g.data_ptr(), g.data_ptr(), cnt.data_ptr(), 65, 1, 2, 1, 0), 0, 10, 1.0, 1, rec.data_ptr())
    with pytest.raises(pm.PmError):                      # pitch smaller than a part
        ctx.ransac_shard_parts_dev(PointsView(g.data_ptr(), g.data_ptr(), cnt.data_ptr(), 2, 64, 100, 1, 0), 0, 10, 1.0, 1, rec.data_ptr())
    with pytest.raises(pm.PmError):                      # empty hypothesis range
        ctx.ransac_shard_parts_dev(good, 5, 5, 1.0, 1, rec.data_ptr())
    with pytest.raises(pm.PmError):                      # unknown error kind
        ctx.ransac_shard_parts_dev(good, 0, 10, 1.0, 1, rec.data_ptr(), kind=7)
    with pytest.raises(pm.PmError):                      # no records
        ctx.ransac_finish_parts_dev(good, 1.0, rec.data_ptr(), 0, 0, 0, msk.data_ptr(), 128, 0)
    # an empty view (all counts zero) is data, not an error: zero record, zero mask
    ctx.ransac_shard_parts_dev(good, 0, 10, 1.0, 1, rec.data_ptr())
    ctx.ransac_finish_parts_dev(good, 1.0, rec.data_ptr(), 1, 0, 0, msk.data_ptr(), 128, 0)
    ctx.synchronize()
    assert not rec.cpu().numpy().any() and not msk.cpu().numpy().any()
