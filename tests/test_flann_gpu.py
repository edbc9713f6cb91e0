"""GPU: the FlannBasedMatcher-compatible approximate matcher (csrc/flann.hip; main.cpp:44, the reference's active
matcher object; SURVEY.md 8f-4, docs/SPEC.md S17).  Parity against OpenCV's FLANN is unpinned twice over (the library
is absent and its trees come from C rand()), so the tests check what can be checked:
  * the forest is a valid kd-forest of the train set (every tree: each point in exactly one leaf, every leaf on the
    side of each cut its coordinates say) and is reproducible from the seed;
  * the HIP search equals an independently written CPU search (oracle) over the same forest, bit for bit;
  * recall against the exact matcher, with the numbers printed; more checks never hurt recall;
  * with checks >= the train size the search is exhaustive and equals the exact matcher."""
import numpy as np
import pytest

import points_matching_amd as pm
from points_matching_amd import synth
from util import assert_matches_equal

pytestmark = pytest.mark.gpu


def _check_forest(nodes, roots, train):
    nt = train.shape[0]
    for root in roots:
        seen = np.zeros(nt, bool)
        stack = [(int(root), [])]            # (node, constraints so far: (dim, value, side))
        while stack:
            i, cons = stack.pop()
            nd = nodes[i]
            if nd["child1"] < 0:
                p = int(nd["divfeat"])
                assert not seen[p]
                seen[p] = True
                continue
            stack.append((int(nd["child1"]), None))
            stack.append((int(nd["child2"]), None))
        assert seen.all()
    assert nodes.shape[0] == len(roots) * (2 * nt - 1)


def _leaf_sets(nodes, i):
    """points under node i"""
    out, st = [], [int(i)]
    while st:
        j = st.pop()
        nd = nodes[j]
        if nd["child1"] < 0:
            out.append(int(nd["divfeat"]))
        else:
            st += [int(nd["child1"]), int(nd["child2"])]
    return out


def test_forest_is_a_valid_reproducible_kd_forest(ctx):
    w = synth.pair_workload(64, 700, 128, seed=3, kind="surf")
    ix = pm.api.FlannIndex(ctx, w["t"], trees=4, checks=32, seed=11)
    nodes, roots = ix.export()
    _check_forest(nodes, roots, w["t"])
    # cut consistency on the top levels of every tree: left subtree values <= cut <= right subtree values
    for root in roots:
        nd = nodes[root]
        left, right = _leaf_sets(nodes, nd["child1"]), _leaf_sets(nodes, nd["child2"])
        d, v = int(nd["divfeat"]), float(nd["divval"])
        assert (w["t"][left, d] <= v).all() and (w["t"][right, d] >= v).all()
        assert 0.2 < len(left) / 700 < 0.8                      # mean split of a unimodal coordinate: roughly balanced
    again = pm.api.FlannIndex(ctx, w["t"], trees=4, checks=32, seed=11).export()
    other = pm.api.FlannIndex(ctx, w["t"], trees=4, checks=32, seed=12).export()
    assert (again[0] == nodes).all() and (again[1] == roots).all()
    assert not (other[0] == nodes).all()
    assert len({int(nodes[r]["divfeat"]) for r in roots}) >= 1   # trees are drawn independently


@pytest.mark.parametrize("kind,nq,nt,dim,k", [("sift", 1000, 3000, 128, 1), ("surf", 700, 2500, 128, 2), ("surf", 300, 17, 64, 2),
                                              ("sift", 64, 1, 128, 1), ("surf", 513, 4097, 36, 4)])
def test_hip_search_equals_the_cpu_search_over_the_same_forest(ctx, oracle, kind, nq, nt, dim, k):
    w = synth.pair_workload(nq, nt, dim, seed=nq + nt, planted=0.5, kind=kind)
    ix = pm.api.FlannIndex(ctx, w["t"], seed=5)
    got = ix.knn(w["q"], k)
    nodes, roots = ix.export()
    want = oracle.flann_search(nodes, roots, w["t"], w["q"], k, 32)
    assert_matches_equal(got, want, "HIP kd-forest search vs CPU search")


def test_recall_against_the_exact_matcher(ctx, oracle, capsys):
    """The reference's sizes (SURF with hessianThreshold 8000: 10^2..10^3 keypoints) and BASELINE's 8k.  Recall is
    reported for all queries and for the TRUE matches (queries planted as noisy copies of a train row: the ones a
    matcher exists for).  Queries without a counterpart are uniform points in 128-D, where no tree search finds the
    nearest of thousands of equidistant rows in 32 checks — neither does FLANN."""
    rows = []
    for kind, n in (("surf", 300), ("surf", 1000), ("sift", 1000), ("sift", 8192)):
        w = synth.pair_workload(n, n, 128, seed=n, planted=0.5, kind=kind)
        exact = ctx.bf_knn_l2(w["q"], w["t"], 1)[:, 0]
        true = w["truth"] >= 0
        assert (exact["trainIdx"][true] == w["truth"][true]).mean() > 0.97         # the exact matcher finds the planted row
        rec, rec_true = {}, {}
        for checks in (8, 32, 128):
            ix = pm.api.FlannIndex(ctx, w["t"], trees=4, checks=checks, seed=1)
            got = ix.knn(w["q"], 1)[:, 0]
            hit = got["trainIdx"] == exact["trainIdx"]
            rec[checks], rec_true[checks] = float(hit.mean()), float(hit[true].mean())
            assert (got["distance"][hit].view(np.uint32) == exact["distance"][hit].view(np.uint32)).all()   # same bits when found
            assert (got["distance"][~hit] >= exact["distance"][~hit]).all()                                   # never better than exact
            ix.close()
        rows.append((kind, n, rec, rec_true))
        assert rec[8] <= rec[32] + 0.02 and rec[32] <= rec[128] + 0.02
        assert rec_true[32] > 0.4 and rec_true[128] > rec_true[32] - 0.02 and rec_true[128] > 0.7
    with capsys.disabled():
        for kind, n, rec, rt in rows:
            print("\nflann recall@1 %s %dx%d: checks 8/32/128 = %.3f / %.3f / %.3f   true matches only: %.3f / %.3f / %.3f"
                  % (kind, n, n, rec[8], rec[32], rec[128], rt[8], rt[32], rt[128]))


def test_exhaustive_checks_equal_the_exact_matcher(ctx):
    w = synth.pair_workload(400, 350, 128, seed=9, planted=0.5, kind="surf")
    ix = pm.api.FlannIndex(ctx, w["t"], trees=4, checks=100000, seed=2)
    got = ix.knn(w["q"], 2)
    exact = ctx.bf_knn_l2(w["q"], w["t"], 2)
    # exhaustive search examines every point: same neighbours and distances (FLANN keeps an EARLIER-examined point ahead
    # of an equal distance, the exact matcher the lower index: compare as sets of (distance, index) rows where they tie)
    same = (got["trainIdx"] == exact["trainIdx"]).all(axis=1)
    assert same.mean() > 0.99
    assert (got["distance"].view(np.uint32) == exact["distance"].view(np.uint32)).all()


def test_flann_argument_checks(ctx):
    w = synth.pair_workload(16, 40, 32, seed=1, kind="surf")
    with pytest.raises(pm.PmError):
        pm.api.FlannIndex(ctx, w["t"][:0])                          # empty train set
    with pytest.raises(pm.PmError):
        pm.api.FlannIndex(ctx, w["t"], trees=17)
    ix = pm.api.FlannIndex(ctx, w["t"])
    with pytest.raises(pm.PmError):
        ix.knn(w["q"], 5)                                           # k > 4
    assert ix.knn(w["q"][:0], 1).shape == (0, 1)
    one = pm.api.FlannIndex(ctx, w["t"][:1])                        # a single train row: every query matches it, no 2nd neighbour
    r = one.knn(w["q"], 2)
    assert (r["trainIdx"][:, 0] == 0).all() and (r["trainIdx"][:, 1] == -1).all() and np.isinf(r["distance"][:, 1]).all()
