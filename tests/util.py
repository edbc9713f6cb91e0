import numpy as np


def assert_matches_equal(got, want, what=""):
    """Bit-exact comparison of two pm_match arrays (indices AND distance bits)."""
    assert got.shape == want.shape, (what, got.shape, want.shape)
    for f in ("queryIdx", "trainIdx", "imgIdx"):
        bad = np.nonzero(got[f] != want[f])
        assert bad[0].size == 0, "%s: %s differs at %s: got %s want %s" % (
            what, f, [b[:5] for b in bad], got[f][bad][:5], want[f][bad][:5])
    gb = got["distance"].view(np.uint32)
    wb = want["distance"].view(np.uint32)
    bad = np.nonzero(gb != wb)
    assert bad[0].size == 0, "%s: distance bits differ at %s: got %s want %s" % (
        what, [b[:5] for b in bad], got["distance"][bad][:5], want["distance"][bad][:5])
