"""GPU: the C++ host tool (points_matching_amd/host/pm_cli.cpp, counterpart of the reference's
main()) prints the reference's stdout surface (main.cpp:58-59, :73, :76, :119, :123) with the
values the oracle computes."""
import json
import os
import subprocess

import numpy as np
import pytest

from points_matching_amd import build, io, synth

pytestmark = pytest.mark.gpu


def _g(x):
    return "%g" % x          # iostream default formatting (precision 6)


@pytest.mark.parametrize("mode", ["midpoint", "ratio"])
def test_cli_stdout_matches_reference_format(tmp_path, oracle, mode):
    exe = build.HOST_BIN
    assert os.path.exists(exe), "run python -m points_matching_amd.build"
    w = synth.pair_workload(nq=300, nt=280, dim=128, seed=77, planted=0.5, kind="surf")
    paths = {}
    for name in ("q", "t", "kp1", "kp2"):
        paths[name] = str(tmp_path / (name + ".pmm"))
        io.save_pmm(paths[name], w[name])
    cmd = [exe, "--desc1", paths["q"], "--desc2", paths["t"], "--kp1", paths["kp1"], "--kp2", paths["kp2"],
           "--filter", mode, "--method", "ransac8", "--iters", "400", "--thresh", "1.0", "--seed", "99", "--json"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.splitlines()

    if mode == "midpoint":
        knn = oracle.bf_knn_l2(w["q"], w["t"], 1).reshape(-1)
        good, mn, mx = oracle.filter_midpoint(knn)
        exp = ["The Best Match is? " + _g(mn), "The Worst Match is? " + _g(mx)]
    else:
        good = oracle.filter_ratio(oracle.bf_knn_l2(w["q"], w["t"], 2), 0.8)
        exp = []
    exp += oracle.format_match_list(good).splitlines()
    xy1 = oracle.gather_points(w["kp1"], good["queryIdx"])
    xy2 = oracle.gather_points(w["kp2"], good["trainIdx"])
    rc, F, mask, ninl, key = oracle.ransac_fundamental(xy1, xy2, 400, 1.0, 99)
    F = oracle.f_scale_f33(F)
    r, mean = oracle.epipolar_residuals(xy1, xy2, F, 1)
    exp += ["result = %d %s" % (i, _g(r[i])) for i in range(good.size)]
    exp += ["The average value is  " + _g(mean)]
    assert lines[:-1] == exp
    js = json.loads(lines[-1])
    assert js["matches"] == good.size and js["inliers"] == ninl
    assert js["best_hyp"] == 0xFFFFFFFF - (key & 0xFFFFFFFF)
    assert np.array_equal(np.array(js["F"]), F.reshape(9))


def test_cli_knn_hint_same_output(tmp_path):
    """--knn-hint auto | int | u8 on SIFT-like (u8-valued) descriptor files: the hint picks the coarse route, never the result."""
    exe = build.HOST_BIN
    w = synth.pair_workload(nq=700, nt=650, dim=128, seed=5, planted=0.5, kind="sift")
    paths = {}
    for name in ("q", "t", "kp1", "kp2"):
        paths[name] = str(tmp_path / (name + ".pmm"))
        io.save_pmm(paths[name], w[name])
    outs = []
    for hint in ("auto", "int", "u8"):
        cmd = [exe, "--desc1", paths["q"], "--desc2", paths["t"], "--kp1", paths["kp1"], "--kp2", paths["kp2"], "--filter", "ratio",
               "--method", "ransac8", "--iters", "300", "--seed", "7", "--knn-hint", hint]
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
        assert out.returncode == 0, out.stderr
        outs.append(out.stdout)
    assert outs[0] == outs[1] == outs[2] and outs[0].count("result = ") > 100
    # ... and "unit" on unit-norm (SURF-like) rows
    w = synth.pair_workload(nq=500, nt=450, dim=64, seed=6, planted=0.5, kind="surf")
    for name in ("q", "t", "kp1", "kp2"):
        io.save_pmm(paths[name], w[name])
    outs = []
    for hint in ("auto", "unit"):
        cmd = [exe, "--desc1", paths["q"], "--desc2", paths["t"], "--kp1", paths["kp1"], "--kp2", paths["kp2"], "--filter", "ratio",
               "--method", "ransac8", "--iters", "300", "--seed", "7", "--knn-hint", hint]
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
        assert out.returncode == 0, out.stderr
        outs.append(out.stdout)
    assert outs[0] == outs[1] and outs[0].count("result = ") > 50
    bad = subprocess.run([exe, "--desc1", paths["q"], "--desc2", paths["t"], "--kp1", paths["kp1"], "--kp2", paths["kp2"], "--knn-hint", "x"],
                         capture_output=True, text=True, timeout=60)
    assert bad.returncode == 2


def test_cli_7point_lmeds_method(tmp_path, oracle):
    """--method 7point-lmeds: what the reference's CV_FM_7POINT call selects (main.cpp:95-98);
    default iteration count = OpenCV's 300."""
    exe = build.HOST_BIN
    w = synth.pair_workload(nq=400, nt=380, dim=128, seed=78, planted=0.5, kind="sift")
    paths = {}
    for name in ("q", "t", "kp1", "kp2"):
        paths[name] = str(tmp_path / (name + ".pmm"))
        io.save_pmm(paths[name], w[name])
    cmd = [exe, "--desc1", paths["q"], "--desc2", paths["t"], "--kp1", paths["kp1"], "--kp2", paths["kp2"],
           "--filter", "ratio", "--method", "7point-lmeds", "--seed", "5", "--json", "--quiet"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    js = json.loads(out.stdout.splitlines()[-1])
    good = oracle.filter_ratio(oracle.bf_knn_l2(w["q"], w["t"], 2), 0.8)
    xy1 = oracle.gather_points(w["kp1"], good["queryIdx"])
    xy2 = oracle.gather_points(w["kp2"], good["trainIdx"])
    rc, F, mask, ninl, best, med = oracle.lmeds_fundamental(xy1, xy2, 300, 5, nthreads=4)
    assert rc == 0 and js["matches"] == good.size and js["inliers"] == ninl and js["best_hyp"] == best
    assert np.array_equal(np.array(js["F"]), oracle.f_scale_f33(F).reshape(9))


def _write_inputs(tmp_path, w):
    paths = {}
    for name in ("q", "t", "kp1", "kp2"):
        paths[name] = str(tmp_path / (name + ".pmm"))
        io.save_pmm(paths[name], w[name])
    return [build.HOST_BIN, "--desc1", paths["q"], "--desc2", paths["t"], "--kp1", paths["kp1"], "--kp2", paths["kp2"]]


def test_cli_epilines_print_and_overlay(tmp_path, oracle):
    """main.cpp:127-142: computeCorrespondEpilines(selPoints1, 1, F) and the cv::line end points, printed and drawn."""
    w = synth.pair_workload(nq=260, nt=240, dim=128, seed=12, planted=0.5, kind="sift")
    ppm = str(tmp_path / "epi.ppm")
    cmd = _write_inputs(tmp_path, w) + ["--filter", "ratio", "--method", "ransac8", "--iters", "300", "--seed", "3", "--quiet",
                                        "--print-epilines", "--epilines", ppm, "--canvas", "993", "660"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    good = oracle.filter_ratio(oracle.bf_knn_l2(w["q"], w["t"], 2), 0.8)
    xy1 = oracle.gather_points(w["kp1"], good["queryIdx"])
    xy2 = oracle.gather_points(w["kp2"], good["trainIdx"])
    rc, F, _, _, _ = oracle.ransac_fundamental(xy1, xy2, 300, 1.0, 3)
    F = oracle.f_scale_f33(F)
    lines = oracle.epilines(xy1, 1, F)
    ends = oracle.epiline_endpoints(lines, 993)
    exp = ["epiline = %d %s %s %s  (%d, %d) -> (%d, %d)" % (i, _g(lines[i, 0]), _g(lines[i, 1]), _g(lines[i, 2]),
                                                             ends[i, 0], ends[i, 1], ends[i, 2], ends[i, 3])
           for i in range(good.size)]
    assert out.stdout.splitlines() == exp
    raw = open(ppm, "rb").read()
    assert raw.startswith(b"P6\n993 660\n255\n")
    px = np.frombuffer(raw[len(b"P6\n993 660\n255\n"):], np.uint8).reshape(660, 993, 3)
    white = (px == 255).all(axis=2)
    assert white.sum() > 993                      # at least one full-width line was drawn
    # a drawn line passes through the pixel its own equation gives at mid-width
    i = next(k for k in range(good.size) if 0 <= ends[k, 1] < 660 and 0 <= ends[k, 3] < 660)
    ymid = int(round(-(lines[i, 2] + lines[i, 0] * 496) / lines[i, 1]))
    assert white[max(0, ymid - 1):ymid + 2, 496].any()


def test_cli_mgpu_path_equals_single_device_path(tmp_path):
    """--gpus N runs through pm_mgpu_match_ransac (RCCL behind the C ABI); with one device it must print exactly
    what the single-context path prints."""
    w = synth.pair_workload(nq=700, nt=650, dim=128, seed=5, planted=0.5, kind="sift")
    base = _write_inputs(tmp_path, w) + ["--filter", "ratio", "--method", "ransac8", "--iters", "500", "--seed", "11", "--json"]
    a = subprocess.run(base, capture_output=True, text=True, timeout=120)
    b = subprocess.run(base + ["--gpus", "1", "--mgpu"], capture_output=True, text=True, timeout=180)
    assert a.returncode == 0 and b.returncode == 0, (a.stderr, b.stderr)
    la, lb = a.stdout.splitlines(), b.stdout.splitlines()
    assert la[:-1] == lb[:-1]
    ja, jb = json.loads(la[-1]), json.loads(lb[-1])
    for k in ("matches", "inliers", "best_hyp", "F", "mean_abs_x1Fx2"):
        assert ja[k] == jb[k], k


def test_cli_flann_matcher_is_the_reference_literal_flow(tmp_path, ctx, oracle):
    """--matcher flann --filter midpoint --method 7point-lmeds = what main.cpp:44-98 literally does (FlannBasedMatcher,
    midpoint filter, CV_FM_7POINT).  The printed list is the midpoint filter of the kd-forest's 1-NN matches."""
    import points_matching_amd as pm
    w = synth.pair_workload(nq=400, nt=380, dim=128, seed=21, planted=0.6, kind="surf")
    cmd = _write_inputs(tmp_path, w) + ["--matcher", "flann", "--seed", "9", "--json"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.splitlines()
    ix = pm.api.FlannIndex(ctx, w["t"], trees=4, checks=32, seed=9)
    knn = ix.knn(w["q"], 1).reshape(-1)
    good, mn, mx = oracle.filter_midpoint(knn)
    exp = ["The Best Match is? " + _g(mn), "The Worst Match is? " + _g(mx)] + oracle.format_match_list(good).splitlines()
    assert lines[:len(exp)] == exp
    js = json.loads(lines[-1])
    assert js["matches"] == good.size


def test_cli_image_pair_in(tmp_path):
    """Image pair in, match list + F + epiline overlay out (the reference's whole surface, main.cpp:14-143) on the
    reference's own two photographs (half-resolution PGM fixtures)."""
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    ppm = str(tmp_path / "epi.ppm")
    cmd = [build.HOST_BIN, "--img1", os.path.join(gold, "img01_half.pgm"), "--img2", os.path.join(gold, "img02_half.pgm"),
           "--filter", "ratio", "--method", "ransac8", "--iters", "5000", "--json", "--epilines", ppm]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.splitlines()
    js = json.loads(lines[-1])
    assert lines[0] == "Good Matches are:" and js["matches"] >= 60 and js["inliers"] >= 0.6 * js["matches"]
    assert js["mean_abs_x2Fx1"] < 5.0
    raw = open(ppm, "rb").read()
    assert raw.startswith(b"P6\n496 330\n255\n")
    px = np.frombuffer(raw[len(b"P6\n496 330\n255\n"):], np.uint8).reshape(330, 496, 3)
    assert (px == 255).all(axis=2).sum() > 496                      # epipolar lines drawn over the right image
    assert len(np.unique(px[..., 0])) > 50                           # ... which is still there under them
    # the reference's literal configuration (FlannBasedMatcher, midpoint filter, CV_FM_7POINT) also runs end to end
    out2 = subprocess.run(cmd[:5] + ["--matcher", "flann", "--json"], capture_output=True, text=True, timeout=300)
    assert out2.returncode == 0, out2.stderr
    js2 = json.loads(out2.stdout.splitlines()[-1])
    assert js2["matches"] >= 8


def test_cli_image_pair_pinned_to_its_fixture(tmp_path):
    """tests/golden/img_half_cli.npz (made by tests/golden/make_img_cli_fixture.py): the keypoints / u8 descriptors the C++
    front end finds in the two half-resolution photographs and the ORACLE's match list, inlier mask, winning hypothesis
    and F for them.  (1) pm_cli on the fixture's descriptors must reproduce every one of those bits through the HIP path;
    (2) pm_cli --img1/--img2 must find the same features again on this machine (another CPU's libm may move a descriptor
    level: the comparison of (2) allows that, (1) allows nothing)."""
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    g = np.load(os.path.join(gold, "img_half_cli.npz"), allow_pickle=False)
    iters, seed = int(g["params"][0]), int(g["params"][1])
    p = {}
    for name, arr in (("d1", g["desc1"].astype(np.float32)), ("d2", g["desc2"].astype(np.float32)), ("k1", g["kp1"]), ("k2", g["kp2"])):
        p[name] = str(tmp_path / (name + ".pmm"))
        io.save_pmm(p[name], arr)
    base = ["--filter", "ratio", "--ratio", "0.8", "--method", "ransac8", "--iters", str(iters), "--seed", str(seed), "--thresh", "1.0",
            "--f-scale", "unit", "--json"]
    out = subprocess.run([build.HOST_BIN, "--desc1", p["d1"], "--desc2", p["d2"], "--kp1", p["k1"], "--kp2", p["k2"]] + base,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.splitlines()
    n = g["ratio_query"].size
    want = ["Good Matches are:"] + ["-- Good Match [%d] Keypoint 1: %d  -- Keypoint 2: %d  " % (i, g["ratio_query"][i], g["ratio_train"][i])
                                    for i in range(n)]
    assert lines[:n + 1] == want
    js = json.loads(lines[-1])
    key = int(g["key"][0])
    assert js["matches"] == n and js["inliers"] == int(g["mask"].sum()) == key >> 32
    assert js["best_hyp"] == 0xFFFFFFFF - (key & 0xFFFFFFFF) and js["ransac_status"] == 0
    assert (np.array(js["F"], np.float64).view(np.uint64) == g["F_bits"]).all()
    # (2) image pair in
    pre = str(tmp_path / "f")
    out2 = subprocess.run([build.HOST_BIN, "--img1", os.path.join(gold, "img01_half.pgm"), "--img2", os.path.join(gold, "img02_half.pgm"),
                           "--save-features", pre] + base, capture_output=True, text=True, timeout=300)
    assert out2.returncode == 0, out2.stderr
    d1, d2 = io.load_pmm(pre + "_desc1.pmm"), io.load_pmm(pre + "_desc2.pmm")
    k1 = io.load_pmm(pre + "_kp1.pmm")
    assert d1.shape == g["desc1"].shape and d2.shape == g["desc2"].shape and np.abs(k1 - g["kp1"]).max() < 1e-3
    assert (np.abs(d1.astype(np.int32) - g["desc1"].astype(np.int32)) <= 1).all() and (d1 == g["desc1"]).mean() > 0.999
    js2 = json.loads(out2.stdout.splitlines()[-1])
    if (d1 == g["desc1"]).all() and (d2 == g["desc2"]).all():
        assert out2.stdout.splitlines()[:n + 1] == want and js2["inliers"] == js["inliers"] and js2["F"] == js["F"]
    else:
        assert abs(js2["matches"] - n) <= 3 and js2["inliers"] >= 0.9 * js["inliers"]
