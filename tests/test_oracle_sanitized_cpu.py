"""The CPU restatement under AddressSanitizer + UndefinedBehaviorSanitizer (the host-side sanitizer
target; GPU sanitizers are not available on the pool).  `make -C oracle sanitize` builds the same source with
-fsanitize=address,undefined -fno-sanitize-recover=all; the oracle's own CPU tests (golden vectors, edge cases, the
independent numpy checks) then run against that build in a child interpreter with libasan preloaded.  Any out-of-bounds
access, use of uninitialised stack through a misread size, signed overflow or misaligned access aborts the child."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _libasan():
    try:
        p = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True, timeout=30).stdout.strip()
    except (OSError, subprocess.SubprocessError):
        return None
    return os.path.realpath(p) if p and os.path.sep in p and os.path.exists(p) else None


def test_oracle_tests_pass_under_asan_and_ubsan():
    asan = _libasan()
    if asan is None:
        pytest.skip("gcc has no libasan.so here")
    build = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "sanitize"], capture_output=True, text=True, timeout=300)
    assert build.returncode == 0, build.stderr[-2000:]
    san = os.path.join(ROOT, "oracle", "build", "libpm_oracle_san.so")
    env = dict(os.environ)
    env.update({"PM_ORACLE_LIB": san, "LD_PRELOAD": asan, "OMP_NUM_THREADS": "2",
                # the interpreter itself is not instrumented: no leak report for it, no link-order check
                "ASAN_OPTIONS": "detect_leaks=0:verify_asan_link_order=0:abort_on_error=1:halt_on_error=1",
                "UBSAN_OPTIONS": "halt_on_error=1:print_stacktrace=1"})
    run = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider",
                          os.path.join(ROOT, "tests", "test_oracle_cpu.py"), os.path.join(ROOT, "tests", "test_golden.py"),
                          os.path.join(ROOT, "tests", "test_independent_cpu.py")],
                         capture_output=True, text=True, timeout=1500, cwd=ROOT, env=env)
    tail = (run.stdout[-3000:] + "\n" + run.stderr[-3000:])
    assert run.returncode == 0, tail
    assert "passed" in run.stdout and "AddressSanitizer" not in tail and "runtime error" not in tail, tail
    # ... and the child really loaded the sanitized build
    probe = subprocess.run([sys.executable, "-c", "from oracle import pm_oracle as O; O.lib(); print(open('/proc/self/maps').read().count('libpm_oracle_san'))"],
                           capture_output=True, text=True, timeout=120, cwd=ROOT, env=env)
    assert probe.returncode == 0 and int(probe.stdout.strip() or 0) > 0, probe.stderr[-1000:]


def test_feature_front_end_is_clean_under_asan_and_ubsan(tmp_path):
    """host/pm_features.cpp (detector + 128-D descriptor of pm_cli --img1/--img2: pyramids, 3-D extrema, histograms —
    the host code with the most index arithmetic) under the same sanitizers, on the two fixture images; the files it
    writes equal the regular build's byte for byte."""
    if _libasan() is None:
        pytest.skip("gcc has no libasan.so here")
    from points_matching_amd import build
    build.build()
    regular = build.build_host()
    san = build.build_host_sanitized()
    gold = os.path.join(ROOT, "tests", "golden")
    env = dict(os.environ)
    env.update({"ASAN_OPTIONS": "detect_leaks=0:abort_on_error=1:halt_on_error=1", "UBSAN_OPTIONS": "halt_on_error=1:print_stacktrace=1"})
    outs = {}
    for name, exe in (("regular", regular), ("san", san)):
        d = tmp_path / name
        d.mkdir()
        cmd = [exe, "--img1", os.path.join(gold, "img01_half.pgm"), "--img2", os.path.join(gold, "img02_half.pgm"),
               "--extract-only", "--save-features", str(d / "f"), "--quiet"]
        run = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
        assert run.returncode == 0, (name, run.stderr[-3000:])
        assert "AddressSanitizer" not in run.stderr and "runtime error" not in run.stderr, run.stderr[-3000:]
        outs[name] = {k: open(str(d / ("f_%s.pmm" % k)), "rb").read() for k in ("desc1", "desc2", "kp1", "kp2")}
    assert outs["regular"] == outs["san"]
