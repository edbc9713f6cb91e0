"""Builds libpm_hip.so (the C-ABI library of include/pm.h) for gfx950 with hipcc, in-tree.

    python -m points_matching_amd.build [--force] [--verbose]

One object per translation unit, then one shared library next to this file.  Flags that matter
for parity: -ffp-contract=off (the only fused multiply-adds are the explicit fma()/fmaf() calls
the spec names) and hipcc's default correctly-rounded fp32 divide/sqrt, which must stay on.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libpm_hip.so")
SOURCES = ["pm_capi.cpp", "knn_l2.hip", "knn_coarse.hip", "knn_hamming.hip", "ransac.hip", "ransac_fused.hip", "ransac_shard.hip", "filter_gather.hip",
           "pair_batch.cpp", "lmeds.hip", "mgpu.cpp", "flann.hip"]
# per-file extra flags: the coarse kernels only nominate candidates (no result bit depends on them)
EXTRA = {"knn_coarse.hip": ["-ffinite-math-only"], "mgpu.cpp": ["-pthread"]}
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-fhip-fp32-correctly-rounded-divide-sqrt", "-Wall", "-Wno-unused-function",
         "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-I/opt/rocm/include"]


def _newer(a, b):
    return (not os.path.exists(b)) or os.path.getmtime(a) > os.path.getmtime(b)


def _deps():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".h"))]
    hdrs.append(os.path.join(ROOT, "include", "pm.h"))
    hdrs.append(os.path.abspath(__file__))
    return hdrs


def build(force=False, verbose=False, extra_flags=()):
    os.makedirs(OBJ, exist_ok=True)
    deps = _deps()
    jobs = []
    objs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.rsplit(".", 1)[0] + ".o")
        objs.append(o)
        if force or _newer(s, o) or any(_newer(d, o) for d in deps):
            cmd = [HIPCC] + FLAGS + EXTRA.get(src, []) + list(extra_flags) + (["-x", "hip"] if src.endswith(".hip") else []) + \
                  ["-c", s, "-o", o]
            jobs.append(cmd)

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n%s\n%s" % (" ".join(cmd), r.stderr))
        if verbose and r.stderr.strip():
            print(r.stderr)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if force or jobs or not os.path.exists(LIB):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl", "-pthread"])
    return LIB


HOST_DIR = os.path.join(HERE, "host")
HOST_SRC = os.path.join(HOST_DIR, "pm_cli.cpp")
HOST_SRCS = [HOST_SRC, os.path.join(HOST_DIR, "pm_features.cpp")]
HOST_BIN = os.path.join(HOST_DIR, "pm_cli")


def build_host(force=False):
    """C++ host tool (the counterpart of the reference's main()); links libpm_hip.so."""
    if not os.path.exists(HOST_SRC):
        return None
    deps = HOST_SRCS + [os.path.join(HOST_DIR, "pm_features.hpp"), os.path.join(ROOT, "include", "pm.h"), LIB]
    if force or any(_newer(d, HOST_BIN) for d in deps):
        cmd = ["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"), "-I" + HOST_DIR] + HOST_SRCS + \
              ["-o", HOST_BIN, "-L" + HERE, "-lpm_hip", "-Wl,-rpath,$ORIGIN/..", "-Wl,-rpath," + HERE]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("host build failed:\n%s\n%s" % (" ".join(cmd), r.stderr))
    return HOST_BIN


def build_host_sanitized():
    """The host tool under AddressSanitizer + UBSan (tests/test_oracle_sanitized_cpu.py: the feature front-end is the
    host code with the most index arithmetic).  CPU only; never shipped."""
    out = os.path.join(HERE, "build", "pm_cli_san")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    deps = HOST_SRCS + [os.path.join(HOST_DIR, "pm_features.hpp"), os.path.join(ROOT, "include", "pm.h"), LIB]
    if any(_newer(d, out) for d in deps):
        cmd = ["g++", "-O1", "-g", "-std=c++17", "-ffp-contract=off", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
               "-fno-omit-frame-pointer", "-I" + os.path.join(ROOT, "include"), "-I" + HOST_DIR] + HOST_SRCS + \
              ["-o", out, "-L" + HERE, "-lpm_hip", "-Wl,-rpath," + HERE]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("sanitized host build failed:\n%s\n%s" % (" ".join(cmd), r.stderr))
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv))
    print(build_host(force="--force" in sys.argv))
