"""points_matching_amd — MI355X-native two-view point matcher (hot path of
wenxiaoshuai/Points-Matching: BF descriptor matching + filter + RANSAC-F + residual report).

Product = libpm_hip.so (hand-written HIP kernels for gfx950 behind the C ABI of include/pm.h)
plus the C++ host in host/.  This Python package is the thin ctypes mirror used by tests and
bench.py.  It never falls back to a CPU implementation.
"""
from . import api, synth  # noqa: F401
from .api import Context, PmError, MATCH_DTYPE  # noqa: F401

__all__ = ["api", "synth", "Context", "PmError", "MATCH_DTYPE"]
