"""GPU: bench.py's N > 1 code path with two ranks sharing this box's one GPU (gloo backend, --single-device): the
launcher branch, both exchanges (survivor blocks, 80-byte records) through the HIP entry points, weak and strong
scaling, checked by the bench's own oracle parity flag.  The driver's real multi-GPU run uses RCCL on N GPUs."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--single-device",
           "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--sustain-seconds", "0"] + extra
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_two_ranks_c3_shape(scaling):
    d = _run(["--nq", "2048", "--nt", "2048", "--hyps", "3000", "--scaling", scaling])
    assert d["n_gpus"] == 2 and d["parity"] == "ok" and d["scaling"] == scaling
    assert d["ransac"]["n_matches"] > 300 and d["ransac"]["inliers"] > 100
    assert "ransac_fused" in d["kernels_us"] and "ransac_finish" in d["kernels_us"]
    # the form of the timed region is the one the warm-up trial found faster (max over ranks); both are reported
    tr = d["step_form_trial"]
    piped = tr["pipelined_ms_per_step"] < tr["serial_ms_per_step"]
    assert d["step_form"].startswith("pipelined" if piped else "serial")
    assert d["ms_per_step"] == (d["ms_per_step_pipelined"] if piped else d["ms_per_step_serial"])
    assert d["ms_per_step_serial"] > 0 and d["ms_per_step_pipelined"] > 0
    assert "timing_note" in d["config"]                   # two processes on one GPU: not performance data
    assert d["collectives"]["ranks"] == 2 and d["collectives"]["allgather_records_us"] > 0


def test_two_ranks_c4_strong_shape():
    d = _run(["--workload", "c4", "--scaling", "strong", "--nq", "4096", "--nt", "4096", "--hyps", "5000"])
    assert d["n_gpus"] == 2 and d["parity"] == "ok" and d["scaling"] == "strong"
