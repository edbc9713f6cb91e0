"""CPU: `python bench.py --gpus N` (no launcher) must start N fresh ranks itself — before anything touches HIP — through
torch.distributed.run on 127.0.0.1, relay their status, and never exec.  Reference slot: none (harness contract)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gpus_flag_spawns_n_ranks_without_touching_the_gpu(monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    calls = {}

    def fake_call(cmd, env=None):
        calls["cmd"], calls["env"] = cmd, env
        return 7
    import subprocess
    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    modules_before = set(sys.modules)
    try:
        bench.main()
        raise AssertionError("launch_ranks must exit with the children's status")
    except SystemExit as e:
        assert e.code == 7
    cmd = calls["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    assert cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"] and cmd[-7].endswith("bench.py")
    assert calls["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert "points_matching_amd" not in set(sys.modules) - modules_before      # the parent never loaded the library


def test_rank_process_does_not_respawn(monkeypatch):
    """With WORLD_SIZE set (we ARE a rank) the launcher branch is skipped: on this GPU-less container the rank then
    stops at the explicit 'needs a GPU' error instead of spawning anything."""
    sys.path.insert(0, ROOT)
    import bench
    import subprocess
    monkeypatch.setattr(subprocess, "call", lambda *a, **k: (_ for _ in ()).throw(AssertionError("spawned")))
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2"])
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setenv("RANK", "0")
    saved = os.dup(1)
    try:
        bench.main()
        raise AssertionError("expected SystemExit")
    except SystemExit as e:
        assert "needs a GPU" in str(e.code)
    finally:
        os.dup2(saved, 1)
        os.close(saved)
