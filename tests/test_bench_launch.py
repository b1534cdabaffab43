"""bench.py's launch contract on a box without a GPU: `--gpus N` with no RANK in the environment must start N fresh
rank processes itself (the driver runs `python bench.py --gpus N`), each with the torch.distributed environment of its
rank, and must do so before anything touches the GPU; here the ranks then stop at "needs a HIP device"."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                          timeout=timeout)


def test_gpus2_spawns_two_ranks_without_torchrun():
    import torch
    if torch.cuda.is_available():
        pytest.skip("CPU-only check (on a GPU box the ranks would really run)")
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "1"])
    assert r.returncode == 1                                  # the ranks failed (no HIP device), loudly
    assert r.stderr.count("bench.py needs a HIP device") == 2, r.stderr      # ... and there were exactly two of them
    assert "a rank process failed" in r.stderr
    assert r.stdout.strip() == ""                             # no JSON line from a failed run


def test_world_size_mismatch_is_an_error_not_an_assert():
    r = _run(["--gpus", "4"], {"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "2"})
    assert r.returncode == 2 and "WORLD_SIZE=2 but --gpus 4" in r.stderr


def test_spawned_ranks_get_their_environment(tmp_path, monkeypatch):
    """spawn_ranks() hands every child RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / one shared MASTER_PORT."""
    sys.path.insert(0, ROOT)
    import bench
    seen = []

    class FakeProc:
        def __init__(self, cmd, env=None, stdout=None):
            seen.append((cmd, {k: env[k] for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}))

        def wait(self, timeout=None):
            return 0

        def poll(self):
            return 0

    monkeypatch.setattr(bench.subprocess, "Popen", FakeProc)
    monkeypatch.delenv("MASTER_PORT", raising=False)
    args = bench.parse_args(["--gpus", "4", "--steps", "3"])
    assert bench.spawn_ranks(args, ["--gpus", "4", "--steps", "3"]) == 0
    assert [e["RANK"] for _, e in seen] == ["0", "1", "2", "3"] and [e["LOCAL_RANK"] for _, e in seen] == ["0", "1", "2", "3"]
    assert all(e["WORLD_SIZE"] == "4" and e["MASTER_ADDR"] == "127.0.0.1" for _, e in seen)
    assert len({e["MASTER_PORT"] for _, e in seen}) == 1
    assert all(cmd[1].endswith("bench.py") and cmd[2:] == ["--gpus", "4", "--steps", "3"] for cmd, _ in seen)


def test_bytes_per_env_step_counts_the_depth_image():
    sys.path.insert(0, ROOT)
    import bench
    from gym_kmanip_amd.model import compile_model
    cm = compile_model("KManipSoloArm")
    assert bench.algorithmic_bytes_per_env_step(cm) == 1197
    assert bench.algorithmic_bytes_per_env_step(cm, 64) == 1197 + 64 * 64 * 4    # BASELINE config 5


def test_rank_path_refuses_a_collective_that_does_not_span_the_job(capsys):
    """bench.py's start-up self-check (run_rank, right after init_process_group): an all-reduce of ones must count WORLD_SIZE
    ranks; anything else is a loud non-zero exit instead of a whole-job value computed from part of the job."""
    sys.path.insert(0, ROOT)
    import bench
    assert bench.check_ranks_seen(8, 8, 0) == 0 and capsys.readouterr().err == ""
    assert bench.check_ranks_seen(1, 8, 3) == 4
    err = capsys.readouterr().err
    assert "sees 1 rank(s), WORLD_SIZE is 8" in err and "rank 3" in err
