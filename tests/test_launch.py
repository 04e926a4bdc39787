"""`bench.py --gpus N` launcher logic (madsim_amd/launch.py): pure host code, runs on CPU."""
import os
import subprocess
import sys

import pytest

from madsim_amd import launch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_single_gpu_runs_inline():
    assert launch.plan(1, {}, [], "bench.py") == ("inline", None)
    assert launch.plan(1, {"WORLD_SIZE": "1"}, [], "bench.py") == ("inline", None)


def test_gpus_flag_spawns_one_rank_per_gpu():
    mode, cmd = launch.plan(8, {}, ["--gpus", "8", "--steps", "5"], "/x/bench.py", port=29511)
    assert mode == "spawn"
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert "--nproc-per-node=8" in cmd and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index("--master-port") + 1] == "29511"
    assert cmd[-5:] == ["/x/bench.py", "--gpus", "8", "--steps", "5"]      # the same arguments reach every rank


def test_under_a_launcher_world_must_match_flag():
    env = {"WORLD_SIZE": "4", "RANK": "2", "LOCAL_RANK": "2"}
    assert launch.plan(4, env, [], "bench.py") == ("inline", None)
    assert launch.rank_env(env) == (2, 2, 4)
    with pytest.raises(launch.LaunchError):          # the round-1 bug: --gpus 8 silently measuring WORLD_SIZE GPUs
        launch.plan(8, env, [], "bench.py")
    with pytest.raises(launch.LaunchError):
        launch.plan(1, env, [], "bench.py")
    with pytest.raises(launch.LaunchError):
        launch.plan(2, {"WORLD_SIZE": "2"}, [], "bench.py")          # no RANK: not started by a launcher
    with pytest.raises(launch.LaunchError):
        launch.plan(0, {}, [], "bench.py")


def test_gathered_report_must_hold_one_row_per_rank():
    assert launch.check_ranks([(0, 0), (1, 1), (2, 2)], 3) == 3
    with pytest.raises(launch.LaunchError):
        launch.check_ranks([(0, 0), (0, 0)], 2)
    with pytest.raises(launch.LaunchError):
        launch.check_ranks([(0, 0)], 2)


def test_bench_refuses_mismatched_world_before_touching_a_gpu():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"], env=env, capture_output=True, text=True)
    assert p.returncode == 2 and "--gpus 8 but the launcher started WORLD_SIZE=2" in p.stderr


def test_bench_without_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"], capture_output=True, text=True)
    assert p.returncode == 2 and "no GPU visible" in p.stderr
