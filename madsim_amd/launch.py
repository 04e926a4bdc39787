"""How `bench.py --gpus N` becomes N ranks (one process per GPU, RCCL over xGMI).

The reference drives every seed from one process (`Builder::run`, runtime/builder.rs:129-150); the multi-GPU form here
is one process per GPU, so a plain `python bench.py --gpus 8` must turn itself into 8 ranks.  Pure host logic (no torch
import, no device code): unit-tested on CPU in tests/test_launch.py.
"""
import os
import socket
import sys


class LaunchError(RuntimeError):
    pass


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def plan(gpus, env, argv, script, port=None):
    """Decide how this invocation runs.

    Returns ("inline", None) when this process is itself a rank (world size 1, or launched by torchrun with a
    WORLD_SIZE that matches --gpus), or ("spawn", cmd) with the torch.distributed.run command line that re-executes
    `script` as `gpus` ranks.  A --gpus / WORLD_SIZE mismatch is an error: silently benchmarking a different number of
    GPUs than asked for is how a scaling curve goes wrong.
    """
    if gpus < 1:
        raise LaunchError("--gpus must be >= 1")
    if "WORLD_SIZE" in env:
        try:
            world = int(env["WORLD_SIZE"])
        except ValueError:
            raise LaunchError("WORLD_SIZE is not an integer")
        if world != gpus:
            raise LaunchError(f"--gpus {gpus} but the launcher started WORLD_SIZE={world} ranks")
        for k in ("RANK", "LOCAL_RANK"):
            if world > 1 and k not in env:
                raise LaunchError(f"WORLD_SIZE={world} without {k}: start the ranks with torch.distributed.run")
        return "inline", None
    if gpus == 1:
        return "inline", None
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port or free_port()), script] + list(argv)
    return "spawn", cmd


def rank_env(env):
    """(rank, local_rank, world) of this process; (0, 0, 1) outside a launcher."""
    return int(env.get("RANK", "0")), int(env.get("LOCAL_RANK", "0")), int(env.get("WORLD_SIZE", "1"))


def check_ranks(rows, world):
    """`rows`: the per-rank identity words of one gathered report, [(rank, device_index), ...].  Every rank must appear
    exactly once and (RCCL runs) sit on its own GPU.  Returns the number of distinct ranks."""
    ranks = sorted(int(r) for r, _ in rows)
    if ranks != list(range(world)):
        raise LaunchError(f"gathered report does not hold one row per rank: {ranks} (world {world})")
    return len(set(ranks))
