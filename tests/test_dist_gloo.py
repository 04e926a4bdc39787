"""The N>1 path on CPU: two gloo ranks shard a seed range and combine their first-fail reports with the same
collective code bench.py runs over RCCL (madsim_amd/dist.py).  The per-rank 'simulation' here is the CPU
oracle because this container has no GPU; the sharding + reduction logic is what is under test."""
import os
import socket

import numpy as np
import torch.distributed as dist
import torch.multiprocessing as mp

from madsim_amd import dist as mdist


def test_shard_range_tiles_the_batch():
    for count in (0, 1, 5, 64, 65536, 65537):
        for world in (1, 2, 3, 8):
            blocks = [mdist.shard_range(1000, count, r, world) for r in range(world)]
            assert sum(n for _, n in blocks) == count
            nxt = 1000
            for s, n in blocks:
                if n:
                    assert s == nxt
                    nxt += n


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    from madsim_amd import _abi as A
    from madsim_amd import workload as W
    w = W.pingpong(4, 16)
    cfg = A.Config.default(packet_loss_rate=0.02)
    seed0, count = mdist.shard_range(0, 600, rank, world)
    out, s = oracle.run_batch(w, seed0, count, cfg)
    rep = mdist.reduce_report(s.first_failing_seed, s.n_failed, s.total_steps, s.total_clock_ns)
    # a rank with no failure reports UINT64_MAX: the unsigned-min mapping must not be fooled by it
    rep2 = mdist.reduce_report((1 << 64) - 1 if rank == 0 else 5, 0, 0, 0)
    # the device-side form bench.py uses over RCCL (here: CPU tensors over gloo); word 0 is seed ^ (1 << 63) as int64
    import torch
    key = (s.first_failing_seed ^ (1 << 63))
    key = key - (1 << 64) if key >= (1 << 63) else key
    t4 = torch.tensor([key, s.n_failed, s.total_steps, s.total_clock_ns], dtype=torch.int64)
    mdist.reduce_report_device(t4)
    rep3 = (mdist.decode_first_fail(t4[0]), int(t4[1]), int(t4[2]), int(t4[3]))
    # the single-gather form
    t4 = torch.tensor([key, s.n_failed, s.total_steps, s.total_clock_ns], dtype=torch.int64)
    g = mdist.combine_gathered(mdist.gather_report_device(t4, torch.zeros((world, 4), dtype=torch.int64)))
    rep4 = (mdist.decode_first_fail(g[0]), int(g[1]), int(g[2]), int(g[3]))
    q.put((rank, rep, rep2, rep3, rep4))
    dist.destroy_process_group()


def test_two_rank_first_fail_report():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    import oracle
    from madsim_amd import _abi as A
    from madsim_amd import workload as W
    out, s = oracle.run_batch(W.pingpong(4, 16), 0, 600, A.Config.default(packet_loss_rate=0.02))
    want = (s.first_failing_seed, s.n_failed, s.total_steps, s.total_clock_ns)
    assert s.n_failed > 0
    for rank, rep, rep2, rep3, rep4 in res:
        assert rep == want, (rank, rep, want)
        assert rep2[0] == 5
        assert rep3 == want, (rank, rep3, want)
        assert rep4 == want, (rank, rep4, want)


# ---- world size 8: bench.py's launcher path end to end on CPU ------------------------------------------------------------
# What `python -m torch.distributed.run --nproc-per-node 8 bench.py --gpus 8` does around the kernel, with the oracle standing in
# for the GPU: the launcher environment -> launch.plan / rank_env, the rank -> seed-block map (uneven: count % 8 != 0), one
# 8-word report row per rank and step ({first-fail key, failed, steps, clock, rank, device, 0, 0}), ONE all-gather per step,
# launch.check_ranks on every gathered step and dist.combine_gathered over the steps (runtime/builder.rs:129-150: the seeds are
# seed0 .. seed0 + count whoever runs them).
REPORT_WORDS = 8


def _worker8(rank, world, port, count, n_steps, q):
    env = {"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)}
    os.environ.update(env)
    import torch
    import oracle
    from madsim_amd import _abi as A, launch
    from madsim_amd import workload as W
    assert launch.plan(world, env, ["--gpus", str(world)], "bench.py") == ("inline", None)
    assert launch.rank_env(env) == (rank, rank, world)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    w = W.pingpong(4, 8)
    cfg = A.Config.default(packet_loss_rate=0.01)
    gathered = torch.zeros((n_steps, world, REPORT_WORDS), dtype=torch.int64)
    blocks = []
    for k in range(n_steps):
        seed0, n = mdist.shard_range(k * count, count, rank, world)        # a fresh block of seeds every step
        blocks.append((seed0, n))
        _, s = oracle.run_batch(w, seed0, n, cfg)
        key = s.first_failing_seed ^ (1 << 63)
        key = key - (1 << 64) if key >= (1 << 63) else key
        row = torch.tensor([key, s.n_failed, s.total_steps, s.total_clock_ns, rank, rank % 8, 0, 0], dtype=torch.int64)
        mdist.gather_report_device(row, gathered[k])
    for row in gathered:
        assert launch.check_ranks([(int(r[4]), int(r[5])) for r in row], world) == world
    rows = mdist.combine_gathered(gathered)
    q.put((rank, blocks, [(mdist.decode_first_fail(r[0]), int(r[1]), int(r[2]), int(r[3])) for r in rows]))
    dist.destroy_process_group()


def test_eight_rank_launcher_path_uneven_count():
    world, count, n_steps = 8, 203, 3                      # 203 = 8 * 25 + 3: blocks of 26 seeds, the last one of 21
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker8, args=(r, world, port, count, n_steps, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    import oracle
    from madsim_amd import _abi as A
    from madsim_amd import workload as W
    w, cfg = W.pingpong(4, 8), A.Config.default(packet_loss_rate=0.01)
    want = []
    for k in range(n_steps):
        _, sm = oracle.run_batch(w, k * count, count, cfg)
        want.append((sm.first_failing_seed, sm.n_failed, sm.total_steps, sm.total_clock_ns))
    assert any(wf[1] for wf in want)
    res.sort()
    for k in range(n_steps):                               # the ranks' blocks tile the step's seed range, in rank order
        nxt = k * count
        for rank, blocks, _ in res:
            s0, n = blocks[k]
            assert n in (26, 21) and (n == 0 or s0 == nxt)
            nxt += n
        assert nxt == (k + 1) * count
    for rank, _, rows in res:
        assert rows == want, (rank, rows, want)


def test_check_ranks_refuses_a_missing_or_doubled_rank():
    import pytest
    from madsim_amd import launch
    assert launch.check_ranks([(r, r) for r in range(8)], 8) == 8
    for bad in ([(r, r) for r in range(7)] + [(6, 6)], [(r, r) for r in range(7)], [(r, r) for r in range(9)]):
        with pytest.raises(launch.LaunchError):
            launch.check_ranks(bad, 8)


# ---- the seed search over ranks (madsim_amd/dist.py campaign_over_ranks): one all-gather per round ------------------------------------
def _oracle_report(w, cfg, batch, stop):
    """The oracle standing in for `madsim_hip_run_campaign` on one chunk: batch after batch, and — with `stop` — no further than the batch
    that holds the chunk's first genuine failure (MADSIM_CAMPAIGN_STOP_AT_FAILURE: seeds_run / batches_run are that prefix)."""
    import oracle
    from madsim_amd import _abi as A

    def run(seed_lo, n):
        first, nf, nr, st, ck, ran, nb = (1 << 64) - 1, 0, 0, 0, 0, 0, 0
        for lo in range(0, n, batch):
            m = min(batch, n - lo)
            out, _ = oracle.run_batch(w, seed_lo + lo, m, cfg)
            v = out["verdict"]
            genuine = (v != A.PASS) & (v < A.OVERFLOW)
            nf += int(genuine.sum()); nr += int((v >= A.OVERFLOW).sum())
            st += int(out["steps"].astype(np.int64).sum()); ck += int(out["clock_ns"].astype(np.int64).sum())
            ran += m; nb += 1
            if genuine.any():
                first = min(first, int(seed_lo + lo + int(np.nonzero(genuine)[0][0])))
                if stop:
                    break
        return first, nf, nr, st, ck, ran, nb
    return run


def _worker_campaign(rank, world, port, cases, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from madsim_amd import _abi as A
    from madsim_amd import workload as W
    w = W.pingpong(4, 8)
    res = []
    for loss, seed0, total, batch, stop, rb in cases:
        res.append(mdist.campaign_over_ranks(_oracle_report(w, A.Config.default(packet_loss_rate=loss), batch, stop), seed0, total, batch, stop, round_batches=rb))
    q.put((rank, res))
    dist.destroy_process_group()


def test_campaign_over_ranks_matches_one_process_and_stops_within_a_round():
    """world 3 over gloo, the oracle standing in for the GPU: without stop the report of the whole range; with stop the smallest failing
    seed, found in the round that holds it, counting exactly the batches up to the failing one; ragged last batch, fewer chunks than
    ranks, no failure at all; one batch per rank and round (round 5's form) and PIPELINED rounds of several batches per rank (round 6:
    every rank runs its chunk as one campaign call).  Every rank computes the same dict from the one all-gather per round, and that
    dict is the single-process campaign's for the same prefix whatever the chunking."""
    from madsim_amd import _abi as A
    from madsim_amd import workload as W
    cases = [(0.01, 7000, 1000, 64, False, 1), (0.002, 50_000, 4000, 128, True, 1), (0.0, 0, 500, 64, True, 1), (0.01, 90_000, 100, 64, True, 1),
             (0.01, 7000, 1000, 64, False, 4), (0.002, 50_000, 4000, 128, True, 3), (0.0, 0, 500, 64, True, 5), (0.002, 50_000, 4000, 128, True, 20)]
    world = 3
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_campaign, args=(r, world, port, cases, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[0] == res[1] == res[2]                      # no second collective needed: every rank folds the same rows
    w = W.pingpong(4, 8)
    for (loss, seed0, total, batch, stop, rb), got in zip(cases, res[0]):
        # the single-process search of the same range, one batch at a time (world 1, no process group, no chunking)
        one = mdist.campaign_over_ranks(_oracle_report(w, A.Config.default(packet_loss_rate=loss), batch, stop), seed0, total, batch, stop)
        for f in ("first_failing_seed", "n_failed", "n_runner", "total_steps", "total_clock_ns", "seeds_run", "batches_run"):
            assert got[f] == one[f], (loss, rb, f, got, one)
        n_batches = -(-total // batch)
        n_chunks = -(-n_batches // rb)
        if stop and got["first_failing_seed"] != (1 << 64) - 1:
            j = (got["first_failing_seed"] - seed0) // batch
            assert got["batches_run"] == j + 1 and got["rounds"] == (j // rb) // world + 1
        else:
            assert got["batches_run"] == n_batches and got["seeds_run"] == total and got["rounds"] == -(-n_chunks // world)
    assert res[0][0]["n_failed"] > 0 and res[0][1]["first_failing_seed"] != (1 << 64) - 1 and res[0][2]["n_failed"] == 0
