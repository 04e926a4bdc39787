"""Multi-GPU sharding of a seed batch: one process per GPU, no data-path collective.

Seeds are independent (the reference shares nothing between seeds but the read-only Config,
madsim/src/sim/runtime/builder.rs:129-150), so rank g simply runs the contiguous block
[seed0 + g*ceil(count/G), ...).  The single exchange is the end-of-batch report: one all-reduce(min) on
the first failing seed and one all-reduce(sum) on the failure / step / sim-time counters
(RCCL over xGMI when the backend is "nccl"; "gloo" on CPU for tests), or a single all-gather of the 32-byte
reports (gather_report_device) folded afterwards.  Payload: 32 bytes.
"""
import torch
import torch.distributed as dist

U64_MAX = (1 << 64) - 1
_I63 = (1 << 63) - 1


def shard_range(seed0, count, rank, world):
    """Contiguous block of rank `rank`: (first seed, number of seeds). Blocks tile [seed0, seed0+count)."""
    chunk = -(-count // world)
    lo = min(rank * chunk, count)
    hi = min(lo + chunk, count)
    return seed0 + lo, hi - lo


def reduce_report(first_failing_seed, n_failed, total_steps, total_clock_ns, device="cpu", group=None):
    """Combine per-rank summaries. Returns (first_failing_seed, n_failed, total_steps, total_clock_ns).

    Seeds are compared as unsigned 64-bit: they are mapped to int64 by flipping the sign bit so that
    all-reduce(MIN) on int64 orders them as u64 (torch has no uint64 reductions).
    """
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return first_failing_seed, n_failed, total_steps, total_clock_ns
    key = (first_failing_seed ^ (1 << 63))
    key = key - (1 << 64) if key > _I63 else key
    mn = torch.tensor([key], dtype=torch.int64, device=device)
    sm = torch.tensor([n_failed, total_steps, total_clock_ns], dtype=torch.int64, device=device)
    dist.all_reduce(mn, op=dist.ReduceOp.MIN, group=group)
    dist.all_reduce(sm, op=dist.ReduceOp.SUM, group=group)
    k = int(mn.item())
    k = k + (1 << 64) if k < 0 else k
    n, s, c = (int(x) for x in sm.tolist())
    return k ^ (1 << 63), n, s, c


def reduce_report_device(summary4, group=None):
    """Device-side report exchange: `summary4` is the int64[4] CUDA tensor madsim_hip_run_batch_async fills
    ({first failing seed ^ (1 << 63), n_failed, total_steps, total_clock_ns}).  Two in-place RCCL all-reduces, no host
    synchronisation: signed MIN on word 0 is the unsigned minimum of the seeds, SUM on words 1-3."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(summary4[0:1], op=dist.ReduceOp.MIN, group=group)
        dist.all_reduce(summary4[1:4], op=dist.ReduceOp.SUM, group=group)
    return summary4


def gather_report_device(summary4, gathered, group=None):
    """The single-collective form: one RCCL all-gather of the per-rank report row into `gathered` (int64[world, W], same
    device; W >= 4 — words 0-3 are the library's report, further words are the caller's, e.g. bench.py's rank and device
    identity), no host synchronisation.  `combine_gathered` folds the rows later, on the host or on the device."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_gather_into_tensor(gathered.view(-1), summary4, group=group)
    else:
        gathered.view(-1)[:summary4.numel()].copy_(summary4)
    return gathered


def combine_gathered(gathered):
    """int64[..., world, W >= 4] of gather_report_device -> (first failing seed key, n_failed, total_steps, total_clock_ns)
    reduced over the world axis: signed MIN of the keys (= unsigned minimum of the seeds), SUM of the counters."""
    return torch.cat([gathered[..., 0].min(dim=-1, keepdim=True).values, gathered[..., 1:4].sum(dim=-2)], dim=-1)


def decode_first_fail(key):
    """int64 key of reduce_report_device -> u64 seed (U64_MAX = no failing seed)."""
    return (int(key) & U64_MAX) ^ (1 << 63)
