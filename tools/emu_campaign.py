#!/usr/bin/env python3
"""CPU campaign: the kernel sources compiled for the host (tests/emu — a debugging aid, never the product path) against the oracle on fresh
random programs of every generator in tests/fuzz.py, odd programs with the per-seed state in the global-memory block.  What it is for:
kernel *logic* (the workload VM, the executor loop) checked at scale without GPU time; what it cannot see: anything the hardware or the
device compiler adds.  Usage: emu_campaign.py [programs per generator] [base seed] [tight]
`tight`: random stingy capacities (tasks, registrations, queued messages, heap slots, connections) instead of generous ones — the kernel
must then give the capacity verdict and, re-run with grown capacities, the oracle's answer."""
import os, sys, random, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, oracle
from tests import fuzz, emu, parity
from madsim_amd import _abi as A
gens = [("random_workload", None, None), ("random_lifecycle_workload", 24, None), ("random_rpc_workload", 24, None), ("random_rpc_workload", 24, "hooks"),
        ("random_addr_workload", None, None), ("random_ephemeral_workload", None, None), ("random_channel_workload", 24, None),
        ("random_guard_workload", 24, None), ("random_supervisor_workload", 48, None), ("random_mixed_workload", 60, None), ("random_ipvs_workload", 24, None), ("random_ipvs_runtime_workload", 24, None),
        ("random_timeout_workload", None, None), ("random_reply_without_receive_workload", None, None),
        ("random_unstructured_workload", 16, None), ("random_latency_workload", None, None)]
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200; base = int(sys.argv[2]) if len(sys.argv) > 2 else 3_000_000; TIGHT = len(sys.argv) > 3 and sys.argv[3] == 'tight'; t0=time.time(); total=0; bad=0; tally=parity.Tally()
for gi,(g,mt,opt) in enumerate(gens):
    for k in range(N):
        rng = random.Random(base + 100000*gi + k)
        r = fuzz.random_rpc_workload(rng, hooks=True) if opt else getattr(fuzz, g)(rng)
        w, cfg, desc = r[0], r[1], r[2]
        lim = fuzz.mixed_limits() if mt == 60 else fuzz.mailbox_limits() if g in ('random_timeout_workload', 'random_reply_without_receive_workload', 'random_latency_workload') else fuzz.generous_limits()
        if mt and mt != 60: lim.max_tasks = mt
        if TIGHT:
            lr = random.Random(k)
            lim = A.Limits(); lim.max_steps = 200000
            lim.max_tasks = lr.choice([0, w.struct.n_progs, w.struct.n_progs + 2, 12])
            lim.mbox_regs, lim.mbox_msgs = lr.choice([1, 2, 4]), lr.choice([1, 2, 4])
            lim.heap_lds_slots, lim.heap_spill_slots = lr.choice([2, 4, 8]), lr.choice([0, 4, 16])
            lim.max_conns, lim.chan_queue = lr.choice([1, 2, 4]), lr.choice([1, 2])
            lim.lanes_per_wave = lr.choice([0, 16, 64])
        # (odd programs: the global-memory block; the re-registration counts with it for the timeout generator always, for the others
        # every fourth program — the builds that do not carry the switch ignore it)
        if k % 2: lim.lanes_per_wave, lim.state_mem = 0, A.STATE_GLOBAL | (A.STATE_DEDUP_TIMERS if g == "random_timeout_workload" or k % 4 == 3 else 0)
        if k % 8 in (5, 7):                           # ... and 8-byte heap entries with a small LDS quota (round 6; honoured where a narrow build exists)
            lim.state_mem |= A.STATE_NARROW_HEAP
            q = 1 + k % 5
            lim.heap_spill_slots, lim.heap_lds_slots = max(4, lim.heap_spill_slots + max(0, lim.heap_lds_slots - q)), q      # (from 4 slots the re-run rounds of tests/parity.py reach the 1 024 a restart storm wants)
        try:
            e = emu.run_batch(w, k * 5, 8, cfg, lim)
        except RuntimeError:                      # refused by validate() (the op-soup generator writes programs that are)
            continue
        o, _ = oracle.run_batch(w, k * 5, 8, cfg, lim)
        total += 8
        try:                                      # every seed is compared: first-pass capacity verdicts are re-run with grown capacities (tests/parity.py)
            parity.compare(e, o, lambda: parity.resolve_seed_by_seed(emu.run_batch, w, k * 5, e, cfg, lim), g + ("+hooks" if opt else ""), tally,
                           (g, base + 100000*gi + k, desc[:120]), lambda i: parity.beyond_ceiling(w, k * 5 + i, cfg, lim))
        except AssertionError as ex:
            bad += 1; print("MISMATCH", ex)
            if bad >= 5: sys.exit(1)
print(f"emu campaign {'ok' if not bad else 'FAILED'}: {len(gens)} generators x {N} programs x 8 seeds = {total} seeds in {time.time()-t0:.0f} s, kernel (compiled for the host) == oracle "
      f"on all 48 result bytes; {tally.rerun} seeds compared after a re-run with grown capacities, {tally.unresolved} proven beyond the layout's ceilings; mismatches {bad}")
print("per generator:", tally)
sys.exit(1 if bad else 0)
