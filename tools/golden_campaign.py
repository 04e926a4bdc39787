#!/usr/bin/env python3
"""CPU campaign: the C oracle against the generator-based literal restatement (tests/golden/make_golden_async.py) on fresh
random programs of every generator in tests/fuzz.py — every result field (the determinism-log hash included), 5 seeds per
program.  Neither side is the product; this is what stands in for a run of the Rust reference, which this image cannot
build.  Usage: golden_campaign.py [programs per generator] [base seed]"""
import collections, os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import oracle
from madsim_amd import _abi as A
from tests import fuzz
import make_golden_async as G

FIELDS = ("verdict", "steps", "clock_ns", "msg_count", "rng_calls", "trace_hash", "obs_hash")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
base = int(sys.argv[2]) if len(sys.argv) > 2 else 2_000_000
gens = ["random_workload", "random_lifecycle_workload", "random_rpc_workload", "random_rpc_workload+hooks", "random_addr_workload",
        "random_ephemeral_workload", "random_channel_workload", "random_guard_workload", "random_supervisor_workload", "random_mixed_workload", "random_ipvs_workload", "random_ipvs_runtime_workload",
        "random_timeout_workload", "random_latency_workload"]
if len(sys.argv) > 3:                              # optional: only the generators whose name contains this
    gens = [g for g in gens if sys.argv[3] in g]
t0 = time.time(); total = 0; verdicts = collections.Counter()
for gi, gname in enumerate(gens):
    for k in range(n):
        rng = random.Random(base + 100_000 * gi + k)
        r = fuzz.random_rpc_workload(rng, hooks=True) if gname.endswith("+hooks") else getattr(fuzz, gname)(rng)
        w, cfg, desc = r[0], r[1], r[2]
        lim = fuzz.generous_limits(); lim.max_tasks = 24
        if gname in ("random_supervisor_workload", "random_mixed_workload"):
            lim = fuzz.mixed_limits()
        seeds = (0, 1, 2, 3, 7)
        want, _ = oracle.run_batch(w, 0, max(seeds) + 1, config=cfg, limits=lim)
        for s in seeds:
            if int(want[s]["verdict"]) in (A.OVERFLOW, A.STEP_LIMIT, A.UNSUPPORTED):      # the runner's limits / the workload model's, not a verdict of the simulation
                continue
            g = G.Sim(w, cfg, s).run()
            o = {f: int(want[s][f]) for f in FIELDS}
            if o != {f: g[f] for f in FIELDS}:
                print(f"MISMATCH generator={gname} gen_seed={base + 100_000 * gi + k} seed={s} desc={desc}\n  oracle {o}\n  golden { {f: g[f] for f in FIELDS} }")
                sys.exit(1)
            verdicts[o["verdict"]] += 1; total += 1
print(f"golden campaign ok: {len(gens)} generators x {n} programs x 5 seeds = {total} runs in {time.time() - t0:.0f} s, "
      f"bit-exact on {', '.join(FIELDS)}; verdicts pass/panic/deadlock/time/overflow/steps = {[verdicts[i] for i in range(6)]}")
