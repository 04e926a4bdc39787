#!/usr/bin/env python3
"""Diff the reference's fingerprints (JSONL printed by madsim-ref-twin, i.e. REAL madsim) against the CPU oracle.

    python compare.py ref.jsonl              # exit 0 = every seed identical: the oracle is pinned on these workloads
    python compare.py --emit-oracle all 0 8  # print the oracle's side in the same schema (what ref.jsonl must equal)

Compared per seed: verdict, elapsed_ns (madsim::time::Instant), msg_count (NetSim::stat), the observed-value list
`obs` (ending with the trailing random::<u32>() — pins draw count and generator state), and, when the reference was
built with --features rng-log, the raw determinism log bytes (rand.rs:64-88).  The schema is tools/ref_twin/schema.json.
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import twin_workloads as T      # noqa: E402

from madsim_amd import _abi as A      # noqa: E402
import oracle                         # noqa: E402


def oracle_record(name, seed, loss=0.0, want_log=False):
    """One record of the schema, from the CPU oracle."""
    w = T.ALL[name]()
    cfg = T.config(name, loss)
    obs, res = oracle.observe_seed(w, seed, cfg)
    ok = res.verdict == A.PASS
    rec = {"workload": name, "seed": seed, "loss": loss, "verdict": T.VERDICTS[res.verdict],
           "elapsed_ns": obs[-2] if ok else None,          # the tail observes elapsed, then the trailing draw
           "msg_count": res.msg_count if ok else None, "obs": obs}
    if want_log:
        rec["log_hex"] = oracle.trace_seed(w, seed, cfg)[0].hex()
    return rec


def compare(ref):
    """ref: a parsed reference record. Returns a list of mismatch strings (empty = identical)."""
    name, seed, loss = ref["workload"], int(ref["seed"]), float(ref.get("loss", 0.0))
    mine = oracle_record(name, seed, loss, want_log="log_hex" in ref)
    bad = []
    for k in ("verdict", "elapsed_ns", "msg_count", "obs"):
        if ref[k] != mine[k]:
            bad.append(f"{k}: reference {ref[k]} != oracle {mine[k]}")
    if "log_hex" in ref and ref["log_hex"] != mine["log_hex"]:
        a, b = bytes.fromhex(ref["log_hex"]), bytes.fromhex(mine["log_hex"])
        i = next((k for k in range(min(len(a), len(b))) if a[k] != b[k]), min(len(a), len(b)))
        bad.append(f"determinism log differs at byte {i} (lengths {len(a)} / {len(b)})")
    return bad


def main(argv):
    if len(argv) >= 2 and argv[1] == "--emit-oracle":
        names = list(T.ALL) if argv[2] == "all" else [argv[2]]
        seed0, count = int(argv[3]), int(argv[4])
        loss = float(argv[5]) if len(argv) > 5 else 0.0
        for n in names:
            for s in range(seed0, seed0 + count):
                print(json.dumps(oracle_record(n, s, loss)))
        return 0
    if len(argv) != 2:
        print(__doc__)
        return 2
    n = n_bad = 0
    for line in open(argv[1]):
        line = line.strip()
        if not line:
            continue
        ref = json.loads(line)
        bad = compare(ref)
        n += 1
        if bad:
            n_bad += 1
            if n_bad <= 20:
                print(f"MISMATCH {ref['workload']} seed {ref['seed']}: " + "; ".join(bad))
    print(f"{n} reference records, {n_bad} mismatches" + ("" if n_bad else " — oracle pinned on these workloads"))
    return 1 if n_bad or not n else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
