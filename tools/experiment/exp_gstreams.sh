#!/bin/bash
# Batches in flight for the global-state workloads: 3 (one per resident wave of a SIMD) against 4 (a fourth launch queued behind them
# fills the slots the tails of the three leave — a launch ends with its slowest wave).  Two interleaved rounds on one box.
# Usage (GPU box): bash tools/experiment/exp_gstreams.sh <out-tag>
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/$1; mkdir -p "$O"
for r in 1 2; do for wl in raft topo kv; do for s in 3 4; do
  timeout 300 python bench.py --workload $wl --steps 12 --warmup 4 --streams $s --no-cpu-baseline --no-measure-traffic --no-first-fail --no-extras > "$O/l.json" 2> "$O/l.err"
  python - "$O/l.json" "$wl streams $s r$r" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); e = d["extra"]
    print(sys.argv[2], "ms/step", round(d["ms_per_step"], 3), round(e["executor_steps_per_sec"] / 1e9, 3), "Gsteps/s kernel_ms", round(e["kernel_ms_per_step"], 3), "verified", d["verified_seeds"], "failed", e["failed_seeds"])
except Exception as ex:
    print(sys.argv[2], "NO LINE:", ex)
PY
done; done; done | tee "$O/gstreams.txt"
