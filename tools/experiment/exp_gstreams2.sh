#!/bin/bash
# Batches in flight x library for the global-state workloads, with regions long enough to be steady state (steps >= 6 rounds of the
# streams): does a shorter wave lifetime show in wall time once enough launches are queued behind the resident ones?
# Usage (GPU box): bash tools/experiment/exp_gstreams2.sh <out-tag> "<libs>" "<workloads>" "<stream counts>" [steps] [repeats]
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/$1; mkdir -p "$O"
for wl in $3; do for s in $4; do for lib in $2; do
  MADSIM_HIP_LIB=$PWD/madsim_amd/$lib timeout 300 python bench.py --workload $wl --steps ${5:-24} --warmup 4 --repeats ${6:-5} --streams $s --no-cpu-baseline --no-measure-traffic --no-first-fail --no-extras > "$O/l.json" 2> "$O/l.err"
  python - "$O/l.json" "$wl streams $s $lib" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); e = d["extra"]
    print(sys.argv[2], "ms/step", round(d["ms_per_step"], 3), round(e["executor_steps_per_sec"] / 1e9, 3), "Gsteps/s kernel_ms", round(e["kernel_ms_per_step"], 3), "verified", d["verified_seeds"], "failed", e["failed_seeds"])
except Exception as ex:
    print(sys.argv[2], "NO LINE:", ex)
PY
done; done; done | tee "$O/gstreams2.txt"
