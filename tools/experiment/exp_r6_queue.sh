#!/bin/bash
# Round 6: the election loop's seeds differ in length (4 258 +- 512 steps; the slowest lane of a wave runs 1.27x the mean): launches that hold MORE seeds
# than resident lanes with the per-launch work queue (madsim_limits_t.sched = 1: a finished lane pulls the next seed) against one seed per lane.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/$1; mkdir -p "$O"
line() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); e = d["extra"]
    print(sys.argv[2], "ms/step", round(d["ms_per_step"], 3), round(e["executor_steps_per_sec"] / 1e9, 3), "Gsteps/s", round(e["seeds_per_sec"] / 1e6, 3), "Mseeds/s kernel_ms", round(e["kernel_ms_per_step"], 3),
          "verified", d["verified_seeds"], "failed", e["failed_seeds"], "waves/cu", e.get("waves_per_cu"))
except Exception as ex:
    print(sys.argv[2], "NO LINE:", ex, open(sys.argv[1].replace(".json", ".err")).read()[-400:])
PY
}
run() { label=$1; shift
  timeout 600 python bench.py "$@" --warmup 2 --no-cpu-baseline --no-measure-traffic --no-first-fail --no-extras > "$O/x.json" 2> "$O/x.err"; line "$O/x.json" "$label"; }
for round in 1 2; do
  for wl in raft topo; do
    run "$wl 65536/launch x4 in flight (bench)      r$round" --workload $wl --steps 16
    run "$wl 262144/launch queue, 2 in flight       r$round" --workload $wl --seeds 262144 --sched 1 --streams 2 --steps 6
    run "$wl 262144/launch static, 2 in flight      r$round" --workload $wl --seeds 262144 --sched 0 --streams 2 --steps 6
    run "$wl 524288/launch queue, 2 in flight       r$round" --workload $wl --seeds 524288 --sched 1 --streams 2 --steps 4
    run "$wl 1048576/launch queue, 2 in flight      r$round" --workload $wl --seeds 1048576 --sched 1 --streams 2 --steps 3
    run "$wl 1048576/launch queue, 1 in flight      r$round" --workload $wl --seeds 1048576 --sched 1 --streams 1 --steps 3
  done
done | tee "$O/ab.txt"
