#!/bin/bash
# Round 6: sub-launch size x launches in flight for the topology and the election loop (bench parameters only; the kernels are the product's).
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/$1; mkdir -p "$O"
line() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); e = d["extra"]
    print(sys.argv[2], "ms/step", round(d["ms_per_step"], 3), round(e["executor_steps_per_sec"] / 1e9, 3), "Gsteps/s", round(e["seeds_per_sec"] / 1e6, 3), "Mseeds/s kernel_ms", round(e["kernel_ms_per_step"], 3),
          "verified", d["verified_seeds"], "failed", e["failed_seeds"])
except Exception as ex:
    print(sys.argv[2], "NO LINE:", ex, open(sys.argv[1].replace(".json", ".err")).read()[-400:])
PY
}
run() { label=$1; shift
  timeout 600 python bench.py "$@" --warmup 4 --no-cpu-baseline --no-measure-traffic --no-first-fail --no-extras > "$O/x.json" 2> "$O/x.err"; line "$O/x.json" "$label"; }
for round in 1 2; do
  for wl in topo raft; do
    run "$wl 65536 x4 (bench)  r$round" --workload $wl --steps 16
    run "$wl 65536 x3          r$round" --workload $wl --steps 16 --streams 3
    run "$wl 65536 x5          r$round" --workload $wl --steps 16 --streams 5
    run "$wl 65536 x6          r$round" --workload $wl --steps 16 --streams 6
    run "$wl 32768 x6          r$round" --workload $wl --seeds 32768 --steps 32 --streams 6
    run "$wl 32768 x8          r$round" --workload $wl --seeds 32768 --steps 32 --streams 8
    run "$wl 32768 x12         r$round" --workload $wl --seeds 32768 --steps 32 --streams 12
    run "$wl 16384 x16         r$round" --workload $wl --seeds 16384 --steps 64 --streams 16
    run "$wl 131072 x2         r$round" --workload $wl --seeds 131072 --steps 8 --streams 2
    run "$wl 131072 x3         r$round" --workload $wl --seeds 131072 --steps 8 --streams 3
  done
done | tee "$O/ab.txt"
