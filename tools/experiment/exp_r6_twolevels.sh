#!/bin/bash
# Round 6: two spilled sift-down levels per global round trip (MADSIM_POP_TWO_LEVELS, narrow entries) against one level per trip (libmadsim_hip_onelevel.so).
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/$1; mkdir -p "$O"
line() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); e = d["extra"]
    print(sys.argv[2], "ms/step", round(d["ms_per_step"], 4), round(e["executor_steps_per_sec"] / 1e9, 3), "Gsteps/s kernel_ms", round(e["kernel_ms_per_step"], 3),
          "verified", d["verified_seeds"], "failed", e["failed_seeds"])
except Exception as ex:
    print(sys.argv[2], "NO LINE:", ex, open(sys.argv[1].replace(".json", ".err")).read()[-400:])
PY
}
run() { label=$1; wl=$2; steps=$3; shift 3
  env "$@" timeout 400 python bench.py --workload "$wl" --steps "$steps" --warmup 4 --no-cpu-baseline --no-measure-traffic --no-first-fail --no-extras \
    > "$O/x.json" 2> "$O/x.err"; line "$O/x.json" "$label"; }
OL=MADSIM_HIP_LIB=$PWD/madsim_amd/libmadsim_hip_onelevel.so
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "narrow or bench_configuration or config2 or config4" > "$O/pytest.txt" 2>&1; tail -2 "$O/pytest.txt"
for round in 1 2 3; do
  run "topo two levels r$round" topo 16 X=1
  run "topo one level  r$round" topo 16 $OL
  run "raft two levels r$round" raft 16 X=1
  run "raft one level  r$round" raft 16 $OL
done | tee "$O/ab.txt"
