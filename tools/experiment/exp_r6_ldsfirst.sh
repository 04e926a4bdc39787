#!/bin/bash
# Round 6: timer_pop walks the LDS-resident heap levels before the heap's last entry (a spill-region load) has arrived (MADSIM_POP_LDS_FIRST, the
# product) against the plain top-down pop (libmadsim_hip_tdonly.so = -DMADSIM_POP_LDS_FIRST=0).
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
TD=MADSIM_HIP_LIB=$PWD/madsim_amd/libmadsim_hip_tdonly.so
timeout 1200 python -m pytest tests -m gpu -x -q > "$O/pytest.txt" 2>&1; tail -2 "$O/pytest.txt"
for round in 1 2 3; do
  run "topo   lds-first r$round" topo 16 X=1
  run "topo   top-down  r$round" topo 16 $TD
  run "raft   lds-first r$round" raft 16 X=1
  run "raft   top-down  r$round" raft 16 $TD
  run "timers lds-first r$round" timers 40 X=1
  run "timers top-down  r$round" timers 40 $TD
done | tee "$O/ab.txt"
