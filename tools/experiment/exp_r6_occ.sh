#!/bin/bash
# Round 6: the re-registration table's occupancy mirrored in a register (MADSIM_DEDUP_OCC: empty buckets are never loaded) against loading the bucket
# every time (libmadsim_hip_noocc.so).  Election loop (the only bench case with MADSIM_STATE_DEDUP_TIMERS).
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
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "dedup or raft or config2 or timeout or narrow" > "$O/pytest.txt" 2>&1; tail -2 "$O/pytest.txt"
for round in 1 2 3 4; do
  run "raft occupancy mirror r$round" raft 20 X=1
  run "raft every bucket     r$round" raft 20 MADSIM_HIP_LIB=$PWD/madsim_amd/libmadsim_hip_noocc.so
done | tee "$O/ab.txt"
