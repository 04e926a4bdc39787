#!/bin/bash
# Round 6: "is anything clogged" mirrored in the lane (MADSIM_CLOG_MIRROR: a send loads the clog masks only then) against loading them on every send
# (libmadsim_hip_noclogm.so).
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
NM=MADSIM_HIP_LIB=$PWD/madsim_amd/libmadsim_hip_noclogm.so
timeout 1200 python -m pytest tests -m gpu -x -q > "$O/pytest.txt" 2>&1; tail -2 "$O/pytest.txt"
for round in 1 2 3; do
  run "raft mirror    r$round" raft 20 X=1
  run "raft no mirror r$round" raft 20 $NM
  run "topo mirror    r$round" topo 16 X=1
  run "topo no mirror r$round" topo 16 $NM
  run "kv mirror      r$round" kv 24 X=1
  run "kv no mirror   r$round" kv 24 $NM
done | tee "$O/ab.txt"
