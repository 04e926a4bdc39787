#!/bin/bash
# Round 6: dependent round trips taken out of the poll handlers and the fire loop (copies popped by the fire loop itself, spawn's reads in one batch,
# timeout(recv)'s first message with the header, rpc unit with the poll's units, flag_add as an add in memory) — the product against
# libmadsim_hip_<base>.so (tools/build_baseline.sh <rev> <base>).   usage: exp_r6_chains.sh <outdir> <base> [rounds]
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/$1; mkdir -p "$O"; BASE=$2; ROUNDS=${3:-3}
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
B=MADSIM_HIP_LIB=$PWD/madsim_amd/libmadsim_hip_$BASE.so
timeout 1200 python -m pytest tests -m gpu -x -q > "$O/pytest.txt" 2>&1; tail -2 "$O/pytest.txt"
for round in $(seq 1 $ROUNDS); do
  run "topo     new  r$round" topo 16 X=1
  run "topo     base r$round" topo 16 $B
  run "raft     new  r$round" raft 16 X=1
  run "raft     base r$round" raft 16 $B
  run "kv       new  r$round" kv 24 X=1
  run "kv       base r$round" kv 24 $B
  run "pingpong new  r$round" pingpong 20 X=1
  run "pingpong base r$round" pingpong 20 $B
done | tee "$O/ab.txt"
