#!/bin/bash
# Round 6: BinaryHeap::pop top-down (MADSIM_POP_TOPDOWN, the product) against sift_down_to_bottom + sift_up (libmadsim_hip_bottomup.so), then the
# phase profile of the round's kernels with the wait behind the poll loop (the other lanes' further rounds) in a bucket of its own.
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
BU=MADSIM_HIP_LIB=$PWD/madsim_amd/libmadsim_hip_bottomup.so
timeout 1200 python -m pytest tests -m gpu -x -q > "$O/pytest.txt" 2>&1; tail -2 "$O/pytest.txt"
for round in 1 2 3; do
  run "topo top-down  r$round" topo 16 X=1
  run "topo bottom-up r$round" topo 16 $BU
  run "raft top-down  r$round" raft 16 X=1
  run "raft bottom-up r$round" raft 16 $BU
  run "kv top-down    r$round" kv 24 X=1
  run "kv bottom-up   r$round" kv 24 $BU
done | tee "$O/ab.txt"
for spec in topo:4 raft:4 kv:3; do
  IFS=: read -r wl n <<< "$spec"
  PROF_STREAMS=$n MADSIM_HIP_LIB=$PWD/madsim_amd/libmadsim_hip_prof.so timeout 300 python tools/phase_prof.py "$wl" > "$O/phase_${wl}_$n.txt" 2>&1; cat "$O/phase_${wl}_$n.txt"
done
