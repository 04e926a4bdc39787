#!/bin/bash
# Round 6: the log's number kept in hand (k_rng.h Peek) in the EXTENDED builds too (libmadsim_hip_peeklife.so, -DMADSIM_RNG_PEEK_LIFE=1) against the
# product (base-op builds only).  Two more live VGPRs: the channel builds spill 5-6 registers instead of 1-2, the every-class builds go 242 -> 246.
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
PL=MADSIM_HIP_LIB=$PWD/madsim_amd/libmadsim_hip_peeklife.so
MADSIM_HIP_LIB=$PWD/madsim_amd/libmadsim_hip_peeklife.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > "$O/pytest_peeklife.txt" 2>&1; tail -2 "$O/pytest_peeklife.txt"
for round in 1 2 3; do
  run "topo product   r$round" topo 16 X=1
  run "topo peek      r$round" topo 16 $PL
  run "raft product   r$round" raft 16 X=1
  run "raft peek      r$round" raft 16 $PL
  run "kv product     r$round" kv 24 X=1
  run "kv peek        r$round" kv 24 $PL
done | tee "$O/ab.txt"
