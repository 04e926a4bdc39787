#!/bin/bash
# Round 6: (1) identical wake-ups fired as a batch (MADSIM_FIRE_COPIES, the product) against one callback per copy (libmadsim_hip_nocopies.so);
# (2) the headline kernel of this round against round 5's tree (_r5tree: `git archive 8add2ed`, its own bench.py and library) on the same box.
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
N=MADSIM_BENCH_STATE_FLAGS=0x200
NC=MADSIM_HIP_LIB=$PWD/madsim_amd/libmadsim_hip_nocopies.so
for round in 1 2 3; do
  run "topo narrow31 copies-batched r$round" topo 16 $N MADSIM_BENCH_HEAP_LDS=31
  run "topo narrow31 one-by-one     r$round" topo 16 $N MADSIM_BENCH_HEAP_LDS=31 $NC
  run "kv copies-batched            r$round" kv 24 X=1
  run "kv one-by-one                r$round" kv 24 $NC
  run "raft l64 narrow no-dedup batched    r$round" raft 16 MADSIM_BENCH_STATE_FLAGS=0x200 MADSIM_BENCH_LPW=64 MADSIM_BENCH_CLEAR_FLAGS=0x100
  run "raft l64 narrow no-dedup one-by-one r$round" raft 16 MADSIM_BENCH_STATE_FLAGS=0x200 MADSIM_BENCH_LPW=64 MADSIM_BENCH_CLEAR_FLAGS=0x100 $NC
  run "raft l64 narrow dedup               r$round" raft 16 MADSIM_BENCH_STATE_FLAGS=0x200 MADSIM_BENCH_LPW=64
  timeout 300 python bench.py --steps 20 --warmup 50 --no-cpu-baseline --no-measure-traffic --no-first-fail --no-extras > "$O/x.json" 2> "$O/x.err"; line "$O/x.json" "pingpong r6 (this tree)  r$round"
  ( cd _r5tree && timeout 300 python bench.py --steps 20 --warmup 50 --no-cpu-baseline --no-measure-traffic --no-first-fail --no-extras > "../$O/y.json" 2> "../$O/y.err" ); line "$O/y.json" "pingpong r5 (8add2ed)    r$round"
done | tee "$O/ab.txt"
