#!/bin/bash
# Round 6: narrow heap entries at lower occupancy (more of the heap in LDS, smaller resident working set), one box, two interleaved rounds.
#   gpurun --timeout 1200 -- 'bash tools/experiment/exp_r6_occ.sh r6b'
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/$1; mkdir -p "$O"
line() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); e = d["extra"]
    print(sys.argv[2], "ms/step", round(d["ms_per_step"], 3), round(e["executor_steps_per_sec"] / 1e9, 3), "Gsteps/s kernel_ms", round(e["kernel_ms_per_step"], 3),
          "verified", d["verified_seeds"], "failed", e["failed_seeds"], "streams", e.get("streams"))
except Exception as ex:
    print(sys.argv[2], "NO LINE:", ex, open(sys.argv[1].replace(".json", ".err")).read()[-400:])
PY
}
run() { label=$1; wl=$2; steps=$3; shift 3
  env "$@" timeout 400 python bench.py --workload "$wl" --steps "$steps" --warmup 4 --no-cpu-baseline --no-measure-traffic --no-first-fail --no-extras \
    > "$O/x.json" 2> "$O/x.err"; line "$O/x.json" "$label"; }
N=MADSIM_BENCH_STATE_FLAGS=0x200
for round in 1 2; do
  run "topo narrow31 2w     r$round" topo 16 $N MADSIM_BENCH_HEAP_LDS=31
  run "topo narrow71 1w     r$round" topo 16 $N MADSIM_BENCH_HEAP_LDS=74 MADSIM_HIP_WAVES_PER_SIMD=1
  run "raft narrow20 l64 3w r$round" raft 16 $N MADSIM_BENCH_LPW=64
  run "raft narrow33 l64 2w r$round" raft 16 $N MADSIM_BENCH_LPW=64 MADSIM_BENCH_HEAP_LDS=34 MADSIM_HIP_WAVES_PER_SIMD=2
  run "raft narrow73 l64 1w r$round" raft 16 $N MADSIM_BENCH_LPW=64 MADSIM_BENCH_HEAP_LDS=74 MADSIM_HIP_WAVES_PER_SIMD=1
  run "raft narrow44 l32 3w r$round" raft 16 $N MADSIM_BENCH_HEAP_LDS=44
  run "raft narrow70 l32 2w r$round" raft 16 $N MADSIM_BENCH_HEAP_LDS=70 MADSIM_HIP_WAVES_PER_SIMD=2
  run "kv                   r$round" kv 16 X=1
done | tee "$O/ab.txt"
