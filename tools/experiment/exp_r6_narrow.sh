#!/bin/bash
# Round 6: 8-byte timer-heap entries (MADSIM_STATE_NARROW_HEAP) against the 16-byte layout, same library, one box, three interleaved rounds.
#   gpurun --timeout 1500 -- 'bash tools/experiment/exp_r6_narrow.sh r6a'
# Every line carries oracle-verified seeds (bench.py's verify leg).  MADSIM_BENCH_STATE_FLAGS / MADSIM_BENCH_HEAP_LDS: bench.py's experiment hooks.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/$1; mkdir -p "$O"
line() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); e = d["extra"]
    print(sys.argv[2], "ms/step", round(d["ms_per_step"], 3), round(e["executor_steps_per_sec"] / 1e9, 3), "Gsteps/s kernel_ms", round(e["kernel_ms_per_step"], 3),
          "verified", d["verified_seeds"], "failed", e["failed_seeds"])
except Exception as ex:
    print(sys.argv[2], "NO LINE:", ex, open(sys.argv[1].replace(".json", ".err")).read()[-400:])
PY
}
run() {   # run <label> <workload> <steps> <env...>
  label=$1; wl=$2; steps=$3; shift 3
  env "$@" timeout 400 python bench.py --workload "$wl" --steps "$steps" --warmup 4 --no-cpu-baseline --no-measure-traffic --no-first-fail --no-extras \
    > "$O/x.json" 2> "$O/x.err"; line "$O/x.json" "$label"
}
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "narrow" > "$O/pytest_narrow.txt" 2>&1; tail -3 "$O/pytest_narrow.txt"
for round in 1 2 3; do
  run "topo wide15        r$round" topo 16 X=1
  run "topo narrow15      r$round" topo 16 MADSIM_BENCH_STATE_FLAGS=0x200
  run "topo narrow24      r$round" topo 16 MADSIM_BENCH_STATE_FLAGS=0x200 MADSIM_BENCH_HEAP_LDS=24
  run "topo narrow31      r$round" topo 16 MADSIM_BENCH_STATE_FLAGS=0x200 MADSIM_BENCH_HEAP_LDS=31
  run "raft wide22 l32    r$round" raft 16 X=1
  run "raft narrow22 l32  r$round" raft 16 MADSIM_BENCH_STATE_FLAGS=0x200
  run "raft narrow44 l32  r$round" raft 16 MADSIM_BENCH_STATE_FLAGS=0x200 MADSIM_BENCH_HEAP_LDS=44
  run "raft narrow22 l64  r$round" raft 16 MADSIM_BENCH_STATE_FLAGS=0x200 MADSIM_BENCH_LPW=64
done | tee "$O/ab.txt"
