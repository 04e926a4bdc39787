#!/bin/bash
# Round 6: narrow heap entries as sibling pairs (one 16-byte access per sift-down level) against rows of 8-byte entries (libmadsim_hip_rows.so:
# tools/build_variant.sh rows -DMADSIM_NH_PAIRS=0), one box, three interleaved rounds, every line with oracle-verified seeds.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/$1; mkdir -p "$O"
line() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); e = d["extra"]
    print(sys.argv[2], "ms/step", round(d["ms_per_step"], 3), round(e["executor_steps_per_sec"] / 1e9, 3), "Gsteps/s kernel_ms", round(e["kernel_ms_per_step"], 3),
          "verified", d["verified_seeds"], "failed", e["failed_seeds"], "waves/cu", e.get("waves_per_cu"), "lds/seed", e.get("lds_bytes_per_seed"))
except Exception as ex:
    print(sys.argv[2], "NO LINE:", ex, open(sys.argv[1].replace(".json", ".err")).read()[-400:])
PY
}
run() { label=$1; wl=$2; steps=$3; shift 3
  env "$@" timeout 400 python bench.py --workload "$wl" --steps "$steps" --warmup 4 --no-cpu-baseline --no-measure-traffic --no-first-fail --no-extras \
    > "$O/x.json" 2> "$O/x.err"; line "$O/x.json" "$label"; }
N=MADSIM_BENCH_STATE_FLAGS=0x200
ROWS=MADSIM_HIP_LIB=$PWD/madsim_amd/libmadsim_hip_rows.so
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "narrow" > "$O/pytest_narrow.txt" 2>&1; tail -3 "$O/pytest_narrow.txt"
for round in 1 2 3; do
  run "topo wide15           r$round" topo 16 X=1
  run "topo narrow31 pairs   r$round" topo 16 $N MADSIM_BENCH_HEAP_LDS=31
  run "topo narrow31 rows    r$round" topo 16 $N MADSIM_BENCH_HEAP_LDS=31 $ROWS
  run "raft wide22 l32       r$round" raft 16 X=1
  run "raft narrow l64 pairs r$round" raft 16 $N MADSIM_BENCH_LPW=64
  run "raft narrow l64 rows  r$round" raft 16 $N MADSIM_BENCH_LPW=64 $ROWS
  run "raft narrow43 l32 pairs r$round" raft 16 $N MADSIM_BENCH_HEAP_LDS=44
done | tee "$O/ab.txt"
