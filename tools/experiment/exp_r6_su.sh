#!/bin/bash
# Round 6 (VERDICT r5 #1b): a socket's header, owner, first registration and first message word in ONE 16-byte unit (libmadsim_hip_su.so:
# tools/build_variant.sh su -DMADSIM_SOCK_UNIT=1) against four plane words, narrow heap entries both; one box, three interleaved rounds.
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
SU=MADSIM_HIP_LIB=$PWD/madsim_amd/libmadsim_hip_su.so
for round in 1 2 3; do
  run "topo narrow31 planes  r$round" topo 16 $N MADSIM_BENCH_HEAP_LDS=31
  run "topo narrow31 sockunit r$round" topo 16 $N MADSIM_BENCH_HEAP_LDS=31 $SU
  run "raft narrow l64 planes r$round" raft 16 $N MADSIM_BENCH_LPW=64
  run "raft narrow l64 sockunit r$round" raft 16 $N MADSIM_BENCH_LPW=64 $SU
  run "kv planes             r$round" kv 24 X=1
  run "kv sockunit           r$round" kv 24 $SU
done | tee "$O/ab.txt"
