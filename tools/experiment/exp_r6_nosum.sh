#!/bin/bash
# Round 6: what does the report reduction (summary_kernel + keyflip behind every launch, on the launch's stream) cost?  libmadsim_hip_nosum.so skips it
# (-DMADSIM_EXP_NO_SUMMARY: the line's step counts are then zero — ms/step is the figure).
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/$1; mkdir -p "$O"
line() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); e = d["extra"]
    print(sys.argv[2], "ms/step", round(d["ms_per_step"], 4), "kernel_ms", round(e["kernel_ms_per_step"], 3))
except Exception as ex:
    print(sys.argv[2], "NO LINE:", ex, open(sys.argv[1].replace(".json", ".err")).read()[-400:])
PY
}
run() { label=$1; wl=$2; steps=$3; shift 3
  env "$@" timeout 400 python bench.py --workload "$wl" --steps "$steps" --warmup 4 --no-cpu-baseline --no-measure-traffic --no-first-fail --no-extras --no-verify \
    > "$O/x.json" 2> "$O/x.err"; line "$O/x.json" "$label"; }
N=MADSIM_HIP_LIB=$PWD/madsim_amd/libmadsim_hip_nosum.so
for round in 1 2; do
  for spec in pingpong:20 topo:16 raft:12 kv:24 timers:40; do
    IFS=: read -r wl st <<< "$spec"
    run "$wl product r$round" $wl $st X=1
    run "$wl nosum   r$round" $wl $st $N
  done
done | tee "$O/ab.txt"
