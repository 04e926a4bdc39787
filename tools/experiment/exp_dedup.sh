#!/bin/bash
# A/B of MADSIM_STATE_DEDUP_TIMERS on the election loop (one box, three interleaved rounds):
#   base  = madsim_amd/libmadsim_hip_base.so (the library before the switch existed: tools/build_baseline.sh <rev> base)
#   off   = this tree's library, switch off      on = this tree's library, switch on
# Usage (GPU box): bash tools/experiment/exp_dedup.sh <out-tag> [workload=raft] [steps=12]
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/$1; mkdir -p "$O"; WL=${2:-raft}; ST=${3:-12}
run() {   # tag, lib, flags
  MADSIM_HIP_LIB=$2 MADSIM_BENCH_STATE_FLAGS=$3 timeout 300 python bench.py --workload $WL --steps $ST --warmup 4 --no-cpu-baseline --no-measure-traffic --no-first-fail --no-extras > "$O/l.json" 2> "$O/l.err"
  python - "$O/l.json" "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); e = d["extra"]
    print(sys.argv[2], "ms/step", round(d["ms_per_step"], 3), round(e["executor_steps_per_sec"] / 1e9, 3), "Gsteps/s", round(e["seeds_per_sec"] / 1e6, 3),
          "Mseeds/s kernel_ms", round(e["kernel_ms_per_step"], 3), "verified", d["verified_seeds"], "failed", e["failed_seeds"])
except Exception as ex:
    print(sys.argv[2], "NO LINE:", ex)
PY
}
for r in 1 2 3; do
  [ -f madsim_amd/libmadsim_hip_base.so ] && run "base r$r" "$PWD/madsim_amd/libmadsim_hip_base.so" 0
  run "off  r$r" "$PWD/madsim_amd/libmadsim_hip.so" -0x100
  run "on   r$r" "$PWD/madsim_amd/libmadsim_hip.so" 0x100
done | tee "$O/ab_dedup_$WL.txt"
