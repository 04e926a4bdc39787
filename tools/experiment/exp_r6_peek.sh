#!/bin/bash
# Round 6: the determinism log's number (the output of the current generator state) kept in hand, and the draws that can start from it:
# MADSIM_RNG_PEEK = 2 (the product: every with() whose accepted output is not needed behind its loop) / 1 (the ready-queue draw only:
# libmadsim_hip_peek1.so) / 0 (computed twice, as until round 6: libmadsim_hip_nopeek.so).  Headline workload in driver-sized regions, timer storm.
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
timeout 1200 python -m pytest tests -m gpu -x -q > "$O/pytest.txt" 2>&1; tail -2 "$O/pytest.txt"
for round in 1 2 3 4; do
  for v in "peek 2 (product)|X=1" "peek 1           |MADSIM_HIP_LIB=$PWD/madsim_amd/libmadsim_hip_peek1.so" "no peek          |MADSIM_HIP_LIB=$PWD/madsim_amd/libmadsim_hip_nopeek.so"; do
    IFS='|' read -r name envv <<< "$v"
    env $envv timeout 300 python bench.py --steps 20 --warmup 50 --no-cpu-baseline --no-measure-traffic --no-first-fail --no-extras > "$O/x.json" 2> "$O/x.err"; line "$O/x.json" "pingpong $name r$round"
  done
  for v in "peek 2 (product)|X=1" "no peek          |MADSIM_HIP_LIB=$PWD/madsim_amd/libmadsim_hip_nopeek.so"; do
    IFS='|' read -r name envv <<< "$v"
    env $envv timeout 300 python bench.py --workload timers --steps 48 --warmup 4 --no-cpu-baseline --no-measure-traffic --no-first-fail --no-extras > "$O/x.json" 2> "$O/x.err"; line "$O/x.json" "timers   $name r$round"
  done
done | tee "$O/ab.txt"
