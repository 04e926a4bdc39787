#!/bin/bash
# Round 6: generator outputs counted in a 32-bit lane register inside the draw loops, folded into the 64-bit result field every 16th pass (the product)
# against a 64-bit add per with() (libmadsim_hip_prev.so: the tree one commit earlier).  Headline workload, driver-sized regions; timer storm; KV.
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
PV=MADSIM_HIP_LIB=$PWD/madsim_amd/libmadsim_hip_prev.so
timeout 1200 python -m pytest tests -m gpu -x -q > "$O/pytest.txt" 2>&1; tail -2 "$O/pytest.txt"
for round in 1 2 3 4; do
  timeout 300 python bench.py --steps 20 --warmup 50 --no-cpu-baseline --no-measure-traffic --no-first-fail --no-extras > "$O/x.json" 2> "$O/x.err"; line "$O/x.json" "pingpong 32-bit count r$round"
  env $PV timeout 300 python bench.py --steps 20 --warmup 50 --no-cpu-baseline --no-measure-traffic --no-first-fail --no-extras > "$O/x.json" 2> "$O/x.err"; line "$O/x.json" "pingpong previous     r$round"
  for wl in timers kv topo; do
    timeout 300 python bench.py --workload $wl --steps 24 --warmup 4 --no-cpu-baseline --no-measure-traffic --no-first-fail --no-extras > "$O/x.json" 2> "$O/x.err"; line "$O/x.json" "$wl 32-bit count r$round"
    env $PV timeout 300 python bench.py --workload $wl --steps 24 --warmup 4 --no-cpu-baseline --no-measure-traffic --no-first-fail --no-extras > "$O/x.json" 2> "$O/x.err"; line "$O/x.json" "$wl previous     r$round"
  done
done | tee "$O/ab.txt"
