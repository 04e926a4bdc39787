#!/bin/bash
# Round 6: the headline kernel of this tree against round 5's (_r5tree = `git archive 8add2ed`, its own bench.py and library), same box, interleaved;
# then the whole GPU suite.
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
for round in 1 2 3 4; do
  timeout 300 python bench.py --steps 20 --warmup 50 --no-cpu-baseline --no-measure-traffic --no-first-fail --no-extras > "$O/x.json" 2> "$O/x.err"; line "$O/x.json" "pingpong r6 (this tree)  r$round"
  ( cd _r5tree && timeout 300 python bench.py --steps 20 --warmup 50 --no-cpu-baseline --no-measure-traffic --no-first-fail --no-extras > "../$O/y.json" 2> "../$O/y.err" ); line "$O/y.json" "pingpong r5 (8add2ed)    r$round"
done | tee "$O/ab.txt"
( time timeout 2400 python -m pytest tests -m gpu -x -q > "$O/pytest.txt" 2>&1 ) 2> "$O/pytest.time"; tail -5 "$O/pytest.txt"; grep real "$O/pytest.time"
