import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); e = d["extra"]
    print(sys.argv[2], "ms/step", round(d["ms_per_step"], 3), round(e["executor_steps_per_sec"] / 1e9, 3), "Gsteps/s", "verified", d["verified_seeds"], "failed", e["failed_seeds"])
except Exception as ex:
    print(sys.argv[2], "NO LINE:", ex)
