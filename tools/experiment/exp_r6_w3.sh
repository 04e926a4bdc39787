#!/bin/bash
# Round 6: every-class global-state build at three waves per SIMD (168 VGPRs, ~40 spilled) against two (243 VGPRs), narrow heap entries both.
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
for round in 1 2; do
  run "topo narrow31 2w        r$round" topo 16 $N MADSIM_BENCH_HEAP_LDS=31
  run "topo narrow   3w (w3)   r$round" topo 16 $N MADSIM_BENCH_HEAP_LDS=31 MADSIM_HIP_LIB=$PWD/madsim_amd/libmadsim_hip_w3.so
  run "topo wide     3w (w3)   r$round" topo 16 X=1 MADSIM_HIP_LIB=$PWD/madsim_amd/libmadsim_hip_w3.so
done | tee "$O/ab.txt"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -s -k "campaign_over_ranks or latency or hard_model or port0 or lifecycle_reference_tests_gpu" > "$O/pytest.txt" 2>&1; tail -5 "$O/pytest.txt"
( time timeout 900 python bench.py --steps 20 > "$O/bench_default20.json" 2> "$O/bench_default20.err" ) 2> "$O/bench.time"; tail -c 600 "$O/bench_default20.err"; cat "$O/bench.time" | grep real
python - "$O/bench_default20.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
print("ms/step",d["ms_per_step"],"value",d["value"],"host value",d.get("value_run_batch_host"))
print({k:r[k] for k in ("bound","achieved","peak","frac","traffic","launch_ms","launch_ms_hip_events","frac_hw","frac_hw_useful_lanes","frac_own_mix_ceiling","hbm_chip_frac")})
for k,v in (d["extra"].get("workloads") or {}).items(): print(k, round(v["steps_per_sec"]/1e9,3), "Gsteps/s measured_gbps", v.get("measured_gbps"), "chip", v.get("measured_chip_gbps"), "copy", v.get("copy_peak_gbps"), "ratio", v.get("measured_over_copy_peak"), "t/a", v.get("traffic_over_algorithmic"))
PY
