#!/bin/bash
# HBM traffic of the heap-spill path: bench --workload timers with a small and a large LDS heap quota.
set -u
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_spill; mkdir -p $OUT
for h in 4 32; do
  CMD="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --streams 1 --workload timers --heap-lds $h"
  $CMD 2>&1 | tail -1 > $OUT/bench_h$h.json
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/h${h}_$c -o p -- $CMD > $OUT/h${h}_$c.log 2>&1
  done
done
python - <<'PY'
import csv, glob, json, os
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/prof_spill"
for h in (4, 32):
    b = json.load(open(f"{out}/bench_h{h}.json"))
    vals = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        f = glob.glob(f"{out}/h{h}_{c}/**/p_counter_collection.csv", recursive=True)[0]
        v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "sim_kernel" in r["Kernel_Name"] and r["Counter_Name"] == c]
        vals[c] = sum(v) / len(v)
    ms = b["extra"]["kernel_ms_per_step"]
    gb = (2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024 / 1e9
    print(f"heap_lds={h:2d}: kernel {ms:8.2f} ms/launch, {b['extra']['seeds_per_sec']/1e6:.2f} Mseeds/s, {b['extra']['executor_steps_per_sec']/1e9:.2f} Gsteps/s, lanes/wave {b['extra']['lanes_per_wave']}, "
          f"FETCH_SIZE {vals['FETCH_SIZE']/1024:.1f} MB (x2 gfx950 correction), WRITE_SIZE {vals['WRITE_SIZE']/1024:.1f} MB => {gb:.2f} GB/launch = {gb/(ms*1e-3):.0f} GB/s of HBM traffic")
PY
