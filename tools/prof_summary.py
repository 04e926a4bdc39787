#!/usr/bin/env python3
"""Summarise a tools/prof_pmc.sh output directory: kernel stats + per-dispatch PMC averages for sim_kernel."""
import csv
import glob
import os
import sys
from collections import defaultdict

d = sys.argv[1]
out = []
ks = os.path.join(d, "trace", "t_kernel_stats.csv")
if os.path.exists(ks):
    out.append("== rocprofv3 --kernel-trace --stats (t_kernel_stats.csv) ==")
    out.append(open(ks).read().strip())
vals = defaultdict(list)
for f in sorted(glob.glob(os.path.join(d, "pmc*", "p_counter_collection.csv"))):
    for row in csv.DictReader(open(f)):
        if "sim_kernel" not in row["Kernel_Name"]:
            continue
        vals[row["Counter_Name"]].append(float(row["Counter_Value"]))
        grid, wg, lds = row.get("Grid_Size"), row.get("Workgroup_Size"), row.get("LDS_Block_Size")
        vg, sg = row.get("VGPR_Count"), row.get("SGPR_Count")
out.append("\n== PMC counters, sim_kernel, mean per dispatch (n dispatches) ==")
out.append(f"grid={grid} workgroup={wg} lds_block={lds} vgpr={vg} sgpr={sg}")
for k in sorted(vals):
    v = vals[k]
    out.append(f"{k:28s} {sum(v)/len(v):18.1f}  (n={len(v)})")
m = {k: sum(v) / len(v) for k, v in vals.items()}
out.append("\n== derived ==")
if "SQ_THREAD_CYCLES_VALU" in m and "SQ_ACTIVE_INST_VALU" in m and m["SQ_ACTIVE_INST_VALU"]:
    out.append(f"VALU lane utilisation (THREAD_CYCLES_VALU / (ACTIVE_INST_VALU*64)) = {m['SQ_THREAD_CYCLES_VALU'] / (m['SQ_ACTIVE_INST_VALU'] * 64):.3f}")
if "SQ_WAVE_CYCLES" in m:
    for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_SCA"):
        if k in m:
            out.append(f"{k} / SQ_WAVE_CYCLES = {m[k] / m['SQ_WAVE_CYCLES']:.3f}")
if "SQ_INSTS_VALU" in m and "SQ_WAVES" in m:
    out.append(f"VALU insts per wave = {m['SQ_INSTS_VALU'] / m['SQ_WAVES']:.0f}; LDS insts per wave = {m.get('SQ_INSTS_LDS', 0) / m['SQ_WAVES']:.0f}; SALU per wave = {m.get('SQ_INSTS_SALU', 0) / m['SQ_WAVES']:.0f}")
if "SQC_ICACHE_REQ" in m and m["SQC_ICACHE_REQ"]:
    out.append(f"instruction cache: {m['SQC_ICACHE_MISSES'] / m['SQC_ICACHE_REQ']:.4f} misses per request ({m['SQC_ICACHE_MISSES']:.0f} of {m['SQC_ICACHE_REQ']:.0f}); "
               f"SQ_IFETCH / SQ_WAVE_CYCLES = {m.get('SQ_IFETCH', 0) / m['SQ_WAVE_CYCLES']:.4f}" if "SQ_WAVE_CYCLES" in m else "")
if "TCC_HIT_sum" in m and (m["TCC_HIT_sum"] + m.get("TCC_MISS_sum", 0)):
    out.append(f"L2 hit rate = {m['TCC_HIT_sum'] / (m['TCC_HIT_sum'] + m['TCC_MISS_sum']):.3f} ({m['TCC_HIT_sum']:.0f} hits, {m['TCC_MISS_sum']:.0f} misses)")
if "FETCH_SIZE" in m:
    out.append(f"FETCH_SIZE = {m['FETCH_SIZE']:.1f} KB/dispatch (gfx950: doubles for wide coalesced reads, MI355X_MICROARCH.md §HBM)")
if "WRITE_SIZE" in m:
    out.append(f"WRITE_SIZE = {m['WRITE_SIZE']:.1f} KB/dispatch")
print("\n".join(out))

if "FETCH_SIZE" in m and "WRITE_SIZE" in m and len(sys.argv) > 2:
    import json
    json.dump({"FETCH_SIZE_KB": m["FETCH_SIZE"], "WRITE_SIZE_KB": m["WRITE_SIZE"], "source": d,
               "note": "mean per sim_kernel dispatch, rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes"},
              open(sys.argv[2], "w"), indent=1)
