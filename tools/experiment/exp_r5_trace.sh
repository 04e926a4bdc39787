# kernel-trace timeline of madsim_hip_run_batch(262 144): when do the four sub-launches start and end?
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r5h}; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python $R/tools/experiment/exp_r5_runbatch.py > $O/run.log 2>&1
python - $O <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
rows.sort()
sims = [r for r in rows if "sim_kernel" in r[2]]
print(len(rows), "dispatches,", len(sims), "sim_kernel")
# groups of launches that overlap: print the first few groups of 4
i = 30
t0 = sims[i][0]
for r in sims[i:i + 16]:
    print(f"start +{(r[0]-t0)/1e6:8.3f} ms  end +{(r[1]-t0)/1e6:8.3f} ms  dur {(r[1]-r[0])/1e6:6.3f} ms  queue {r[3]} stream {r[4]}")
PY
