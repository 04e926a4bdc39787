#!/bin/bash
cd $GRAFT_REPO_ROOT
for wl in kv raft topo; do
  tools/prof_workload.sh r2c/prof_$wl "--workload $wl --steps 3 --warmup 1" full
  tail -14 gpurun_out/r2c/prof_$wl/summary.txt
done
