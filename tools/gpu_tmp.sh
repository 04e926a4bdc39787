#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/prof_workload.sh r2j_pp "" full
bash tools/prof_workload.sh r2j_kv "--workload kv"
bash tools/prof_workload.sh r2j_topo "--workload topo"
for d in r2j_pp r2j_kv r2j_topo; do echo "== $d"; head -12 gpurun_out/$d/summary.txt; done
