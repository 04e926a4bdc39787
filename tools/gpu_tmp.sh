#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/prof_workload.sh r2k_raft "--workload raft"
bash tools/prof_workload.sh r2k_topo "--workload topo"
bash tools/prof_workload.sh r2k_kv "--workload kv"
for d in r2k_raft r2k_topo r2k_kv; do echo "== $d"; grep -v "at::native\|rocclr\|summary_kernel\|keyflip" gpurun_out/$d/summary.txt | head -40; done
