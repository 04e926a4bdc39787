#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2g
for wl in raft kv topo pingpong; do echo "== $wl"; MADSIM_HIP_LIB=$PWD/madsim_amd/libmadsim_hip_prof.so timeout 200 python tools/phase_prof.py $wl 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r2g/phase.txt
cat gpurun_out/r2g/phase.txt | head -3
