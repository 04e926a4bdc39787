#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2h
timeout 400 python tools/fuzz_campaign.py 240 5000000 2>&1 | grep -v amdgpu.ids | tail -4 > gpurun_out/r2h/fuzz3.txt; cat gpurun_out/r2h/fuzz3.txt
