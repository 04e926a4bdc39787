#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2h
python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r2h/gputest.txt; cat gpurun_out/r2h/gputest.txt
bash tools/gpu_exp.sh gpurun_exp.txt
