#!/bin/bash
# Round-2 GPU call D: the N>1 launcher path end to end on one GPU (gloo hook), and a fuzz campaign on the final kernels.
cd $GRAFT_REPO_ROOT; O=gpurun_out/r2d; mkdir -p $O
MADSIM_BENCH_BACKEND=gloo timeout 300 python bench.py --gpus 2 --steps 20 --warmup 4 --no-cpu-baseline > $O/bench_gloo2.json 2> $O/bench_gloo2.err; tail -c 1500 $O/bench_gloo2.json; tail -3 $O/bench_gloo2.err
timeout 400 python tools/fuzz_campaign.py 300 7000000 > $O/fuzz.txt 2>&1; tail -3 $O/fuzz.txt
