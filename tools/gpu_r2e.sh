#!/bin/bash
# Round-2 GPU call E: the profiles that get committed under profiles/ (final kernels) + one bench line per workload.
cd $GRAFT_REPO_ROOT; O=gpurun_out/r2e; mkdir -p $O
tools/prof_workload.sh r2e/prof_pingpong "--steps 12 --warmup 4" full
for wl in kv raft topo; do tools/prof_workload.sh r2e/prof_$wl "--workload $wl --steps 3 --warmup 1" full; done
timeout 600 python bench.py > $O/bench_pingpong.json 2> $O/bench_pingpong.err
for wl in raft kv topo timers; do timeout 300 python bench.py --workload $wl --steps 10 --warmup 2 > $O/bench_$wl.json 2> $O/bench_$wl.err; done
for wl in pingpong raft kv topo timers; do python -c "import json; d=json.load(open('$O/bench_$wl.json')); e=d['extra']; print('$wl', round(d['ms_per_step'],3), 'ms/step', round(e['executor_steps_per_sec']/1e9,3), 'Gsteps/s', round(e['seeds_per_sec']/1e6,2), 'Mseeds/s verified', d['verified_seeds'], 'frac', round(d['roofline']['frac'],3), d['roofline']['kernel'])"; done
