#!/bin/bash
# On the GPU box: bench every madsim_amd/libmadsim_hip*.so variant back to back (3 interleaved rounds).
cd $GRAFT_REPO_ROOT
for round in 1 2 3; do
  for lib in madsim_amd/libmadsim_hip*.so; do
    MADSIM_HIP_LIB=$PWD/$lib timeout 120 python bench.py --steps 10 --warmup 2 --no-cpu-baseline ${BENCH_ARGS:-} 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', round(d['extra']['kernel_ms_per_step'],3), 'ms', round(d['extra']['seeds_per_sec']/1e6,2), 'Mseeds/s')"
  done
done
