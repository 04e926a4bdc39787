#!/bin/bash
# Round-2 GPU call B: parity suite + bench lines after the feature-class kernel split.
cd $GRAFT_REPO_ROOT; O=gpurun_out/r2b; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1; tail -5 $O/pytest.log
timeout 600 python bench.py --steps 100 --warmup 10 > $O/bench_pingpong.json 2> $O/bench_pingpong.err; python -c "import json; d=json.load(open('$O/bench_pingpong.json')); print('pingpong', d['ms_per_step'], d['extra']['executor_steps_per_sec']/1e9, d['verified_seeds'])"
for wl in raft kv topo timers; do
  timeout 300 python bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_$wl.json 2> $O/bench_$wl.err
  python -c "import json,sys; d=json.load(open('$O/bench_$wl.json')); e=d['extra']; print('$wl', round(d['ms_per_step'],2), round(e['executor_steps_per_sec']/1e9,3), 'Gsteps/s verified', d['verified_seeds'], 'failed', e['failed_seeds'], 'lanes', e['lanes_per_wave'], 'waves/cu', e['waves_per_cu'], d['roofline']['kernel'])"
done
