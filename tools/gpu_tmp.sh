#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r2i; mkdir -p $O
timeout 600 python bench.py > $O/pingpong.json 2> $O/pingpong.err
for wl in raft kv topo timers; do
  timeout 300 python bench.py --workload $wl --steps 200 --warmup 10 --no-cpu-baseline --no-measure-traffic > $O/$wl.json 2> $O/$wl.err
done
timeout 200 python bench.py --streams 1 --steps 300 --warmup 20 --no-cpu-baseline --no-measure-traffic --no-first-fail > $O/pingpong_1stream.json 2> $O/pingpong_1stream.err
for f in pingpong raft kv topo timers pingpong_1stream; do python -c "
import json
d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1]); e=d['extra']
print('$f', 'ms/step', round(d['ms_per_step'],3), round(e['executor_steps_per_sec']/1e9,3), 'Gsteps/s', round(e['seeds_per_sec']/1e6,3), 'Mseeds/s', 'verified', d['verified_seeds'], 'failed', e['failed_seeds'], 'frac', d['roofline']['frac'], 'traffic', d['roofline'].get('traffic'), 'cpu', (d.get('cpu_baseline') or {}).get('value'))
"; done
