#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r2l; mkdir -p $O
python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > $O/gputest.txt; cat $O/gputest.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
for wl in raft kv topo timers; do
  timeout 300 python bench.py --workload $wl --steps 200 --warmup 10 --no-cpu-baseline --no-measure-traffic > $O/$wl.json 2> $O/$wl.err
done
for f in raft kv topo timers; do python -c "
import json
d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1]); e=d['extra']
print('$f', 'ms/step', round(d['ms_per_step'],3), round(e['executor_steps_per_sec']/1e9,3), 'Gsteps/s', round(e['seeds_per_sec']/1e6,3), 'Mseeds/s', 'verified', d['verified_seeds'], 'failed', e['failed_seeds'], 'frac', round(d['roofline']['frac'],4), 'lds', e['lds_bytes_per_seed'], 'waves', e['waves_per_cu'])
"; done
