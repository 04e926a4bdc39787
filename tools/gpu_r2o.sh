#!/bin/bash
# Round-2 GPU call O (final): GPU suite, smoke(), the default bench line, the rocprofv3 summary of the final ping-pong kernel
# (after the single-exit restructuring), the other workloads' lines, a fuzz campaign with all nine generators.
cd $GRAFT_REPO_ROOT; O=gpurun_out/r2o; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; python -c "import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print('bench ms/step', d['ms_per_step'], 'verified', d['verified_seeds'], 'frac', d['roofline']['frac'], 'traffic', d['roofline']['traffic'], 'search', d['extra']['first_fail']['search_seeds_per_hour'])"
[ -n "$SKIP_PROF" ] || { bash tools/prof_workload.sh r2o/pp "" full; tail -4 $O/pp/summary.txt; }
for wl in "raft 40" "kv 300" "topo 40" "timers 200"; do set -- $wl
  timeout 300 python bench.py --workload $1 --steps $2 --warmup 6 --no-cpu-baseline --no-measure-traffic > $O/bench_$1.json 2> $O/bench_$1.err
  python -c "import json; d=json.loads(open('$O/bench_$1.json').read().strip().splitlines()[-1]); e=d['extra']; print('$1 ms/step', round(d['ms_per_step'],3), round(e['executor_steps_per_sec']/1e9,3), 'Gsteps/s', round(e['seeds_per_sec']/1e6,3), 'Mseeds/s verified', d['verified_seeds'], 'failed', e['failed_seeds'])"
done
timeout 200 python tools/fuzz_campaign.py ${FUZZ_S:-90} 15000000 > $O/fuzz.txt 2>&1; tail -2 $O/fuzz.txt
