#!/bin/bash
# Round-2 GPU call A: parity suite, the bench lines, and rocprof profiles of the extended-op workloads.
cd $GRAFT_REPO_ROOT; O=gpurun_out/r2a; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1; tail -5 $O/pytest.log
timeout 600 python bench.py > $O/bench_pingpong.json 2> $O/bench_pingpong.err; tail -c 600 $O/bench_pingpong.json
for wl in raft kv topo timers; do
  timeout 300 python bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_$wl.json 2> $O/bench_$wl.err
  python -c "import json,sys; d=json.load(open('$O/bench_$wl.json')); print('$wl', d['ms_per_step'], d['extra']['executor_steps_per_sec']/1e9, d['verified_seeds'], d['extra']['failed_seeds'])"
done
for wl in kv raft topo; do
  tools/prof_workload.sh r2a/prof_$wl "--workload $wl --steps 3 --warmup 1"
  tail -12 $O/prof_$wl/summary.txt
done
# work distribution: 4 x 65 536 on two streams vs one 262 144-seed launch, loss 0.01
timeout 200 python bench.py --loss 0.01 --steps 40 --warmup 4 --no-cpu-baseline --no-first-fail > $O/wq_4x64k_2streams.json 2>&1
timeout 200 python bench.py --loss 0.01 --seeds 262144 --streams 1 --steps 10 --warmup 1 --no-cpu-baseline --no-first-fail > $O/wq_256k_static.json 2>&1
MADSIM_BENCH_SCHED=1 timeout 200 python bench.py --loss 0.01 --seeds 262144 --streams 1 --steps 10 --warmup 1 --no-cpu-baseline --no-first-fail > $O/wq_256k_queue.json 2>&1
for f in wq_4x64k_2streams wq_256k_static wq_256k_queue; do python -c "import json; d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1]); print('$f', d['extra']['seeds_per_sec']/1e6, 'Mseeds/s', d['verified_seeds'])"; done
