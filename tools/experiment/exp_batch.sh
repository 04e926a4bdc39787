# Global-state workloads at their BASELINE batch sizes: seeds per launch x work distribution x launches in flight.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3b
run() {  # workload seeds sched streams steps
MADSIM_BENCH_SCHED=$3 timeout 300 python bench.py --workload $1 --seeds $2 --streams $4 --steps $5 --warmup 2 --no-cpu-baseline --no-measure-traffic --no-first-fail --no-extras > gpurun_out/r3b/s.json 2> gpurun_out/r3b/s.err
python - "$@" <<'PY'
import json,sys
try:
    d=json.loads(open('gpurun_out/r3b/s.json').read().strip().splitlines()[-1]); e=d['extra']
    print(sys.argv[1],'seeds',sys.argv[2],'sched',sys.argv[3],'streams',sys.argv[4],'ms/step',round(d['ms_per_step'],3),'Gsteps/s',round(e['executor_steps_per_sec']/1e9,3),'Mseeds/s',round(e['seeds_per_sec']/1e6,3),'verified',d['verified_seeds'],'failed',e['failed_seeds'])
except Exception as ex: print('fail',sys.argv[1:],ex, open('gpurun_out/r3b/s.err').read()[-300:])
PY
}
{
run raft 65536 0 3 9
for seeds in 262144 786432; do for sched in 0 1; do for st in 1 2 3; do run raft $seeds $sched $st 6; done; done; done
run topo 65536 0 3 9
for sched in 0 1; do for st in 1 2; do run topo 524288 $sched $st 4; done; done
run kv 65536 0 3 12
for sched in 0 1; do for st in 1 2 3; do run kv 131072 $sched $st 12; done; done
} | tee gpurun_out/r3b/batch.txt
