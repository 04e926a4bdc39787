cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3s
for sched in 0 1; do for seeds in 196608 786432 3145728; do for st in 1 2; do
MADSIM_BENCH_SCHED=$sched timeout 300 python bench.py --seeds $seeds --streams $st --steps 6 --warmup 2 --no-cpu-baseline --no-measure-traffic --no-first-fail --no-extras > gpurun_out/r3s/s.json 2> gpurun_out/r3s/s.err
python - $sched $seeds $st <<'PY'
import json,sys
try:
    d=json.loads(open('gpurun_out/r3s/s.json').read().strip().splitlines()[-1]); e=d['extra']
    print('sched',sys.argv[1],'seeds',sys.argv[2],'streams',sys.argv[3],'ms/step',round(d['ms_per_step'],3),'Mseeds/s',round(e['seeds_per_sec']/1e6,2),'verified',d['verified_seeds'])
except Exception as ex: print('fail',sys.argv[1:],ex, open('gpurun_out/r3s/s.err').read()[-300:])
PY
done; done; done | tee gpurun_out/r3s/sched.txt
