# Does a 4th wave per SIMD pay?  The 2-node ping-pong needs 136 B of LDS per seed (4-node: 200 B), so four batches fit a CU.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3t
for rep in 1 2; do for st in 2 3 4 5; do
timeout 300 python bench.py --nodes 2 --streams $st --steps 60 --warmup 8 --no-cpu-baseline --no-measure-traffic --no-first-fail --no-extras > gpurun_out/r3t/s.json 2> gpurun_out/r3t/s.err
python - $st <<'PY'
import json,sys
try:
    d=json.loads(open('gpurun_out/r3t/s.json').read().strip().splitlines()[-1]); e=d['extra']
    print('nodes 2 streams',sys.argv[1],'ms/step',round(d['ms_per_step'],4),'Mseeds/s',round(e['seeds_per_sec']/1e6,2),'Gsteps/s',round(e['executor_steps_per_sec']/1e9,2),'lds',e['lds_bytes_per_seed'],'waves/cu',e['waves_per_cu'],'verified',d['verified_seeds'])
except Exception as ex: print('fail',sys.argv[1:],ex, open('gpurun_out/r3t/s.err').read()[-400:])
PY
done; done | tee gpurun_out/r3t/wave4.txt
