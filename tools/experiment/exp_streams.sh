# More batches in flight than waves fit (4-node ping-pong: three waves per SIMD by LDS): does a queued launch fill the tails?
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3u
for rep in 1 2; do for st in 3 4 5 6; do
timeout 300 python bench.py --streams $st --steps 120 --warmup 12 --no-cpu-baseline --no-measure-traffic --no-first-fail --no-extras > gpurun_out/r3u/s.json 2> gpurun_out/r3u/s.err
python - $st <<'PY'
import json,sys
try:
    d=json.loads(open('gpurun_out/r3u/s.json').read().strip().splitlines()[-1]); e=d['extra']
    print('nodes 4 streams',sys.argv[1],'ms/step',round(d['ms_per_step'],4),'Mseeds/s',round(e['seeds_per_sec']/1e6,2),'Gsteps/s',round(e['executor_steps_per_sec']/1e9,2),'kernel_ms',round(e['kernel_ms_per_step'],3),'verified',d['verified_seeds'])
except Exception as ex: print('fail',sys.argv[1:],ex, open('gpurun_out/r3u/s.err').read()[-400:])
PY
done; done | tee gpurun_out/r3u/streams.txt
