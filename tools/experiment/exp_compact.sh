# The compact base-op layout (MADSIM_STATE_COMPACT, auto for the 4-node ping-pong): 152 B per seed, four waves per SIMD.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3v
run() {  # lib streams state_mem
MADSIM_HIP_LIB=$PWD/madsim_amd/$1 timeout 300 python bench.py --streams $2 --state-mem $3 --steps 200 --warmup 16 --no-cpu-baseline --no-measure-traffic --no-first-fail --no-extras > gpurun_out/r3v/s.json 2> gpurun_out/r3v/s.err
python - "$@" <<'PY'
import json,sys
try:
    d=json.loads(open('gpurun_out/r3v/s.json').read().strip().splitlines()[-1]); e=d['extra']
    print(*sys.argv[1:],'ms/step',round(d['ms_per_step'],4),'Mseeds/s',round(e['seeds_per_sec']/1e6,2),'lds',e['lds_bytes_per_seed'],'waves/cu',e['waves_per_cu'],'kernel_ms',round(e['kernel_ms_per_step'],3),'verified',d['verified_seeds'],d['roofline'].get('kernel'))
except Exception as ex: print('fail',sys.argv[1:],ex, open('gpurun_out/r3v/s.err').read()[-500:])
PY
}
for rep in 1 2; do
run libmadsim_hip_base.so 3 0
run libmadsim_hip.so 3 1
run libmadsim_hip.so 3 0
run libmadsim_hip.so 4 0
run libmadsim_hip.so 5 0
done | tee gpurun_out/r3v/compact.txt
