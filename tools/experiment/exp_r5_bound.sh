# upper bound of what any decoupling of xoshiro generation from consumption could buy the base-op kernel: every rejection loop accepts its
# first output (libmadsim_hip_aa.so = tools/build_variant.sh aa -DEXP_ALWAYS_ACCEPT: NOT bit-exact, a timing experiment only)
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r5i}; mkdir -p $O
B="--no-cpu-baseline --no-measure-traffic --no-first-fail --no-extras --no-verify --warmup 10"
for r in 1 2; do for lib in libmadsim_hip.so libmadsim_hip_aa.so; do for n in 20 200; do
  MADSIM_HIP_LIB=$PWD/madsim_amd/$lib timeout 200 python bench.py $B --steps $n > $O/x.json 2> $O/x.err
  python -c "
import json; d=json.loads(open('$O/x.json').read().strip().splitlines()[-1]); print('$lib steps $n round $r: ms/step', round(d['ms_per_step'],4), 'G steps/s', round(d['extra']['executor_steps_per_sec']/1e9,2))"
done; done; done
