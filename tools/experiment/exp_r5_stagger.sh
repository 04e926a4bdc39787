# do launches that start together run faster when their workgroups start a few microseconds apart?  (MADSIM_HIP_STAGGER, k_main.h)
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r5g}; mkdir -p $O
B="--no-cpu-baseline --no-measure-traffic --no-first-fail --no-extras --warmup 10 --steps 20"
for s in 0 4 32 256 1024 0 32; do
  echo "== stagger $s"
  MADSIM_HIP_STAGGER=$s python tools/experiment/exp_r5_runbatch.py 2>&1 | grep -E "^(null|touched) " | head -2
  MADSIM_HIP_STAGGER=$s timeout 200 python bench.py $B > $O/b$s.json 2> $O/b$s.err; python -c "
import json; d=json.loads(open('$O/b$s.json').read().strip().splitlines()[-1]); print('bench 20-step regions: median', round(d['ms_per_step'],4), 'verified', d['verified_seeds'])"
done
