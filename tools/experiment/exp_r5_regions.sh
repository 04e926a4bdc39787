# region length vs wall time per batch (headline kernel): is the 20-batch region's surcharge a constant (ramp + drain) or a rate?
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r5e}; mkdir -p $O
B="--no-cpu-baseline --no-measure-traffic --no-first-fail --no-extras --no-verify --warmup 10"
t() { tag=$1; shift; timeout 300 python bench.py $B "$@" > $O/$tag.json 2> $O/$tag.err; echo "$tag faults=$(grep -c 'Memory access fault' $O/$tag.err) $(python -c "
import json
d=json.loads(open('$O/$tag.json').read().strip().splitlines()[-1]); r=d['extra']['regions']
print('median', round(d['ms_per_step'],4), {k: (round(v,4) if isinstance(v,float) else v) for k,v in r.items() if not isinstance(v,list)})" 2>&1 | tail -1)"; }
for n in 5 10 20 40 80 200; do t s5_$n --steps $n --streams 5; done
t s8_200 --steps 200 --streams 8
t s5_1000 --steps 1000 --streams 5
