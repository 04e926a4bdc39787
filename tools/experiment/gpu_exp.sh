#!/bin/bash
# scratch experiment runner: each line of $1 = "<tag> <bench args...>"
cd $GRAFT_REPO_ROOT; O=gpurun_out/exp; mkdir -p $O
while read -r tag envs args; do
  [ -z "$tag" ] && continue
  # envs: comma-separated VAR=value list or "-" (e.g. MADSIM_HIP_LIB=madsim_amd/libmadsim_hip_g3.so,MADSIM_BENCH_HEAP_LDS=8)
  if [ "$envs" != "-" ]; then for kv in ${envs//,/ }; do export "$kv"; done; fi
  [ -n "$MADSIM_HIP_LIB" ] && export MADSIM_HIP_LIB=$(realpath $MADSIM_HIP_LIB)
  timeout 100 python bench.py --no-cpu-baseline --no-first-fail --no-measure-traffic --no-extras $args > $O/$tag.json 2> $O/$tag.err
  python -c "import json,sys; d=json.loads(open('$O/$tag.json').read().strip().splitlines()[-1]); e=d['extra']; print('$tag', 'ms/step', round(d['ms_per_step'],3), round(e['executor_steps_per_sec']/1e9,3), 'Gsteps/s', round(e['seeds_per_sec']/1e6,3), 'Mseeds/s kernel_ms', round(e['kernel_ms_per_step'],3), 'verified', d['verified_seeds'], 'failed', e['failed_seeds'], 'lanes', e['lanes_per_wave'], 'waves/cu', e['waves_per_cu'], 'B/seed', e['lds_bytes_per_seed'])" 2>&1 | tail -1
  if [ "$envs" != "-" ]; then for kv in ${envs//,/ }; do unset "${kv%%=*}"; done; fi
done < $1
