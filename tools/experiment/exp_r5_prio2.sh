# priority schedules: quarters (libmadsim_hip.so) / the last 1/2, 1/4, 1/8 (ps1) / the last 1/4, 1/8, 1/16 (ps2) / none (MADSIM_HIP_NO_PRIO)
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r5q}; mkdir -p $O
B="--no-cpu-baseline --no-measure-traffic --no-first-fail --no-extras --warmup 10"
for r in 1 2; do for lib in libmadsim_hip.so libmadsim_hip_ev4.so libmadsim_hip_ev64.so; do
  if [ $lib = off ]; then export MADSIM_HIP_NO_PRIO=1 MADSIM_HIP_LIB=$PWD/madsim_amd/libmadsim_hip.so; else unset MADSIM_HIP_NO_PRIO; export MADSIM_HIP_LIB=$PWD/madsim_amd/$lib; fi
  for n in 20 200; do
    timeout 200 python bench.py $B --steps $n > $O/x.json 2> $O/x.err; python tools/experiment/line.py $O/x.json "$lib steps=$n r$r"
  done
  python tools/experiment/exp_r5_runbatch.py 2>&1 | grep -E "^(null|touched) |count " | sed -n '1,2p;7,11p' | tr '\n' ';'; echo
done; done
