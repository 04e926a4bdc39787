# cap on the timer pops of one pass in the global-state builds (MADSIM_FIRE_CAP, k_main.h / k_net.h): libmadsim_hip.so = no cap
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r5v}; mkdir -p $O
B="--no-cpu-baseline --no-measure-traffic --no-first-fail --no-extras --warmup 4 --steps 12"
for r in 1 2; do for w in topo raft kv; do for lib in libmadsim_hip.so libmadsim_hip_cap3.so libmadsim_hip_cap2.so libmadsim_hip_cap1.so; do
  MADSIM_HIP_LIB=$PWD/madsim_amd/$lib timeout 300 python bench.py $B --workload $w > $O/x.json 2> $O/x.err; python tools/experiment/line.py $O/x.json "$w $lib r$r"
done; done; done
