# cache policy of the timer-heap spill region's accesses (MADSIM_SPILL_LD_AUX / _ST_AUX: 2 = nt, 16 = sc1); libmadsim_hip.so = default policy
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r5w}; mkdir -p $O
B="--no-cpu-baseline --no-measure-traffic --no-first-fail --no-extras --warmup 4 --steps 12"
for r in 1 2; do for w in topo raft timers; do for lib in libmadsim_hip.so libmadsim_hip_nt.so libmadsim_hip_ntst.so libmadsim_hip_sc1.so; do
  MADSIM_HIP_LIB=$PWD/madsim_amd/$lib timeout 300 python bench.py $B --workload $w > $O/x.json 2> $O/x.err; python tools/experiment/line.py $O/x.json "$w $lib r$r"
done; done; done
