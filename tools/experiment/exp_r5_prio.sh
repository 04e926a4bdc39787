# progress-based wave priority (k_main.h): finite sets of launches should finish together.  A/B through MADSIM_HIP_NO_PRIO.
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r5p}; mkdir -p $O
B="--no-cpu-baseline --no-measure-traffic --no-first-fail --no-extras --warmup 10"
for r in 1 2 3; do for off in 0 1; do
  if [ $off = 1 ]; then export MADSIM_HIP_NO_PRIO=1; else unset MADSIM_HIP_NO_PRIO; fi
  for n in 20 200; do
    timeout 200 python bench.py $B --steps $n > $O/x.json 2> $O/x.err; python tools/experiment/line.py $O/x.json "prio_off=$off steps=$n r$r"
  done
  python tools/experiment/exp_r5_runbatch.py 2>&1 | grep -E "^(null|touched) " | head -2
done; done
unset MADSIM_HIP_NO_PRIO
for w in kv raft topo timers; do for off in 0 1; do
  if [ $off = 1 ]; then export MADSIM_HIP_NO_PRIO=1; else unset MADSIM_HIP_NO_PRIO; fi
  timeout 300 python bench.py $B --workload $w --steps 12 > $O/x.json 2> $O/x.err; python tools/experiment/line.py $O/x.json "$w prio_off=$off"
done; done
