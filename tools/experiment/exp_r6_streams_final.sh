cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r6au; mkdir -p $O
for round in 1 2; do for spec in topo:3 topo:4 topo:5 topo:6 raft:4 raft:5 raft:6 kv:2 kv:3 kv:4; do IFS=: read -r wl n <<< "$spec"
  st=16; [ $wl = kv ] && st=24; [ $wl = raft ] && st=10
  timeout 300 python bench.py --workload $wl --steps $st --warmup 4 --streams $n --no-cpu-baseline --no-measure-traffic --no-first-fail --no-extras --no-verify > $O/x.json 2> $O/x.err
  python -c "import json; d=json.loads(open('$O/x.json').read().strip().splitlines()[-1]); e=d['extra']; print('$wl streams $n r$round', round(d['ms_per_step'],3), round(e['executor_steps_per_sec']/1e9,3))"
done; done | tee $O/streams.txt
