cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r5t}; mkdir -p $O
B="--no-cpu-baseline --no-measure-traffic --no-first-fail --no-extras --warmup 10"
for r in 1 2; do for s in 4 5 6 8; do for n in 20 200; do
  timeout 200 python bench.py $B --steps $n --streams $s > $O/x.json 2> $O/x.err; python tools/experiment/line.py $O/x.json "streams=$s steps=$n r$r"
done; done; done
