# the topology (every-class global-state build, two waves per SIMD) with the re-registered Sleep timers as counts against the literal heap
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r5j}; mkdir -p $O
B="--no-cpu-baseline --no-measure-traffic --no-first-fail --no-extras --warmup 4 --workload topo --steps 16"
for r in 1 2 3; do
  timeout 300 python bench.py $B > $O/on.json 2> $O/on.err;  python tools/experiment/line.py $O/on.json "dedup on  r$r"
  MADSIM_BENCH_STATE_FLAGS=-0x100 timeout 300 python bench.py $B > $O/off.json 2> $O/off.err; python tools/experiment/line.py $O/off.json "dedup off r$r"
done
for h in 12 15 18 22; do
  MADSIM_BENCH_HEAP_LDS=$h timeout 300 python bench.py $B > $O/h$h.json 2> $O/h$h.err; python tools/experiment/line.py $O/h$h.json "dedup on, heap_lds $h"
done
