#!/bin/bash
# One documented launcher for everything this repo runs on a GPU box (replaces the per-call scratch scripts of round 2).
#   gpurun --timeout 900 -- 'bash tools/gpu_round.sh <out-tag> <stage>...'
# Stages run in the order given; each writes under gpurun_out/<out-tag>/ and prints a one-line summary.
#   ubench            tools/ubench_issue (VALU / SALU issue ceilings by wall time, attainable HBM copy bandwidth)
#   vmem              tools/ubench_vmem (cost of divergent global-memory instructions, round-trip latency)
#   tests             python -m pytest tests -m gpu -x -q
#   smoke             __graft_entry__.smoke()
#   bench             python bench.py (the default line: headline + extra.workloads + first-fail legs)
#   bench:<args>      python bench.py <args with ',' for spaces>, e.g. bench:--workload,raft,--steps,12
#   line:<wl>[:<steps>]  one bench line of workload <wl> without the extra legs (fast A/B)
#   prof:<wl>[:full]  tools/prof_workload.sh on workload <wl> (kernel-trace --stats + PMC passes; full adds FETCH/WRITE)
#   phase:<wl>[:<n>[:<tag>]]  tools/phase_prof.py on workload <wl> (in-kernel s_memtime probes of an EXP_PROF build), n batches in flight, libmadsim_hip_<tag>.so
#   fuzz[:<seconds>[:<generators>]]  tools/fuzz_campaign.py with a clock-derived base seed
#   ab:<wl>:<steps>   every madsim_amd/libmadsim_hip*.so back to back on workload <wl> (A/B builds from tools/build_variant.sh)
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=$1; shift
O=gpurun_out/$TAG; mkdir -p "$O"
line() {   # print the fields that matter of a bench JSON line
  python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); e = d["extra"]; r = d.get("roofline", {})
    print(sys.argv[2], "ms/step", round(d["ms_per_step"], 3), round(e["executor_steps_per_sec"] / 1e9, 3), "Gsteps/s",
          round(e["seeds_per_sec"] / 1e6, 3), "Mseeds/s kernel_ms", round(e["kernel_ms_per_step"], 3), "verified", d["verified_seeds"],
          "failed", e["failed_seeds"], "bound", r.get("bound"), "frac", round(r.get("frac") or 0, 3))
except Exception as ex:
    print(sys.argv[2], "NO LINE:", ex)
PY
}
for st in "$@"; do
  IFS=: read -r kind a1 a2 a3 <<< "$st"
  case $kind in
    ubench)
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/ubench_issue.hip -o tools/ubench_issue 2> "$O/ubench_build.err" \
        && timeout 300 tools/ubench_issue > "$O/ubench_issue.txt" 2> "$O/ubench_issue.err"; tail -8 "$O/ubench_issue.txt";;
    vmem)
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/ubench_vmem.hip -o tools/ubench_vmem 2> "$O/vmem_build.err" \
        && timeout 400 tools/ubench_vmem > "$O/ubench_vmem.txt" 2> "$O/ubench_vmem.err"; tail -12 "$O/ubench_vmem.txt";;
    tests)
      timeout 1500 python -m pytest tests -m gpu -x -q > "$O/pytest.txt" 2>&1; tail -3 "$O/pytest.txt";;
    smoke)
      python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1;;
    bench)
      if [ -n "$a1" ]; then args=${a1//,/ }; else args=""; fi
      n=$(ls "$O" | grep -c '^bench[0-9]*\.json$')
      ( time timeout 900 python bench.py $args > "$O/bench$n.json" 2> "$O/bench$n.err" ) 2> "$O/bench$n.time"; line "$O/bench$n.json" "bench[$args]"; grep real "$O/bench$n.time";;
    line)
      timeout 400 python bench.py --workload "$a1" --steps "${a2:-20}" --warmup 4 --no-cpu-baseline --no-measure-traffic --no-first-fail --no-extras \
        > "$O/line_$a1.json" 2> "$O/line_$a1.err"; line "$O/line_$a1.json" "$a1";;
    prof)
      args=""; [ "$a1" != pingpong ] && args="--workload $a1"
      bash tools/prof_workload.sh "$TAG/prof_$a1" "$args" "$a2"; tail -14 "$O/prof_$a1/summary.txt";;
    phase)   # needs madsim_amd/libmadsim_hip_prof.so (tools/build_variant.sh prof -DEXP_PROF), built before the call
      # phase:<wl>[:<batches in flight>[:<lib tag>]] — 0 batches in flight = one launch alone; lib tag picks libmadsim_hip_<tag>.so (default prof)
      PROF_STREAMS=${a2:-0} MADSIM_HIP_LIB=$PWD/madsim_amd/libmadsim_hip_${a3:-prof}.so timeout 300 python tools/phase_prof.py "$a1" > "$O/phase_${a1}_${a2:-0}_${a3:-prof}.txt" 2>&1; cat "$O/phase_${a1}_${a2:-0}_${a3:-prof}.txt";;
    fuzz)
      timeout $(( ${a1:-90} + 120 )) python tools/fuzz_campaign.py "${a1:-90}" "$(( $(date +%s) * 1000 ))" $a2 > "$O/fuzz.txt" 2>&1; tail -2 "$O/fuzz.txt";;
    ab)
      for round in 1 2 3; do for lib in madsim_amd/libmadsim_hip*.so; do
        MADSIM_HIP_LIB=$PWD/$lib timeout 300 python bench.py --workload "$a1" --steps "${a2:-20}" --warmup 4 --no-cpu-baseline --no-measure-traffic --no-first-fail --no-extras \
          > "$O/ab.json" 2> "$O/ab.err"; line "$O/ab.json" "$lib r$round"
      done; done | tee -a "$O/ab.txt";;
    *) echo "unknown stage $st";;
  esac
done
