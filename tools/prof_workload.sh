#!/bin/bash
# rocprofv3 kernel-trace stats + the PMC passes that give instruction counts, VALU lane utilisation and wait fractions
# for one bench workload (run on the GPU box).  Usage: prof_workload.sh <outdir under gpurun_out> "<bench args>" [full]
# PMC passes are separate runs with --kernel-trace only (never combined with sys/hip traces).
set -u
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$1; mkdir -p $OUT
CMD="python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-verify --no-first-fail --no-extras --no-measure-traffic ${2:-}"
# the stats pass runs what bench.py's live trace pass runs (measure_counters "TRACE": 40 steps after 10, one region), so that the AverageNs of
# sim_kernel here and roofline.launch_ms in the bench line are the same measurement (round 6; the PMC passes below serialise launches and stay short)
TCMD="python $R/bench.py --steps 40 --warmup 10 --repeats 1 --no-cpu-baseline --no-verify --no-first-fail --no-extras --no-measure-traffic ${2:-}"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $TCMD > $OUT/trace.log 2>&1
SETS=("SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD"
      "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU")
if [ "${3:-}" = "full" ]; then
  SETS+=("SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"
         "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_BUSY_CYCLES SQ_IFETCH SQ_WAVE_CYCLES"      # the 110-134 KB every-class builds against a 64 KB instruction cache
         "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum")
fi
i=0
for set in "${SETS[@]}"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pmc$i -o p -- $CMD > $OUT/pmc$i.log 2>&1
done
python $R/tools/prof_summary.py $OUT $OUT/traffic.json > $OUT/summary.txt 2>&1
