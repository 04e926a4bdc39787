#!/bin/bash
# Focused PMC passes (issue / wait / ifetch) for one bench configuration.  Usage: prof_pmc2.sh <outdir> "<bench args>"
set -u
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$1; mkdir -p $OUT
CMD="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-measure-traffic ${2:-}"
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU" \
           "SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_IFETCH SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" \
           "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_BUSY_CYCLES SQ_LEVEL_WAVES SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_BUSY_CU_CYCLES"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pmc$i -o p -- $CMD > $OUT/pmc$i.log 2>&1
done
