#!/bin/bash
# rocprofv3 evidence for one bench workload: kernel-trace stats + PMC passes (each counter group in its own run, as
# MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE cannot share a pass).  Usage (on the GPU box, repo root):
#   tools/profile_configs.sh <workload: c2|s4096_20hz|c3|c5> <tag> [steps]
# Writes gpurun_out/prof_<tag>_<workload>/{stats,pmc_sq,pmc_mops,pmc_fetch,pmc_write}/... and the bench line.
set -u
WL=$1; TAG=$2; STEPS=${3:-4}
OUT=$PWD/gpurun_out/prof_${TAG}_${WL}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $OLDPWD/bench.py --workload $WL --configs= --no-latency --no-cpu-baseline --paced-sec 0 --steps $STEPS --warmup 1"
$BENCH > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats -d $OUT/stats -o run -- $BENCH > $OUT/stats.log 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE -d $OUT/pmc_sq -o run -- $BENCH > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAVES -d $OUT/pmc_mops -o run -- $BENCH > $OUT/pmc_mops.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o run -- $BENCH > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o run -- $BENCH > $OUT/pmc_write.log 2>&1
cd $OLDPWD
find $OUT -name "*.csv" | head -20
# keep the merge small: drop the per-dispatch traces, keep stats + counter collections
find $OUT -name "*kernel_trace.csv" -delete
du -sh $OUT
