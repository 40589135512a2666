#!/bin/bash
# rocprofv3 evidence for one bench workload: kernel-trace stats + PMC passes (each counter group in its own run, as
# MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE cannot share a pass).  Usage (on the GPU box, repo root):
#   tools/profile_configs.sh <workload: c2|s4096_20hz|c3|c5> <tag> [steps] [extra bench args] [output suffix]
# Keeps only the small summaries (the raw per-dispatch CSVs are hundreds of MB) under gpurun_out/prof_<tag>_<workload>/:
#   bench.json, kernel_stats.csv, pmc_sq.txt, pmc_mops.txt, pmc_fetch.txt, pmc_write.txt   (means per dispatch, per kernel)
set -u
WL=$1; TAG=$2; STEPS=${3:-4}; EXTRA=${4:-}
REPO=$PWD
SUF=${5:-}
OUT=$REPO/gpurun_out/prof_${TAG}_${WL}${SUF}
RAW=/tmp/prof_raw_$$
mkdir -p $OUT $RAW
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --workload $WL --configs= --no-latency --no-cpu-baseline --paced-sec 0 --steps $STEPS --warmup 1 $EXTRA"
python -c "import sys; sys.path.insert(0, '$REPO'); from vap_realtime_amd import provenance as p; import json; print(json.dumps({'csrc_sha': p.kernel_source_hash()}))" > $OUT/source.json
$BENCH > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats -f csv -d $RAW/stats -o run -- $BENCH > $OUT/stats.log 2>&1
cp $(find $RAW/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv 2>/dev/null
rm -rf $RAW/stats
pass() {  # name, counters...
  # Under --pmc every profiled dispatch is serialised: profiling all 251 window-fill ticks of C3 at 4096 streams takes half an hour per
  # pass.  For c3 only steady-state dispatches are profiled (--kernel-iteration-range): the window-dependent attention kernel at its last
  # 26 launches (ticks 254-259, window full), everything else (window-independent work per launch) at launches 250-259 of each kernel.
  local name=$1; shift
  : > $OUT/$name.txt
  if [ "$WL" = "c3" ]; then
    rocprofv3 --pmc "$@" --kernel-exclude-regex "attention_(long|proj)" --kernel-iteration-range "[250-259]" -f csv -d $RAW/${name}_a -o run -- $BENCH > $OUT/$name.log 2>&1
    case "$EXTRA" in
      *split-f16*)   # split path (round 5): attention_long_f16x3 runs 3 x per tick (layer 0 + the two cross-attentions), attention_proj_f16x3 2 x
        rocprofv3 --pmc "$@" --kernel-include-regex attention_long --kernel-iteration-range "[762-779]" -f csv -d $RAW/${name}_b -o run -- $BENCH >> $OUT/$name.log 2>&1
        rocprofv3 --pmc "$@" --kernel-include-regex attention_proj --kernel-iteration-range "[508-519]" -f csv -d $RAW/${name}_c -o run -- $BENCH >> $OUT/$name.log 2>&1 ;;
      *)
        rocprofv3 --pmc "$@" --kernel-include-regex attention_long --kernel-iteration-range "[1270-1295]" -f csv -d $RAW/${name}_b -o run -- $BENCH >> $OUT/$name.log 2>&1 ;;
    esac
  else
    rocprofv3 --pmc "$@" -f csv -d $RAW/${name}_a -o run -- $BENCH > $OUT/$name.log 2>&1
  fi
  for csv in $(find $RAW/${name}_a $RAW/${name}_b $RAW/${name}_c -name "*counter_collection.csv" 2>/dev/null); do python $REPO/tools/pmc_summary.py $csv kernel >> $OUT/$name.txt; done
  [ -s $OUT/$name.txt ] || echo "no counter csv" > $OUT/$name.txt
  tail -3 $OUT/$name.log > $OUT/$name.log.tail; rm -f $OUT/$name.log
  rm -rf $RAW/${name}_a $RAW/${name}_b $RAW/${name}_c
}
pass pmc_sq SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE
pass pmc_mops SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAVES
pass pmc_fetch FETCH_SIZE
pass pmc_write WRITE_SIZE
rm -rf $RAW
cd $REPO
du -sh $OUT
