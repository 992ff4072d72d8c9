#!/bin/bash
# Round profile on the GPU box (run through gpurun): bench line, rocprofv3 kernel stats, PMC passes.
# FETCH_SIZE and WRITE_SIZE do not fit one pass (TCC has 4 slots, they cost 3 + 2).
# Counters are collected in their own passes with --kernel-trace only (never with sys/hip/hsa traces).
# Outputs land in gpurun_out/profile/; tools/summarize_profile.py turns them into profiles/<round>_*.
set -u
R=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out/profile
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python bench.py --steps 5 --warmup 1 > $OUT/bench_line.json 2> $OUT/bench.err
export TMPDIR=/tmp
BENCH="python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline"
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $BENCH > $OUT/stats.log 2>&1)
i=0
for G in "FETCH_SIZE" "WRITE_SIZE" "GRBM_COUNT GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $G --output-format csv -d $OUT/pmc$i -- \
      python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 0 --no-cpu-baseline > $OUT/pmc$i.log 2>&1)
done
python tools/summarize_profile.py $OUT $R > $OUT/summary.log 2>&1
cat $OUT/bench_line.json; tail -5 $OUT/summary.log
