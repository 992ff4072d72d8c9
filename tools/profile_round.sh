#!/bin/bash
# rocprofv3 evidence for profiles/ (round 3).  Run on the GPU box:  bash tools/profile_round.sh [--quick]
#   1. kernel stats of the default bench command
#   2. PMC passes, ONE counter group per pass, --kernel-trace only (never combined with other trace domains):
#        VALU issue:  SQ_INSTS_VALU (instructions), SQ_ACTIVE_INST_VALU (quad-cycles the VALU works: = instructions when every
#                     instruction takes one 4-cycle window, which is the packed-f16 recurrence), SQ_ACTIVE_INST_VALU2
#                     (quad-cycles in which TWO 2-cycle-class instructions issued), GRBM_GUI_ACTIVE
#        occupancy / stalls: SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVES
#        LDS / scalar / memory instructions, then the three TCC passes (HBM traffic)
#   3. FETCH_SIZE / WRITE_SIZE calibration on known byte counts at the access widths the kernels use (tools/ubench/fetch_calib.hip)
# tools/summarise_profiles.py turns the output into profiles/r03_*.csv and profiles/hbm_traffic.json.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_r03
if [ "$2" = "stats-only" ]; then rm -rf $O/stats; else rm -rf $O; fi; mkdir -p $O
CMD="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-verify --no-cold $1"
# kernel stats: the command the driver runs, unabridged (the PMC passes below use fewer steps and skip the host-side sections)
timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --gpus 1 --steps 20 --warmup 5 $1 > $O/bench_under_stats.json 2> $O/stats.err
if [ "$2" = "stats-only" ]; then find $O -name "*.csv" | wc -l; exit 0; fi
i=0
for grp in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 SQ_WAVES GRBM_GUI_ACTIVE" \
           "SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 900 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/pmc$i -- $CMD > $O/bench_under_pmc$i.json 2> $O/pmc$i.err
  echo "pass $i ($grp) rc=$?"
done
for grp in FETCH_SIZE WRITE_SIZE "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/calib$i -- ./tools/ubench/fetch_calib.bin > $O/calib$i.txt 2> $O/calib$i.err
  echo "calibration pass $i ($grp) rc=$?"
done
find $O -name "*.csv" | wc -l
du -sh $O
