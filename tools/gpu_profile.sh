#!/bin/bash
# rocprofv3 evidence for profiles/: kernel stats of the default bench command, then PMC passes (one counter group per
# pass, --kernel-trace only - never combined with other trace domains).  Run on the GPU box: bash tools/gpu_profile.sh
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_r02
rm -rf $O; mkdir -p $O
CMD="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-verify --no-cold"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $CMD > $O/bench_under_stats.json 2> $O/stats.err
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES GRBM_GUI_ACTIVE" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_WAIT_INST_ANY" "TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 400 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/pmc$i -- $CMD > $O/bench_under_pmc$i.json 2> $O/pmc$i.err
done
find $O -name "*.csv" | head -40
du -sh $O
