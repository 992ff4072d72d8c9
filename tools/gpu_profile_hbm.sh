#!/bin/bash
# the two TCC passes (FETCH_SIZE, WRITE_SIZE) of tools/gpu_profile.sh on their own
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_r02
mkdir -p $O
CMD="python bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-verify --no-cold"
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rm -rf $O/pmc$i
  timeout 420 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/pmc$i -- $CMD > $O/bench_under_pmc$i.json 2> $O/pmc$i.err
  echo "pass $grp rc=$?"
done
find $O/pmc1 $O/pmc2 -name "*.csv" | head
