#!/bin/bash
# Everything that wants an MI355X, in one gpurun call, most important first; every step under its own timeout (a hang costs one
# step, not the call):   bash tools/round6_gpu.sh [steps...]     steps: tests bench stats pmc concat first redzones rq dropin dasan (default: all)
# Output under gpurun_out/r06/ (merged back by gpurun); copy what is to be judged into profiles/r06_*.
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06; mkdir -p $O
export SWA_WATCHDOG_S=60
STEPS="${*:-tests bench stats pmc concat first redzones rq dropin dasan}"
for s in $STEPS; do
  t0=$(date +%s)
  case $s in
    tests)    timeout 1800 python -m pytest tests -m gpu -q --durations=25 > $O/tests.txt 2>&1; echo "rc=$?" >> $O/tests.txt; tail -4 $O/tests.txt ;;
    bench)    ( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err ) 2> $O/bench_time.txt; tail -c 600 $O/bench.json; tail -3 $O/bench_time.txt ;;
    stats)    rm -rf $O/stats; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --gpus 1 --steps 20 --warmup 5 --no-live-traffic > $O/bench_under_stats.json 2> $O/stats.err; echo "rc=$?"; find $O/stats -name "*kernel_stats.csv" | head -2 ;;
    pmc)      for c in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"; do n=$(echo $c | tr ' ' '_'); rm -rf $O/pmc_$n; timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$n -- python bench.py --traffic-child --nseq 10000000 > $O/pmc_$n.out 2>&1; echo "pmc $n rc=$?"; done ;;
    first)    SWA_LOAD_TRACE=1 SWA_CLI_TRACE=1 timeout 600 python tools/probe.py first > $O/first.txt 2>&1; grep -v "^swa load\|^    " $O/first.txt | tail -16 ;;
    redzones) SWA_REDZONES=1 timeout 900 python - > $O/redzones.txt 2>&1 <<'PY'
import re, sys, os
sys.path.insert(0, os.getcwd())
src = open("tests/test_gpu_loading.py").read()
code = re.search(r'_REDZONE_SCRIPT = r"""(.*?)"""', src, re.S).group(1) % (os.getcwd(), "/tmp")
exec(compile(code, "redzones", "exec"))
PY
              tail -3 $O/redzones.txt ;;
    concat)   timeout 900 python tools/concat_probe.py > $O/concat_probe.txt 2>&1; tail -28 $O/concat_probe.txt ;;
    rq)       timeout 900 python tools/rq_probe.py > $O/rq_probe.txt 2>&1; tail -20 $O/rq_probe.txt ;;
    dropin)   timeout 900 python tools/probe.py dropin > $O/dropin.txt 2>&1; tail -12 $O/dropin.txt ;;
    dasan)    timeout 1500 bash tools/device_asan.sh > /dev/null 2>&1; cp gpurun_out/device_asan.txt $O/ 2>/dev/null; tail -8 $O/device_asan.txt ;;
  esac
  echo "== $s: $(( $(date +%s) - t0 )) s"
done
