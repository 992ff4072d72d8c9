#!/bin/bash
# PMC passes over an arbitrary probe command (one counter group per pass, --kernel-trace only), per-kernel means printed.
# usage on the GPU box: bash tools/gpu_pmc_probe.sh <tag> <command...>
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
TAG=$1; shift
O=gpurun_out/pmc_$TAG
rm -rf $O; mkdir -p $O
i=0
for grp in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES GRBM_GUI_ACTIVE" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/p$i -- "$@" > $O/out$i.txt 2> $O/err$i.txt
done
python - "$O" <<'PY'
import sys, glob, csv, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(acc.items()):
    if "swa_" not in k or "format" in k: continue
    print(k[:90])
    for c, v in sorted(d.items()): print("   %-24s n=%3d mean=%.4g" % (c, len(v), sum(v) / len(v)))
PY
