"""Condense gpurun_out/profile/ (tools/profile_round.sh) into the small files kept under profiles/:
<round>_kernel_stats_bench10M.csv, <round>_pmc_bench10M.csv, hbm_traffic.json, <round>_bench_line.json."""
import csv, glob, json, os, sys
src, rnd = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(src, "summary")
os.makedirs(dst, exist_ok=True)
line = open(os.path.join(src, "bench_line.json")).read().strip().splitlines()[-1]
bench = json.loads(line)
open(os.path.join(dst, f"{rnd}_bench_line.json"), "w").write(line + "\n")
def newest(pattern):
    """gpurun merges every call's files into the same directories: keep only the latest run of each"""
    files = glob.glob(pattern, recursive=True)
    return [max(files, key=os.path.getmtime)] if files else []
stats = newest(os.path.join(src, "stats", "**", "*kernel_stats.csv"))
kernel, kms = None, None
if stats:
    rows = list(csv.DictReader(open(stats[0])))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    with open(os.path.join(dst, f"{rnd}_kernel_stats_bench10M.csv"), "w") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()), quoting=csv.QUOTE_NONNUMERIC)
        w.writeheader()
        for r in rows[:12]:
            w.writerow(r)
    kernel, kms = rows[0]["Name"], float(rows[0]["AverageNs"]) / 1e6
vals = {}
for f in [x for d in sorted(glob.glob(os.path.join(src, "pmc[0-9]*/"))) for x in newest(os.path.join(d, "**", "*counter_collection.csv"))]:
    for r in csv.DictReader(open(f)):
        if kernel and r["Kernel_Name"] != kernel:
            continue
        vals.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
mean = {k: sum(v) / len(v) for k, v in vals.items()}
nseq = bench["config"]["sequences_per_gpu"]
nsym = bench["roofline"]["algorithmic_bytes_per_launch"] - 12 * nseq
with open(os.path.join(dst, f"{rnd}_pmc_bench10M.csv"), "w") as f:
    f.write("# rocprofv3 --pmc passes (one counter group per pass, --kernel-trace only), %s\n" % rnd)
    f.write("# command: rocprofv3 --kernel-trace --pmc <group> --output-format csv -- python bench.py --steps 2 --warmup 0 --no-cpu-baseline\n")
    f.write("# kernel: %s, %d sequences / %d residues, 375-aa query; values = mean per launch\n" % (kernel, nseq, nsym))
    f.write("counter,value_per_launch\n")
    for k in sorted(mean):
        f.write("%s,%.1f\n" % (k, mean[k]))
    if "FETCH_SIZE" in mean and kms:
        rd, wr = mean["FETCH_SIZE"] * 1024 * 2, mean.get("WRITE_SIZE", 0) * 1024
        alg = nsym + 12 * nseq
        f.write("#\n# derived (kernel duration %.2f ms from %s_kernel_stats_bench10M.csv):\n" % (kms, rnd))
        f.write("# HBM read bytes  = FETCH_SIZE KB x 1024 x 2 (gfx950 correction) = %.3e" % rd)
        if "TCC_EA0_RDREQ_sum" in mean:
            f.write("  (TCC_EA0_RDREQ_sum x 128 B = %.3e)" % (mean["TCC_EA0_RDREQ_sum"] * 128))
        f.write("\n# HBM write bytes = WRITE_SIZE KB x 1024 = %.3e\n" % wr)
        f.write("# algorithmic bytes per launch = %d + 12 x %d = %.3e  -> traffic/algorithmic = %.2f\n" % (nsym, nseq, alg, (rd + wr) / alg))
        if "GRBM_GUI_ACTIVE" in mean:
            ghz = mean["GRBM_GUI_ACTIVE"] / 8 / (kms * 1e-3) / 1e9
            f.write("# shader clock during the kernel = GRBM_GUI_ACTIVE / 8 XCD / %.2f ms = %.2f GHz\n" % (kms, ghz))
            if "SQ_INSTS_VALU" in mean:
                cap = 1024 * ghz * 1e9 * kms * 1e-3 / 4
                f.write("# VALU issue capacity = 1024 SIMD x cycles / 4 = %.3e wave-instructions; SQ_INSTS_VALU = %.3e -> %.1f %% of VALU issue slots\n"
                        % (cap, mean["SQ_INSTS_VALU"], 100 * mean["SQ_INSTS_VALU"] / cap))
                f.write("# cells per VALU wave-instruction = %.2f (128 / %.2f instructions per cell pair incl. padding, skew and per-step overhead)\n"
                        % (nsym * 375 / mean["SQ_INSTS_VALU"], 128 / (nsym * 375 / mean["SQ_INSTS_VALU"])))
        if mean.get("SQ_LDS_BANK_CONFLICT", 1) == 0:
            f.write("# SQ_LDS_BANK_CONFLICT = 0: one 16-byte profile unit per lane position of a DPP row keeps the lane chains of a row on disjoint bank groups\n")
        json.dump({"nseq": nseq, "bytes_per_launch": int(rd + wr), "fetch_size_kb_raw": mean["FETCH_SIZE"],
                   "write_size_kb_raw": mean.get("WRITE_SIZE", 0),
                   "correction": "FETCH_SIZE x 2 on gfx950 (MI355X_MICROARCH.md HBM section); WRITE_SIZE as reported",
                   "source": f"profiles/{rnd}_pmc_bench10M.csv"}, open(os.path.join(dst, "hbm_traffic.json"), "w"), indent=1)
print("kernel", kernel, "avg ms", kms, "counters", sorted(mean))
