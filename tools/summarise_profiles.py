"""gpurun_out/prof_r02 (tools/gpu_profile.sh) -> profiles/r02_kernel_stats_bench.csv, profiles/r02_pmc_bench.csv,
profiles/hbm_traffic.json.  Run in the build container after the GPU call merged its output back."""
import csv, glob, json, os, re, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", "prof_r02")
out = os.path.join(ROOT, "profiles")

def short(name):
    return re.sub(r"\(.*", "", name).replace("void ", "").strip()

# --- kernel stats
stats = glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True)
rows = []
if stats:
    for r in csv.DictReader(open(stats[0])):
        rows.append(r)
    with open(os.path.join(out, "r02_kernel_stats_bench.csv"), "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-verify --no-cold\n")
        f.write("# (protein headline 1 + 3 steps, exact first pass 1 + 3 steps, nucleotide secondary 1 + 3 steps, pair section: two 375-aa queries\n")
        f.write("# per pass = swa_dual_bound_kernel 1 + 3 steps, then each of the two alone through swa_narrow_bound_kernel; durations in ns)\n")
        f.write("# Calls = 1 untimed warm-up + 3 timed steps: bench.py reports the mean of the timed ones (AverageNs within 0.3 % of it).\n")
        f.write("# swa_requeue_follow_kernel runs BESIDE the first-pass kernel on a second stream (DESIGN.md 4.10): its duration is its\n")
        f.write("# lifetime = the producer's, not work; swa_requeue_wave_kernel is the finishing kernel after it.\n")
        cols = list(rows[0].keys())
        f.write(",".join(cols) + "\n")
        for r in rows[:14]:
            f.write(",".join('"%s"' % r[c] if c == "Name" else r[c] for c in cols) + "\n")
    print("kernel stats:", len(rows), "kernels")
    for r in rows[:6]:
        print("  %-70s calls %4s avg %12s ns" % (short(r["Name"])[:70], r["Calls"], r.get("AverageNs", r.get("Average", "?"))))

# --- PMC: mean per launch per kernel
pmc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(src, "pmc*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        pmc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
want = ("swa_narrow_bound_kernel", "swa_narrow_split_kernel", "swa_dual_kernel")
lines = []
traffic = {}
for k in sorted(pmc):
    if not k.startswith(want):
        continue
    c = {n: sum(v) / len(v) for n, v in pmc[k].items()}
    n_launch = max(len(v) for v in pmc[k].values())
    lines.append((k, n_launch, c))
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        traffic[k] = {"fetch_size_kb_raw": c["FETCH_SIZE"], "write_size_kb_raw": c["WRITE_SIZE"],
                      "bytes_per_launch": int(c["FETCH_SIZE"] * 1024 * 2 + c["WRITE_SIZE"] * 1024)}
with open(os.path.join(out, "r02_pmc_bench.csv"), "w") as f:
    f.write("# rocprofv3 --kernel-trace --pmc <group> (one group per pass) -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-verify --no-cold\n")
    f.write("# mean per launch; 10 M sequences (protein: 3 237 270 683 residues, 375-aa query; nucleotide: 3 237 408 910 bases, 1 kb query, both strands)\n")
    f.write("kernel,launches,counter,value_per_launch\n")
    for k, n, c in lines:
        for name in sorted(c):
            f.write('"%s",%d,%s,%.1f\n' % (k, n, name, c[name]))
    f.write("# HBM bytes = FETCH_SIZE KB x 1024 x 2 (gfx950 correction, MI355X_MICROARCH.md) + WRITE_SIZE KB x 1024\n")
    for k, t in traffic.items():
        f.write("# %s: %.3e bytes per launch\n" % (k, t["bytes_per_launch"]))
rec = {}
for k, t in traffic.items():
    key = "protein" if k.startswith("swa_narrow_bound") else "exact" if k.startswith("swa_narrow_split") else "nucleotide"
    rec[key] = dict(t, nseq=10_000_000, kernel=k, correction="FETCH_SIZE x 2 on gfx950 (MI355X_MICROARCH.md HBM section); WRITE_SIZE as reported",
                    source="profiles/r02_pmc_bench.csv (rocprofv3 --pmc passes of this command, committed; not re-measured in this run)")
if rec:
    json.dump(rec, open(os.path.join(out, "hbm_traffic.json"), "w"), indent=1)
print(json.dumps(rec, indent=1))
for k, n, c in lines:
    print(k, n, {a: "%.4g" % b for a, b in c.items()})
