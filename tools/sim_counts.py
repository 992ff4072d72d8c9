"""Wave-instruction counts of the shipped kernels, from tools/gfx950sim (no GPU): per cell pair for the first-pass builds, per
step for the re-queue forms.  The first-pass kernels are VALU-issue bound on MI355X (98.7 % busy, profiles/r03_pmc_bench.csv), so
issued instructions per cell pair predict their rate; for the bench kernel the hardware counter says 6.63 (SQ_INSTS_VALU /
cell pairs, r03) - the count below must agree.

    tools/gfx950sim/run.sh python tools/sim_counts.py > profiles/r05_sim_instruction_counts.txt
"""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import os, sys, json
sys.path.insert(0, %r)
import numpy as np
np.seterr(over="ignore")
import swipe_amd
from swipe_amd import blastdb, synth
q0 = blastdb.encode_protein(synth.QUERY_P07327)
kind, qlen, mode = sys.argv[1], int(sys.argv[2]), sys.argv[3]
if kind == "protein":
    res, off = swipe_amd.synth_db(1, 20000, query=q0)
    db = swipe_amd.Database.from_arrays(res, off)
    db.set_scoring(swipe_amd.matrix_builtin("BLOSUM62"), 11, 1)
    q = q0[:qlen] if qlen <= len(q0) else np.concatenate([q0, q0])[:qlen]
    if mode == "exact":
        c = db.search(q, want_scores=False)[1]
    else:
        db.set_option("bound", "1")
        c = db.search_topk(q, keep=250, minscore=80 if qlen > 100 else 30)[3]
else:
    res, off = swipe_amd.synth_db(3, 20000, protein=False)
    db = swipe_amd.Database.from_arrays(res, off, symtype=0)
    db.set_scoring(swipe_amd.matrix_nucleotide(1, -3), 5, 2)
    q = synth._random_residues(99, 1, qlen, synth.residue_table_nucleotide())
    c = db.search2_topk(q, blastdb.revcomp_nt16(q), keep=250, minscore=25)[3]
print(json.dumps({"cells": c["cells"], "rows": c["narrow_rows"], "form": c["narrow_shifted"]}))
"""

RQ = r"""
import os, sys, json
sys.path.insert(0, %r)
import numpy as np
np.seterr(over="ignore")
import swipe_amd
from swipe_amd import blastdb, synth
q0 = blastdb.encode_protein(synth.QUERY_P07327)
qlen, form = int(sys.argv[1]), sys.argv[2]
rng = np.random.default_rng(5)
res, off = swipe_amd.synth_db(2, 3000, query=q0)
extra = [rng.integers(1, 21, n).astype(np.uint8) for n in (6000, 3500, 2500)]
res = np.concatenate([res] + extra); off = np.concatenate([off, off[-1] + np.cumsum([len(e) for e in extra])])
db = swipe_amd.Database.from_arrays(res, off)
db.set_scoring(swipe_amd.matrix_builtin("BLOSUM62"), 11, 1)
db.set_option("window", 0); db.set_option("bound", "1"); db.set_option("requeue_block", form)
q = np.concatenate([q0, q0, q0])[:qlen]
c = db.search_topk(q, keep=200, minscore=35)[3]
lens = np.diff(off)
print(json.dumps({"requeued": int(c["wide"]), "columns": int(lens.sum()), "nseq": int(len(lens))}))
"""


def run(code, args, extra_env=None):
    with tempfile.TemporaryDirectory() as d:
        stats = os.path.join(d, "s.jsonl")
        env = dict(os.environ, HIPSIM_STATS=stats, **(extra_env or {}))
        r = subprocess.run([sys.executable, "-c", code % ROOT] + [str(a) for a in args], capture_output=True, text=True, env=env)
        if r.returncode:
            raise SystemExit(r.stderr[-2000:])
        meta = json.loads(r.stdout.strip().splitlines()[-1])
        rows = [json.loads(l) for l in open(stats)]
    return meta, rows


def total(d):
    return sum(d[k] for k in ("salu", "valu", "vop3p", "lds", "vmem", "smem", "branch", "other"))


def main():
    if os.environ.get("HIPSIM") != "1":
        raise SystemExit("run under tools/gfx950sim/run.sh")
    print("Wave-instruction counts of the kernels in swipe_amd/libswipe_amd.so, interpreted by tools/gfx950sim (tools/sim_counts.py).")
    print("Per cell pair = per (database residue x query row) / 2 of the 20 000-sequence synthetic shard, padding and skew charged.\n")
    print("%-40s %4s %4s | %6s %6s %6s %6s %6s %6s | %6s %8s %9s | %s" % ("first pass", "rows", "form", "vop3p", "valu", "salu", "lds", "nop/w", "branch", "VALU", "if bound", "r03 meas.", "kernel"))
    # a VALU-issue-bound kernel on MI355X: 256 CUs x 4 SIMDs, one wave64 VALU instruction per 4 cycles per SIMD, 2.4 GHz, 128 cells per
    # wave instruction slot of a cell pair -> GCUPS = 256 * 4 * 2.4e9 / (4 * VALU per cell pair) * 128 / 1e9
    bound = lambda v: 256 * 4 * 2.4e9 / (4.0 * v) * 128 / 1e9
    measured = {("protein", 5, "exact"): 7600, ("protein", 10, "exact"): 8800, ("protein", 20, "exact"): 9700, ("protein", 48, "exact"): 9736,
                ("protein", 49, "exact"): 8950, ("protein", 375, "exact"): 9539, ("protein", 5, "bound"): 8400, ("protein", 10, "bound"): 10500,
                ("protein", 375, "bound"): 11614, ("nucleotide", 1000, "both strands"): 10734}
    only = [a for a in sys.argv[1:] if not a.startswith("-")]
    table = [("protein", 5, "exact", None), ("protein", 10, "exact", None), ("protein", 20, "exact", None), ("protein", 48, "exact", None),
             ("protein", 49, "exact", None), ("protein", 375, "exact", None), ("protein", 5, "bound", None), ("protein", 10, "bound", None),
             ("protein", 60, "bound", None),
             # the bench kernel: the round-3 form (every set of batches on its own, one copy of the profile), then round 6's two steps.
             # SWA_CONCAT_TAIL=16: 1.3 % of the 1 250 sets singly, which is what the default (four sets per resident wave) comes to on
             # the 10 M-sequence database (8 192 of 625 000 sets)
             ("protein", 375, "bound", {"SWA_CONCAT": "1", "SWA_TWIN": "0"}), ("protein", 375, "bound", {"SWA_CONCAT": "16", "SWA_CONCAT_TAIL": "16", "SWA_TWIN": "0"}),
             ("protein", 375, "bound", {"SWA_CONCAT": "16", "SWA_CONCAT_TAIL": "16", "SWA_TWIN": "1"}),
             ("protein", 375, "bound", {"SWA_CONCAT": "8", "SWA_CONCAT_TAIL": "16", "SWA_TWIN": "1"}),
             ("protein", 375, "bound", {"SWA_CONCAT": "32", "SWA_CONCAT_TAIL": "16", "SWA_TWIN": "1"}),
             ("protein", 100, "bound", {"SWA_CONCAT": "1", "SWA_TWIN": "0"}), ("protein", 100, "bound", {"SWA_CONCAT": "16", "SWA_CONCAT_TAIL": "16", "SWA_TWIN": "1"}),
             ("nucleotide", 1000, "both strands", None)]
    for kind, qlen, mode, extra in table:
        label = "%s %d, %s%s" % (kind, qlen, mode, (", " + " ".join("%s=%s" % (k[4:].lower(), v) for k, v in extra.items())) if extra else "")
        if only and not all(o in label for o in only):
            continue
        meta, rows = run(CHILD, [kind, qlen, mode], extra)
        d = max((x for x in rows if x["vop3p"] > 0), key=lambda x: x["vop3p"])
        pairs = meta["cells"] / 2 / 64 / (2 if kind == "nucleotide" else 1) * (2 if kind == "nucleotide" else 1)
        v = (d["vop3p"] + d["valu"]) / pairs
        m = measured.get((kind, qlen, mode))
        print("%-40s %4d %4d | %6.2f %6.2f %6.2f %6.2f %6.2f %6.2f | %6.2f %8.0f %9s | %s" % (
            label, meta["rows"], meta["form"], d["vop3p"] / pairs, d["valu"] / pairs, d["salu"] / pairs,
            d["lds"] / pairs, d["other"] / pairs, d["branch"] / pairs, v, bound(v), ("%d" % m) if m else "-", d["kernel"][:52]))
    print("\n('if bound' = GCUPS of a kernel that does nothing but issue these VALU instructions, 4 cycles each per SIMD, 256 CUs, 2.4 GHz;")
    print(" 'r03 meas.' = GCUPS measured on MI355X in round 3 on the 10 M-sequence database (profiles/r03_sweep_protein_qlen_10M.txt, BENCH_r03).")
    print(" Where the two agree the build is VALU-issue bound and only fewer instructions make it faster; where they do not - 5 and 10 rows -")
    print(" something the count does not see holds it back.)")
    print("(MI355X, r03 PMC pass of the bench command: SQ_INSTS_VALU = 6.63 per cell pair for swa_narrow_bound_kernel<47,2,8,16>; floor 6.0;")
    print(" the exact form's floor is 7.5, the two-query form's 6.5 / 5.)\n")
    if only:
        return
    print("Re-queue behind the first pass: wave-instructions per step of ONE wave, and on the critical path of a 6 000-column entry")
    print("%5s %-6s %5s | %8s %7s %7s %6s | %9s" % ("qlen", "form", "K", "all/step", "valu", "salu", "lds", "path, M"))
    for qlen in (200, 375, 768, 1024):
        for form, name in (("0", "wave"), ("1", "block")):
            meta, rows = run(RQ, [qlen, form])
            d = max((x for x in rows if "requeue" in x["kernel"]), key=total)
            skew = 255 if name == "block" else 63
            steps = (meta["columns"] + skew * meta["nseq"]) * (4 if name == "block" else 1)
            k = d["kernel"].split("kernelILi")[1].split("E")[0]
            print("%5d %-6s %5s | %8.1f %7.1f %7.1f %6.1f | %9.2f" % (qlen, name, k, total(d) / steps, d["valu"] / steps, d["salu"] / steps, d["lds"] / steps,
                                                                     (6000 + skew) * total(d) / steps / 1e6))
    print("\n(block = swa_requeue_block_kernel: four waves per sequence, one barrier per step - not counted here, the hardware's to price.)")


if __name__ == "__main__":
    main()
