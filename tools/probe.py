#!/usr/bin/env python3
"""One probe tool for the GPU box (round 3: replaces the one-off tools/gpu_*.py A/B scripts).

    python tools/probe.py qlen  [--nseq N] [--topk T] [--nt] [--opt key=value ...] [--ab key=v1,v2] QLEN [QLEN ...]
        kernel throughput (best of --reps launches) by query length with the default kernel choice; --topk T = top-250
        search with score threshold T (bound builds) instead of all scores; --nt = nucleotide database, both strands;
        --opt sets handle options for every run; --ab runs every length once per value of one option, side by side
    python tools/probe.py rates [--nseq N]
        kernel GCUPS of EVERY (G, K) build of the one-query first pass, exact and bound, at qlen = G x K: the measured
        table tools/gen_kernel_table.py turns into swipe_amd/csrc/kernel_rates.inc (the kernel selection is its argmax)
    python tools/probe.py rates2 [--nseq N]
        the same for the two-query kernels (nucleotide both strands; protein pairs exact and bound)
    python tools/probe.py table [--max 1100]
        the kernel-selection table as the library resolves it (no GPU needed): qlen -> (form, G, K) for exact / top-K
    python tools/probe.py longest [--nseq N] QLEN ...
        where the time goes for short queries: kernel time with windows auto / off / forced small

Prints plain text; run on the box through gpurun and redirect into gpurun_out/."""
import argparse
import os
import sys

import numpy as np

np.seterr(over="ignore")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def make_db(nseq, nt=False, seed=1):
    import swipe_amd
    from swipe_amd import synth
    rtab = synth.residue_table_nucleotide() if nt else synth.residue_table_protein()
    full = synth._random_residues(7, 1, 6000, rtab)
    res, off = swipe_amd.synth_db(3 if nt else seed, nseq, query=None if nt else full[:375], protein=not nt)
    db = swipe_amd.Database.from_arrays(res, off, symtype=0 if nt else 1)
    if nt:
        db.set_scoring(swipe_amd.matrix_nucleotide(1, -3), 5, 2)
    else:
        db.set_scoring(swipe_amd.matrix_builtin("BLOSUM62"), 11, 1)
    return db, full, int(off[-1])


def run_one(db, q, nt, topk, reps):
    from swipe_amd import blastdb
    best, c = 1e9, None
    for _ in range(reps + 1):
        if nt:
            qm = blastdb.revcomp_nt16(q)
            c = db.search2_topk(q, qm, keep=250, minscore=topk)[3] if topk else db.search2(q, qm, want_scores=False)[2]
        else:
            c = db.search_topk(q, 250, topk)[3] if topk else db.search(q, want_scores=False)[1]
        best = min(best, c["kernel_ms"])
    return best, c


def cmd_qlen(a):
    db, full, nsym = make_db(a.nseq, a.nt)
    for kv in a.opt or []:
        k, v = kv.split("=")
        db.set_option(k, v)
    ab_key, ab_vals = None, [None]
    if a.ab:
        ab_key, vals = a.ab.split("=")
        ab_vals = vals.split(",")
    print(f"# {a.nseq} sequences ({nsym} residues), {'nt both strands' if a.nt else 'protein'}, "
          f"{'top-250 with threshold %d' % a.topk if a.topk else 'all scores exact'}, best of {a.reps}; options {a.opt or []}")
    for qlen in a.qlens:
        q = full[:qlen]
        cols = []
        for v in ab_vals:
            if ab_key:
                db.set_option(ab_key, v)
            ms, c = run_one(db, q, a.nt, a.topk, a.reps)
            cols.append("%s%7.0f GCUPS %8.3f ms K=%2d form=%2d" % ((ab_key + "=" + v + ": ") if ab_key else "", c["cells"] / ms / 1e6, ms,
                                                                    c["narrow_rows"], c["narrow_shifted"]))
        print("qlen %5d  " % qlen + "  |  ".join(cols), flush=True)
    db.close()


def cmd_longest(a):
    import swipe_amd
    db, full, nsym = make_db(a.nseq)
    info = db.info()
    print(f"# {a.nseq} sequences, longest {info['longest']}")
    for qlen in a.qlens:
        q = full[:qlen]
        out = []
        for name, opts in (("auto", {}), ("off", {"window": 0}), ("win>1000 step 512", {"window": 1000, "window_step": 512}),
                           ("win>600 step 384", {"window": 600, "window_step": 384})):
            for k in ("window", "window_step"):
                db.set_option(k, opts.get(k))
            ms, c = run_one(db, q, False, 0, a.reps)
            out.append("%s %.3f ms %.0f GCUPS" % (name, ms, c["cells"] / ms / 1e6))
        print("qlen %4d  " % qlen + " | ".join(out), flush=True)
    db.close()


def cmd_rates(a):
    """every build of the one-query first pass at qlen = G x K (no padding rows): the input of tools/gen_kernel_table.py"""
    db, full, nsym = make_db(a.nseq)
    print(f"# kernel GCUPS of every (G, K) build at qlen = G x K, {a.nseq} sequences ({nsym} residues), BLOSUM62 11/1, best of {a.reps}")
    print("# mode G K qlen gcups form")
    for mode, topk in (("exact", 0), ("bound", 80)):
        db.set_option("bound", 1 if topk else 0)
        for G in (1, 2, 4, 8, 16):
            db.set_option("lanes", G)
            kmax = (60 if topk else 48) if G == 1 else (58 if G == 16 else (62 if topk else 48))
            for K in range(1, kmax + 1):
                qlen = G * K
                if qlen > len(full):
                    break
                ms, c = run_one(db, full[:qlen], False, topk, a.reps)
                form = c["narrow_shifted"]
                want = 8 if topk else {1: 11, 2: 7, 4: 3, 8: 2, 16: 1}[G]
                if form != want or c["narrow_rows"] != K:
                    continue                                   # no such build (or the scoring system rules it out)
                print("%s %2d %2d %4d %6.0f %2d" % (mode, G, K, qlen, c["cells"] / ms / 1e6, form), flush=True)
    db.close()


def cmd_rates2(a):
    """every build of the TWO-query first pass (both strands of a nucleotide query; two protein queries / frames) at
    qlen = G x K: the second input of tools/gen_kernel_table.py"""
    print(f"# two-query kernels: GCUPS (both queries counted) of every (G, K) build at qlen = G x K, {a.nseq} sequences, best of {a.reps}")
    print("# mode G K qlen gcups form")
    for nt in (True, False):
        db, full, nsym = make_db(a.nseq, nt)
        kone = 48 if nt else 32
        for mode, topk in ((("dual16" if nt else "dual32"), 0),) + ((("dualbound32", 80),) if not nt else ()):
            db.set_option("bound", 1 if topk else 0)
            for G in (1, 2, 4, 8, 16):
                db.set_option("lanes", G)
                kmax = kone if G == 1 else 32 if (G == 2 or not nt) and not topk else 63 if G == 16 else 62 if topk else 60
                for K in range(1, kmax + 1):
                    qlen = G * K
                    q = full[:qlen]
                    q2 = blast_rc(q) if nt else np.ascontiguousarray(q[::-1])
                    best, c = 1e9, None
                    for _ in range(a.reps + 1):
                        c = db.search2_topk(q, q2, keep=250, minscore=topk)[3] if topk else db.search2(q, q2, want_scores=False)[2]
                        best = min(best, c["kernel_ms"])
                    want = 10 if topk else (12 if G == 1 else 4)
                    if c["narrow_shifted"] != want or c["narrow_rows"] != K:
                        continue
                    print("%s %2d %2d %4d %6.0f %2d" % (mode, G, K, qlen, c["cells"] / best / 1e6, c["narrow_shifted"]), flush=True)
        db.close()


def blast_rc(q):
    from swipe_amd import blastdb
    return blastdb.revcomp_nt16(q)


def cmd_table(a):
    import ctypes as C
    from swipe_amd import _lib
    L = _lib.load()
    for bound, label in ((0, "exact (all scores)"), (1, "top-K (bound build wanted)")):
        print("#", label, "- BLOSUM62 11/1")
        last = None
        for qlen in range(1, a.max + 1):
            g, k, b, p = (C.c_int32() for _ in range(4))
            L.swa_kernel_choice(qlen, bound, 11, 12, 1, 35000, 0.0, 0, C.byref(g), C.byref(k), C.byref(b), C.byref(p))
            cur = (g.value, b.value)
            if cur != last or a.all:
                print("qlen %5d..  G %2d  K %2d  %s  predicted %5d GCUPS" % (qlen, g.value, k.value, "bound" if b.value else "exact", p.value))
                last = cur


def main():
    ap = argparse.ArgumentParser()
    sub = ap.add_subparsers(dest="cmd", required=True)
    p = sub.add_parser("qlen")
    p.add_argument("qlens", type=int, nargs="+")
    p.add_argument("--nseq", type=int, default=2_000_000)
    p.add_argument("--topk", type=int, default=0)
    p.add_argument("--nt", action="store_true")
    p.add_argument("--reps", type=int, default=4)
    p.add_argument("--opt", action="append")
    p.add_argument("--ab")
    p.set_defaults(fn=cmd_qlen)
    p = sub.add_parser("longest")
    p.add_argument("qlens", type=int, nargs="+")
    p.add_argument("--nseq", type=int, default=2_000_000)
    p.add_argument("--reps", type=int, default=4)
    p.set_defaults(fn=cmd_longest)
    p = sub.add_parser("rates")
    p.add_argument("--nseq", type=int, default=4_000_000)
    p.add_argument("--reps", type=int, default=2)
    p.set_defaults(fn=cmd_rates)
    p = sub.add_parser("rates2")
    p.add_argument("--nseq", type=int, default=4_000_000)
    p.add_argument("--reps", type=int, default=2)
    p.set_defaults(fn=cmd_rates2)
    p = sub.add_parser("table")
    p.add_argument("--max", type=int, default=1100)
    p.add_argument("--all", action="store_true")
    p.set_defaults(fn=cmd_table)
    a = ap.parse_args()
    a.fn(a)


if __name__ == "__main__":
    main()
