"""Randomised parity sweep (needs a GPU): random scoring systems, query lengths, alphabets, one- and two-query
searches, top-K searches with random thresholds (automatic first pass and the bound build forced), inclusion
subsets - every score against the oracle's 63-bit recurrence.

    python tools/gpu_fuzz.py [configs] [seed]

tests/test_gpu_parity.py::test_seeded_fuzz_slice runs a bounded slice of the same generator under `-m gpu`."""
import os
import sys

import numpy as np

np.seterr(over='ignore')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle
import swipe_amd
from swipe_amd import blastdb, synth

MATS = ["BLOSUM45", "BLOSUM50", "BLOSUM62", "BLOSUM80", "BLOSUM90", "PAM30", "PAM70", "PAM250"]


def one_config(rng, threads, max_qlen=2500, max_nseq=3000):
    """One random configuration; returns (description, ok_scores, ok_two_queries, ok_topk, ok_subset)."""
    protein = rng.random() < 0.7
    qlen = int(rng.choice([rng.integers(1, 64), rng.integers(64, 800), rng.integers(800, max_qlen)]))
    nseq = int(rng.integers(200, max_nseq))
    if protein:
        m = str(rng.choice(MATS)); M, Mo = swipe_amd.matrix_builtin(m), oracle.matrix_builtin(m)
        go, ge = int(rng.integers(0, 20)), int(rng.integers(1, 5))
        tab = synth.residue_table_protein()
    else:
        a, b = int(rng.integers(1, 6)), -int(rng.integers(1, 6))
        m = "%d/%d" % (a, b); M, Mo = swipe_amd.matrix_nucleotide(a, b), oracle.matrix_nucleotide(a, b)
        go, ge = int(rng.integers(0, 12)), int(rng.integers(1, 7))
        tab = synth.residue_table_nucleotide()
    q = synth._random_residues(int(rng.integers(1 << 30)), 1, qlen, tab)
    res, off = swipe_amd.synth_db(int(rng.integers(1 << 20)), nseq, query=q if protein else None, protein=protein)
    seqs = [res[off[i]:off[i + 1]] for i in range(nseq)]
    for k in range(5):                                      # homologs of all strengths, incl. the query itself
        cut = int(rng.integers(0, max(1, qlen // 2)))
        piece = q[cut:].copy()
        mut = rng.random(len(piece)) < rng.random() * 0.3
        piece[mut] = tab[rng.integers(0, len(tab), int(mut.sum()))]
        seqs.append(np.concatenate([seqs[k][:20], piece, seqs[k + 1][:int(rng.integers(0, 30))]]))
    if rng.random() < 0.3:                                  # a long sequence carrying the query: searched as windows
        body = tab[rng.integers(0, len(tab), int(rng.integers(6000, 30000)))].astype(np.uint8)
        at = int(rng.integers(0, len(body) - qlen))
        body[at:at + qlen] = q
        seqs.append(body)
    r2, o2 = oracle.pack(seqs)
    db = swipe_amd.Database.from_arrays(r2, o2, symtype=1 if protein else 0)
    db.set_scoring(M, go, ge)
    want = oracle.search_all63(r2, o2, q, Mo, go + ge, ge, threads=threads)
    got, c = db.search(q)
    ok = bool(np.array_equal(got, want))
    q2 = (blastdb.revcomp_nt16(q) if not protein else q[::-1].copy())
    want2 = oracle.search_all63(r2, o2, q2, Mo, go + ge, ge, threads=threads)
    g1, g2, c2 = db.search2(q, q2)
    ok2 = bool(np.array_equal(g1, want) and np.array_equal(g2, want2))
    # top-K searches: random threshold and cap, automatic choice of the first pass and the bound build forced
    ok4, forms = True, []
    for force in (None, 1):
        db.set_option("bound", force)
        lo = int(rng.choice([1, rng.integers(1, 60), rng.integers(60, 200), max(1, int(want.max()) - int(rng.integers(0, 40)))]))
        hi = int(rng.choice([1 << 62, lo + int(rng.integers(0, 300))]))
        keep = int(rng.integers(1, 300))
        hits, tot, obv, ck = db.search_topk(q, keep=keep, minscore=lo, maxscore=hi)
        order = sorted((i for i in range(len(want)) if want[i] >= lo), key=lambda i: (-int(want[i]), -i))
        exp = [(i, int(want[i])) for i in order if want[i] <= hi][:keep]
        ok4 = ok4 and hits == exp and tot == len(order) and obv == int((want > hi).sum())
        forms.append(ck["narrow_shifted"])
        # both queries at once with the same window
        h2, t2, o2b, ck2 = db.search2_topk(q, q2, keep=keep, minscore=lo, maxscore=hi)
        e2 = sorted([(int(s), i, 0) for i, s in enumerate(want) if lo <= s <= hi] +
                    [(int(s), i, 1) for i, s in enumerate(want2) if lo <= s <= hi], key=lambda t: (-t[0], -t[1], t[2]))[:keep]
        ok4 = ok4 and h2 == [(i, s, w) for s, i, w in e2] and t2 == int((want >= lo).sum() + (want2 >= lo).sum())
    db.set_option("bound", None)
    # two DIFFERENT queries in one pass, each with its own window
    if protein:
        q3 = synth._random_residues(int(rng.integers(1 << 30)), 1, max(1, int(qlen * rng.uniform(0.75, 1.0))), tab)
        want3 = oracle.search_all63(r2, o2, q3, Mo, go + ge, ge, threads=threads)
        lo1, lo3 = int(rng.integers(1, 80)), int(rng.integers(1, 80))
        (h1, t1, o1), (h3, t3, o3), _ = db.search_pair_topk(q, q3, keep=(25, 40), minscore=(lo1, lo3))
        def exp(w, keep, lo):
            order = sorted((i for i in range(len(w)) if w[i] >= lo), key=lambda i: (-int(w[i]), -i))
            return [(i, int(w[i])) for i in order[:keep]], len(order)
        ok4 = ok4 and (h1, t1) == exp(want, 25, lo1) and (h3, t3) == exp(want3, 40, lo3)
    inc = (rng.random(len(seqs)) < 0.6).astype(np.uint8)
    db.set_inclusion(inc)
    g3, _ = db.search(q)
    ok3 = bool(np.array_equal(g3[inc == 1], want[inc == 1]) and np.all(g3[inc == 0] == -1))
    db.close()
    desc = "%s %-9s go=%2d ge=%d qlen=%4d nseq=%4d max=%6d narrow_rows=%2d shifted=%d wide=%d full=%d topk forms %s" % (
        "aa" if protein else "nt", m, go, ge, qlen, len(seqs), int(want.max()), c["narrow_rows"], c["narrow_shifted"], c["wide"],
        c["full"], forms)
    return desc, ok, ok2, bool(ok4), ok3


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    T = os.cpu_count() or 1
    bad = 0
    for it in range(n):
        desc, ok, ok2, ok4, ok3 = one_config(rng, T)
        bad += not (ok and ok2 and ok3 and ok4)
        print("%3d %s : %s %s %s %s" % (it, desc, "ok" if ok else "MISMATCH", "ok" if ok2 else "MISMATCH2", "ok" if ok4 else "MISMATCH-TOPK",
                                        "ok" if ok3 else "MISMATCH3"), flush=True)
    print("fuzz done:", n, "configs,", bad, "bad")
