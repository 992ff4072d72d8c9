"""Randomised parity sweep (run on the GPU box): random scoring systems, query lengths, alphabets, one- and
two-query searches, inclusion subsets - every score against the oracle's 63-bit recurrence."""
import os, sys, numpy as np
np.seterr(over='ignore')
sys.path.insert(0, '.')
import swipe_amd, oracle
from swipe_amd import synth, blastdb
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
T = os.cpu_count() or 1
bad = 0
mats = ["BLOSUM45", "BLOSUM50", "BLOSUM62", "BLOSUM80", "BLOSUM90", "PAM30", "PAM70", "PAM250"]
for it in range(n):
    protein = rng.random() < 0.7
    qlen = int(rng.choice([rng.integers(1, 64), rng.integers(64, 800), rng.integers(800, 2500)]))
    nseq = int(rng.integers(200, 3000))
    if protein:
        m = str(rng.choice(mats)); M, Mo = swipe_amd.matrix_builtin(m), oracle.matrix_builtin(m)
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
    r2, o2 = oracle.pack(seqs)
    db = swipe_amd.Database.from_arrays(r2, o2, symtype=1 if protein else 0)
    db.set_scoring(M, go, ge)
    want = oracle.search_all63(r2, o2, q, Mo, go + ge, ge, threads=T)
    got, c = db.search(q)
    ok = np.array_equal(got, want)
    q2 = (blastdb.revcomp_nt16(q) if not protein else q[::-1].copy())
    want2 = oracle.search_all63(r2, o2, q2, Mo, go + ge, ge, threads=T)
    g1, g2, c2 = db.search2(q, q2)
    ok2 = np.array_equal(g1, want) and np.array_equal(g2, want2)
    # top-K searches: random threshold and cap, automatic choice of the first pass and the bound build forced
    ok4, forms = True, []
    for force in (None, "1"):
        if force: os.environ["SWA_BOUND"] = force
        else: os.environ.pop("SWA_BOUND", None)
        lo = int(rng.choice([1, rng.integers(1, 60), rng.integers(60, 200), max(1, int(want.max()) - int(rng.integers(0, 40)))]))
        hi = int(rng.choice([1 << 62, lo + int(rng.integers(0, 300))]))
        keep = int(rng.integers(1, 300))
        hits, tot, obv, ck = db.search_topk(q, keep=keep, minscore=lo, maxscore=hi)
        order = sorted((i for i in range(len(want)) if want[i] >= lo), key=lambda i: (-int(want[i]), -i))
        exp = [(i, int(want[i])) for i in order if want[i] <= hi][:keep]
        ok4 = ok4 and hits == exp and tot == len(order) and obv == int((want > hi).sum())
        forms.append(ck["narrow_shifted"])
    os.environ.pop("SWA_BOUND", None)
    ok = ok and ok4
    inc = (rng.random(len(seqs)) < 0.6).astype(np.uint8)
    db.set_inclusion(inc)
    g3, _ = db.search(q)
    ok3 = np.array_equal(g3[inc == 1], want[inc == 1]) and np.all(g3[inc == 0] == -1)
    db.close()
    if not (ok and ok2 and ok3):
        bad += 1
    print("%3d %s %-9s go=%2d ge=%d qlen=%4d nseq=%4d max=%6d narrow_rows=%2d shifted=%d wide=%d full=%d : %s %s %s" % (
        it, "aa" if protein else "nt", m, go, ge, qlen, len(seqs), int(want.max()), c["narrow_rows"], c["narrow_shifted"], c["wide"], c["full"],
        "ok" if ok else "MISMATCH", "ok" if ok2 else "MISMATCH2", "ok" if ok3 else "MISMATCH3"), "topk forms", forms, flush=True)
print("fuzz done:", n, "configs,", bad, "bad")
