"""Randomised parity sweep of the alignment phase: GPU end points + host traceback (Database.align) against the
oracle's search16s + align() (both pinned on the compiled reference) for gappy homologs under random scoring systems."""
import os, sys, numpy as np
np.seterr(over='ignore')
sys.path.insert(0, '.')
import swipe_amd, oracle
from swipe_amd import synth, blastdb
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
mats = ["BLOSUM45", "BLOSUM62", "BLOSUM80", "PAM30", "PAM250"]
bad = 0
for it in range(n):
    protein = rng.random() < 0.7
    qlen = int(rng.integers(5, 1200))
    if protein:
        m = str(rng.choice(mats)); M, Mo = swipe_amd.matrix_builtin(m), oracle.matrix_builtin(m)
        go, ge = int(rng.integers(1, 16)), int(rng.integers(1, 4)); tab = synth.residue_table_protein()
    else:
        a, b = int(rng.integers(1, 5)), -int(rng.integers(1, 5))
        m = "%d/%d" % (a, b); M, Mo = swipe_amd.matrix_nucleotide(a, b), oracle.matrix_nucleotide(a, b)
        go, ge = int(rng.integers(1, 8)), int(rng.integers(1, 4)); tab = synth.residue_table_nucleotide()
    q = synth._random_residues(int(rng.integers(1 << 30)), 1, qlen, tab)
    seqs = []
    for k in range(60):
        piece = q[int(rng.integers(0, max(1, qlen // 2))): int(rng.integers(qlen // 2, qlen)) + 1].copy()
        mut = rng.random(len(piece)) < rng.random() * 0.25
        piece[mut] = tab[rng.integers(0, len(tab), int(mut.sum()))]
        for _ in range(int(rng.integers(0, 4))):            # indels
            if len(piece) > 10:
                at = int(rng.integers(1, len(piece) - 1))
                if rng.random() < 0.5: piece = np.delete(piece, slice(at, at + int(rng.integers(1, 6))))
                else: piece = np.insert(piece, at, tab[rng.integers(0, len(tab), int(rng.integers(1, 6)))])
        flank = lambda: tab[rng.integers(0, len(tab), int(rng.integers(0, 80)))]
        seqs.append(np.concatenate([flank(), piece, flank()]).astype(np.uint8))
    r2, o2 = oracle.pack(seqs)
    db = swipe_amd.Database.from_arrays(r2, o2, symtype=1 if protein else 0)
    db.set_scoring(M, go, ge)
    ids = list(range(len(seqs)))
    ds = [int(rng.random() < 0.3) if not protein else 0 for _ in ids]
    lim16 = oracle.score_limits(Mo)[3]
    got = db.align(q, ids, ds)
    ok = True
    for i, a in zip(ids, got):
        d = blastdb.revcomp_nt16(seqs[i]) if ds[i] else seqs[i]
        sc, bp, bq = oracle.search16s_lane(d, q, Mo, go + ge, ge)
        hint = (sc, bq, bp) if (sc < lim16 and bq > 0 and bp != 0) else None
        if sc <= 0:
            continue
        want = oracle.align(q, d, Mo, go, ge, hint)
        if (a["score"], a["q_start"], a["d_start"], a["q_end"], a["d_end"], a["cigar"]) != want:
            ok = False
    db.close()
    bad += not ok
    print("%3d %s %-8s go=%2d ge=%d qlen=%4d : %s" % (it, "aa" if protein else "nt", m, go, ge, qlen, "ok" if ok else "MISMATCH"), flush=True)
print("align fuzz done:", n, "configs,", bad, "bad")
