"""At scale: top-250 hit lists of the bound build vs the exact first pass on the 10 M-sequence database, for queries of
many lengths (random ones and database sequences, which have real hits), E <= 10 thresholds from the statistics."""
import os, sys, time, numpy as np
np.seterr(over='ignore')
sys.path.insert(0, '.')
import swipe_amd
from swipe_amd import synth, blastdb
nseq = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
q0 = blastdb.encode_protein(synth.QUERY_P07327)
res, off = swipe_amd.synth_db(1, nseq, query=q0)
db = swipe_amd.Database.from_arrays(res, off)
db.set_scoring(swipe_amd.matrix_builtin("BLOSUM62"), 11, 1)
rtab = synth.residue_table_protein()
rng = np.random.default_rng(17)
lens = np.diff(off)
queries = [q0]
for L in (30, 45, 60, 90, 120, 180, 250, 330, 400, 520, 700, 900, 1100, 1700, 2600):
    queries.append(synth._random_residues(1000 + L, 1, L, rtab))
    cand = np.nonzero((lens > 0.9 * L) & (lens < 1.1 * L))[0]
    i = int(cand[rng.integers(0, len(cand))])
    queries.append(res[off[i]:off[i + 1]].copy())
bad = 0
for q in queries:
    st = swipe_amd.stats_init(qlen=len(q), db_seqcount=nseq, db_symcount=int(off[-1]))
    out = {}
    for mode in ("0", None):
        if mode: db.set_option("bound", mode)
        else: db.set_option("bound", None)
        t = time.time()
        hits, tot, obv, c = db.search_topk(q, keep=250, minscore=st.scorethreshold, maxscore=st.upperscorethreshold)
        out[mode] = (hits, tot, obv, c, time.time() - t)
    same = out["0"][:3] == out[None][:3]
    bad += not same
    c = out[None][3]
    print("qlen %4d threshold %3d: form %d K=%2d requeued %5d totalhits %5d  %.1f ms vs exact %.1f ms  %s" % (
        len(q), st.scorethreshold, c["narrow_shifted"], c["narrow_rows"], c["wide"], out[None][1], c["total_ms"], out["0"][3]["total_ms"],
        "same" if same else "DIFFERENT"), flush=True)
print("scale check done:", len(queries), "queries,", bad, "different")
