"""Two different 375-aa queries per pass (swa_search_pair_topk) vs one query per pass, bench database and thresholds:
aggregate GCUPS of a multi-query file."""
import sys, time, numpy as np
np.seterr(over='ignore')
sys.path.insert(0, '.')
import swipe_amd
from swipe_amd import synth, blastdb
nseq = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
q1 = blastdb.encode_protein(synth.QUERY_P07327)
rtab = synth.residue_table_protein()
q2 = synth._random_residues(4242, 1, 375, rtab)
q3 = synth._random_residues(4243, 1, 330, rtab)
res, off = swipe_amd.synth_db(1, nseq, query=q1)
db = swipe_amd.Database.from_arrays(res, off)
db.set_scoring(swipe_amd.matrix_builtin("BLOSUM62"), 11, 1)
nsym = int(off[-1])
def thr(q):
    st = swipe_amd.stats_init(qlen=len(q), db_seqcount=nseq, db_symcount=nsym)
    return st.scorethreshold, st.upperscorethreshold
for a, b in ((q1, q2), (q1, q3)):
    (la, ha), (lb, hb) = thr(a), thr(b)
    singles = [db.search_topk(a, keep=250, minscore=la, maxscore=ha), db.search_topk(b, keep=250, minscore=lb, maxscore=hb)]
    t = time.perf_counter()
    for _ in range(3):
        db.search_topk(a, keep=250, minscore=la, maxscore=ha)
        db.search_topk(b, keep=250, minscore=lb, maxscore=hb)
    t_single = (time.perf_counter() - t) / 3
    r = db.search_pair_topk(a, b, keep=250, minscore=(la, lb), maxscore=(ha, hb))
    t = time.perf_counter()
    for _ in range(3):
        r = db.search_pair_topk(a, b, keep=250, minscore=(la, lb), maxscore=(ha, hb))
    t_pair = (time.perf_counter() - t) / 3
    same = r[0][0] == singles[0][0] and r[1][0] == singles[1][0] and r[0][1] == singles[0][1] and r[1][1] == singles[1][1]
    cells = nsym * (len(a) + len(b))
    print("queries %d + %d aa: one per pass %.1f ms = %.0f GCUPS; paired %.1f ms = %.0f GCUPS (kernel %.1f ms, form %d, K %d); same hits %s" % (
        len(a), len(b), t_single * 1e3, cells / t_single / 1e9, t_pair * 1e3, cells / t_pair / 1e9, r[2]["kernel_ms"],
        r[2]["narrow_shifted"], r[2]["narrow_rows"], same), flush=True)
