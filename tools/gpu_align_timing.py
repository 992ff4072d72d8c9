"""How long does the alignment phase take?  250 hits of the 375-aa bench query, and a long-query / long-sequence
worst case, through swa_align_hits (GPU end points + host traceback)."""
import sys, time, numpy as np
np.seterr(over='ignore')
sys.path.insert(0, '.')
import swipe_amd
from swipe_amd import synth, blastdb
q = blastdb.encode_protein(synth.QUERY_P07327)
nseq = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
res, off = swipe_amd.synth_db(1, nseq, query=q)
db = swipe_amd.Database.from_arrays(res, off)
db.set_scoring(swipe_amd.matrix_builtin("BLOSUM62"), 11, 1)
hits, tot, obv, c = db.search_topk(q, keep=250, minscore=35)
ids = [h[0] for h in hits]
lens = np.diff(off)[ids]
for rep in range(2):
    t = time.time(); e = db.search_endpoints(q, ids); t1 = time.time() - t
    t = time.time(); al = db.align(q, ids); t2 = time.time() - t
    print("250 hits (mean len %.0f, max %d): end points %.1f ms, whole alignment phase %.1f ms" % (lens.mean(), lens.max(), t1 * 1e3, t2 * 1e3))
# worst case: the 100 longest sequences, and a 3000-aa query
order = np.argsort(np.diff(off))[::-1][:100]
rtab = synth.residue_table_protein()
ql = synth._random_residues(5, 1, 3000, rtab)
for name, qq in (("375-aa", q), ("3000-aa", ql)):
    t = time.time(); e = db.search_endpoints(qq, order); t1 = time.time() - t
    print("%s query vs the 100 longest sequences (%d..%d aa): end points %.1f ms" % (name, np.diff(off)[order].min(), np.diff(off)[order].max(), t1 * 1e3))
