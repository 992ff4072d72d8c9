import sys, time, numpy as np
np.seterr(over='ignore')
sys.path.insert(0, '.')
import swipe_amd, oracle
from swipe_amd import synth, blastdb
q = blastdb.encode_protein(synth.QUERY_P07327)
M = swipe_amd.matrix_builtin("blosum62")
nseq = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
res, off = swipe_amd.synth_db(1, nseq, query=q)
# append forced special cases: empty, 1-residue, the query itself, triple query (overflow), long
extra = [np.zeros(0, np.uint8), q[:1], q, np.concatenate([q, q, q]), np.concatenate([q] * 8)[:2900]]
seqs = [res[off[i]:off[i + 1]] for i in range(nseq)] + extra
db = swipe_amd.Database.from_sequences(seqs)
print(db.info())
db.set_scoring(M, 11, 1)
t = time.time(); scores, c = db.search(q); print("search wall", time.time() - t, c)
r2, o2 = oracle.pack(seqs)
t = time.time(); ref = oracle.search_all63(r2, o2, q, M, 12, 1, threads=64); print("oracle", time.time() - t)
bad = np.nonzero(scores != ref)[0]
print("mismatches", len(bad), bad[:10], scores[bad[:10]], ref[bad[:10]])
print("special", scores[-5:], ref[-5:])
hits, tot, obv, c2 = db.search_topk(q, keep=10, minscore=40)
print(hits, tot, obv)
cells = c['cells']
for _ in range(3):
    _, c = db.search(q, want_scores=False)
    print("GCUPS kernel %.1f total %.1f" % (cells / c['kernel_ms'] / 1e6, cells / c['total_ms'] / 1e6), c)
