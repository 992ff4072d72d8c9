import sys, time, numpy as np
np.seterr(over='ignore')
sys.path.insert(0, '.')
import swipe_amd
from swipe_amd import synth, blastdb
q = blastdb.encode_protein(synth.QUERY_P07327)
res, off = swipe_amd.synth_db(1, 10_000_000, query=q)
db = swipe_amd.Database.from_arrays(res, off)
db.set_scoring(swipe_amd.matrix_builtin("BLOSUM62"), 11, 1)
db.search(q, want_scores=False)
for w in (False, True, True):
    t = time.time(); s, c = db.search(q, want_scores=w); dt = time.time() - t
    print("want_scores=%s: %.1f ms wall (kernel %.1f ms)" % (w, dt * 1e3, c["kernel_ms"]))
