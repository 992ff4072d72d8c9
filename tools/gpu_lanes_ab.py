"""A/B: 16 lanes vs 8 lanes per sequence pair in the row-shifted kernel, same database, bench query."""
import os, sys, numpy as np
np.seterr(over='ignore')
sys.path.insert(0, '.')
import swipe_amd
from swipe_amd import synth, blastdb
nseq = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
q = blastdb.encode_protein(synth.QUERY_P07327)
res, off = swipe_amd.synth_db(1, nseq, query=q)
db = swipe_amd.Database.from_arrays(res, off)
db.set_scoring(swipe_amd.matrix_builtin("BLOSUM62"), 11, 1)
ref = None
for lanes in ("16", "8", "16", "8"):
    db.set_option("lanes", lanes)
    s, c = db.search(q)
    if ref is None:
        ref = s
    best = min(db.search(q, want_scores=False)[1]["kernel_ms"] for _ in range(4))
    print("lanes %2s: K=%2d form=%d  %.2f ms  %.0f GCUPS  identical=%s" % (lanes, c["narrow_rows"], c["narrow_shifted"], best, c["cells"] / best / 1e6, np.array_equal(s, ref)))
for qlen in (30, 60, 100, 150, 192, 200, 300, 375):
    qq = q[:qlen]
    for lanes in ("16", "8", "4"):
        db.set_option("lanes", lanes)
        s, c = db.search(qq)
        best = min(db.search(qq, want_scores=False)[1]["kernel_ms"] for _ in range(3))
        print("qlen %3d lanes %2s: K=%2d  %.0f GCUPS" % (qlen, lanes, c["narrow_rows"], c["cells"] / best / 1e6))
