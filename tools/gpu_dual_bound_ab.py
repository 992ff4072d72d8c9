"""Two protein queries of equal length (frames of a translated search), top-K with a threshold: exact two-query kernel
vs its bound build."""
import os, sys, numpy as np
np.seterr(over='ignore')
sys.path.insert(0, '.')
import swipe_amd
from swipe_amd import synth
rtab = synth.residue_table_protein()
full = synth._random_residues(7, 1, 600, rtab)
res, off = swipe_amd.synth_db(1, 2_000_000)
db = swipe_amd.Database.from_arrays(res, off)
db.set_scoring(swipe_amd.matrix_builtin("BLOSUM62"), 11, 1)
for qlen in map(int, sys.argv[1:]):
    q1 = full[:qlen]; q2 = full[::-1][:qlen].copy()
    ref = None
    for mode in ("0", "1"):
        db.set_option("bound", mode)
        hits, tot, obv, c = db.search2_topk(q1, q2, keep=250, minscore=80)
        if ref is None: ref = (hits, tot, obv)
        best = min(db.search2_topk(q1, q2, keep=250, minscore=80)[3]["kernel_ms"] for _ in range(3))
        print("qlen %3d bound=%s form %2d K=%2d %6.0f GCUPS requeued %d same=%s" % (qlen, mode, c["narrow_shifted"], c["narrow_rows"], c["cells"] / best / 1e6, c["wide"], (hits, tot, obv) == ref), flush=True)
