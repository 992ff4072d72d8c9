import os, sys, numpy as np
np.seterr(over='ignore')
sys.path.insert(0, '.')
import swipe_amd
from swipe_amd import synth
rtab = synth.residue_table_protein()
full = synth._random_residues(7, 1, 6000, rtab)
res, off = swipe_amd.synth_db(1, 2_000_000)
db = swipe_amd.Database.from_arrays(res, off)
db.set_scoring(swipe_amd.matrix_builtin("BLOSUM62"), 11, 1)
db.set_option("bound", 1)
for qlen in (41, 44, 48, 52, 56, 60, 64, 70, 76, 80, 88, 96):
    q = full[:qlen]; out = []
    for lanes in ("2", "4"):
        db.set_option("lanes", lanes)
        hits, tot, obv, c = db.search_topk(q, keep=250, minscore=80)
        best = min(db.search_topk(q, keep=250, minscore=80)[3]["kernel_ms"] for _ in range(3))
        out.append("G=%s form %d K=%2d %6.0f" % (lanes, c["narrow_shifted"], c["narrow_rows"], c["cells"] / best / 1e6))
    print(qlen, " | ".join(out), flush=True)
