"""Short protein queries: one lane per sequence pair vs chains of 2, 4 and 8 lanes (exact kernel, all scores)."""
import os, sys, numpy as np
np.seterr(over='ignore')
sys.path.insert(0, '.')
import swipe_amd
from swipe_amd import synth
rtab = synth.residue_table_protein()
full = synth._random_residues(7, 1, 400, rtab)
res, off = swipe_amd.synth_db(1, 2_000_000)
db = swipe_amd.Database.from_arrays(res, off)
db.set_scoring(swipe_amd.matrix_builtin("BLOSUM62"), 11, 1)
for qlen in map(int, sys.argv[1:]):
    q = full[:qlen]
    out = []
    ref = None
    for lanes in ("1", "2", "4", "8"):
        db.set_option("lanes", lanes)
        s, c = db.search(q)
        if ref is None: ref = s
        best = min(db.search(q, want_scores=False)[1]["kernel_ms"] for _ in range(3))
        out.append("form %d K=%2d %5.0f GCUPS %s" % (c["narrow_shifted"], c["narrow_rows"], c["cells"] / best / 1e6, "" if np.array_equal(s, ref) else "MISMATCH"))
    print("qlen %3d: %s" % (qlen, " | ".join(out)), flush=True)
