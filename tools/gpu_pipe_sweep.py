"""Per-K comparison of the split kernel's two builds (all profile units staged vs pipelined one unit ahead)."""
import os, sys, numpy as np
np.seterr(over='ignore')
sys.path.insert(0, '.')
import swipe_amd
from swipe_amd import synth
rtab = synth.residue_table_protein()
full = synth._random_residues(7, 1, 800, rtab)
res, off = swipe_amd.synth_db(1, 2_000_000, query=full[:375])
db = swipe_amd.Database.from_arrays(res, off)
db.set_scoring(swipe_amd.matrix_builtin("BLOSUM62"), 11, 1)
for G in (8, 4):
    db.set_option("lanes", str(G))
    for K in range(30, 37):
        q = full[: G * K]
        out = []
        for pipe in ("0", "1"):
            db.set_option("pipe", pipe)
            db.search(q, want_scores=False)
            best = min(db.search(q, want_scores=False)[1]["kernel_ms"] for _ in range(3))
            out.append(len(q) * float(off[-1]) / best / 1e6)
        print("G=%d K=%2d  staged %.0f  pipelined %.0f  %s" % (G, K, out[0], out[1], "PIPE" if out[1] > out[0] * 1.005 else ""), flush=True)
