"""K = 40..48 of the 8-lane split kernel: staged build vs cross-step pipelined build (SWA_PIPE=2)."""
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
db.set_option("lanes", 8)
for K in range(45, 49):
    q = full[: 8 * K - 1]
    out, ref = [], None
    for pipe in ("0", "2"):
        db.set_option("pipe", pipe)
        sc, _ = db.search(q)
        ref = sc if ref is None else ref
        same = np.array_equal(sc, ref)
        best = min(db.search(q, want_scores=False)[1]["kernel_ms"] for _ in range(3))
        out.append((len(q) * float(off[-1]) / best / 1e6, same))
    print("K=%2d staged %.0f  cross-step %.0f (identical=%s)" % (K, out[0][0], out[1][0], out[1][1]), flush=True)
