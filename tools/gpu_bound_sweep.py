"""Bound build vs exact first pass, kernel GCUPS for every rows-per-lane K of each chain length G (top-250 search, threshold 80)."""
import os, sys, numpy as np
np.seterr(over='ignore')
sys.path.insert(0, '.')
import swipe_amd
from swipe_amd import synth
rtab = synth.residue_table_protein()
full = synth._random_residues(7, 1, 1000, rtab)
res, off = swipe_amd.synth_db(1, 2_000_000)
db = swipe_amd.Database.from_arrays(res, off)
db.set_scoring(swipe_amd.matrix_builtin("BLOSUM62"), 11, 1)
want = [int(x) for x in sys.argv[1:]]            # optional: G K0 K1
for G in ((want[0],) if want else (2, 4, 8, 16)):
    db.set_option("lanes", str(G))
    for K in range(want[1] if want else 25, (want[2] if want else (58 if G == 16 else 48)) + 1):
        q = full[:G * K]
        out, ref = [], None
        for mode in ("0", "1"):
            db.set_option("bound", mode)
            hits, tot, obv, c = db.search_topk(q, keep=250, minscore=80)
            if ref is None: ref = (hits, tot, obv)
            best = min(db.search_topk(q, keep=250, minscore=80)[3]["kernel_ms"] for _ in range(3))
            out.append((c["cells"] / best / 1e6, c["narrow_shifted"], c["wide"], (hits, tot, obv) == ref))
        print("G=%2d K=%2d exact %5.0f  bound %5.0f (form %d, requeued %d, same hits %s)  %+.1f %%" % (
            G, K, out[0][0], out[1][0], out[1][1], out[1][2], out[1][3], 100 * (out[1][0] / out[0][0] - 1)), flush=True)
