"""Top-250 search with the E <= 10 threshold of a 10 M-sequence database (score 80), kernel GCUPS by query length:
whatever first pass the library picks (bound builds where they exist) next to the exact first pass (SWA_BOUND=0)."""
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
for qlen in map(int, sys.argv[1:]):
    q = full[:qlen]
    out = []
    for mode in (None, "0"):
        if mode: db.set_option("bound", mode)
        else: db.set_option("bound", None)
        hits, tot, obv, c = db.search_topk(q, keep=250, minscore=80)
        best = min(db.search_topk(q, keep=250, minscore=80)[3]["kernel_ms"] for _ in range(3))
        tot_ms = min(db.search_topk(q, keep=250, minscore=80)[3]["total_ms"] for _ in range(3))
        out.append("form %2d K=%2d kernel %6.0f GCUPS, search %6.0f GCUPS, requeued %4d" % (c["narrow_shifted"], c["narrow_rows"], c["cells"] / best / 1e6, c["cells"] / tot_ms / 1e6, c["wide"]))
    print("qlen %4d: %s | exact: %s" % (qlen, out[0], out[1]), flush=True)
