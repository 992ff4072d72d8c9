"""Throughput of selected query lengths (argv: qlen ...) with the default kernel choice."""
import os, sys, numpy as np
np.seterr(over='ignore')
sys.path.insert(0, '.')
import swipe_amd
from swipe_amd import synth
rtab = synth.residue_table_protein()
full = synth._random_residues(7, 1, 6000, rtab)
res, off = swipe_amd.synth_db(1, 2_000_000, query=full[:375])
db = swipe_amd.Database.from_arrays(res, off)
db.set_scoring(swipe_amd.matrix_builtin("BLOSUM62"), 11, 1)
for qlen in map(int, sys.argv[1:]):
    q = full[:qlen]
    db.search(q, want_scores=False)
    best, c = 1e9, None
    for _ in range(4):
        _, c = db.search(q, want_scores=False)
        best = min(best, c["kernel_ms"])
    print("qlen %4d K=%2d form=%d  %.0f GCUPS" % (qlen, c["narrow_rows"], c["narrow_shifted"], c["cells"] / best / 1e6), flush=True)
