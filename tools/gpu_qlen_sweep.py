"""Protein query-length sweep: which kernel wins where (tuned single pass vs multi-pass K=16)."""
import os, sys, numpy as np
np.seterr(over='ignore')
sys.path.insert(0, '.')
import swipe_amd
from swipe_amd import synth
nseq = 1_000_000
rtab = synth.residue_table_protein()
for qlen in (50, 64, 100, 128, 200, 256, 300, 375, 450, 512, 600, 700, 768):
    q = synth._random_residues(7, 1, qlen, rtab)
    res, off = swipe_amd.synth_db(1, nseq, query=q)
    db = swipe_amd.Database.from_arrays(res, off)
    db.set_scoring(swipe_amd.matrix_builtin("BLOSUM62"), 11, 1)
    out = []
    for force in ("0",):
        db.set_option("force_mp", force)
        best = 1e9
        for _ in range(2):
            _, c = db.search(q, want_scores=False)
            best = min(best, c['kernel_ms'])
        out.append("%s %.0f GCUPS (K=%d)" % ("mp" if force == "1" else "tuned", c['cells'] / best / 1e6, c['narrow_rows']))
    print("qlen", qlen, " | ".join(out))
    db.close()
