import os, sys, numpy as np
np.seterr(over='ignore')
sys.path.insert(0, '.')
import swipe_amd
from swipe_amd import synth, blastdb
nseq = 1_000_000
rtab = synth.residue_table_protein()
for qlen in (1000, 3000):
    q = synth._random_residues(7, 1, qlen, rtab)
    res, off = swipe_amd.synth_db(1, nseq, query=q)
    db = swipe_amd.Database.from_arrays(res, off)
    db.set_scoring(swipe_amd.matrix_builtin("BLOSUM62"), 11, 1)
    base = None
    for k in (16, 24, 32, 16, 24, 32):
        db.set_option("mp_k", str(k))
        s1, c = db.search(q)
        if base is None: base = s1
        print("qlen %d pair K=%d: %.0f GCUPS (%.2f ms) rows %d same=%s requeued %d" % (qlen, k, c['cells'] / c['kernel_ms'] / 1e6, c['kernel_ms'], c['narrow_rows'], np.array_equal(s1, base), c['wide']))
    db.close()
