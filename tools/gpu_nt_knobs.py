import os, sys, numpy as np
np.seterr(over='ignore')
sys.path.insert(0, '.')
import swipe_amd
from swipe_amd import synth, blastdb
nseq = 1_000_000
rtab = synth.residue_table_nucleotide()
q = synth._random_residues(99, 1, 1000, rtab); qm = blastdb.revcomp_nt16(q)
res, off = swipe_amd.synth_db(3, nseq, protein=False)
db = swipe_amd.Database.from_arrays(res, off, symtype=0)
db.set_scoring(swipe_amd.matrix_nucleotide(1, -3), 5, 2)
base = None
for k, w in [(32, 3), (32, 2), (24, 3), (16, 4), (32, 3), (32, 2), (24, 3), (16, 4)]:
    db.set_option("mp_k", str(k)); db.set_option("mp_w", str(w))
    s1, s2, c = db.search2(q, qm)
    if base is None: base = (s1, s2)
    ok = np.array_equal(s1, base[0]) and np.array_equal(s2, base[1])
    print("K=%d W=%d: %.0f GCUPS (%.2f ms) same=%s" % (k, w, c['cells'] / c['kernel_ms'] / 1e6, c['kernel_ms'], ok))
