"""Short nucleotide queries on both strands: one lane per sequence vs chains of 2 and 4 lanes."""
import os, sys, numpy as np
np.seterr(over='ignore')
sys.path.insert(0, '.')
import swipe_amd
from swipe_amd import synth, blastdb
rtab = synth.residue_table_nucleotide()
full = synth._random_residues(99, 1, 400, rtab)
res, off = swipe_amd.synth_db(3, 2_000_000, protein=False)
db = swipe_amd.Database.from_arrays(res, off, symtype=0)
db.set_scoring(swipe_amd.matrix_nucleotide(1, -3), 5, 2)
for qlen in map(int, sys.argv[1:]):
    q = full[:qlen]; qm = blastdb.revcomp_nt16(q)
    out = []
    ref = None
    for lanes in ("1", "2", "4"):
        db.set_option("lanes", lanes)
        s1, s2, c = db.search2(q, qm)
        if ref is None: ref = (s1, s2)
        best = min(db.search2(q, qm, want_scores=False)[2]["kernel_ms"] for _ in range(3))
        ok = np.array_equal(s1, ref[0]) and np.array_equal(s2, ref[1])
        out.append("K=%2d %5.0f GCUPS %s" % (c["narrow_rows"], c["cells"] / best / 1e6, "" if ok else "MISMATCH"))
    print("qlen %3d: %s" % (qlen, " | ".join(out)), flush=True)
