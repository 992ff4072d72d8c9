"""Both-strand nucleotide search throughput by query length (single-pass dual kernel vs multi-pass)."""
import os, sys, numpy as np
np.seterr(over='ignore')
sys.path.insert(0, '.')
import swipe_amd
from swipe_amd import synth, blastdb
rtab = synth.residue_table_nucleotide()
full = synth._random_residues(99, 1, 6000, rtab)
res, off = swipe_amd.synth_db(3, 2_000_000, protein=False)
db = swipe_amd.Database.from_arrays(res, off, symtype=0)
db.set_scoring(swipe_amd.matrix_nucleotide(1, -3), 5, 2)
for qlen in map(int, sys.argv[1:]):
    q = full[:qlen]; qm = blastdb.revcomp_nt16(q)
    out = []
    for mp in ("0", "16", "1"):
        db.set_option("dual_mp", "1" if mp == "1" else "0")
        if mp == "16":
            db.set_option("lanes", "16")
        else:
            db.set_option("lanes", None)
        db.search2(q, qm, want_scores=False)
        best, c = 1e9, None
        for _ in range(3):
            _, _, c = db.search2(q, qm, want_scores=False)
            best = min(best, c["kernel_ms"])
        out.append("%s K=%2d %.0f GCUPS" % ({"0": "default", "16": "16-lane chains", "1": "block-synchronous multi-pass"}[mp], c["narrow_rows"], c["cells"] / best / 1e6))
    print("qlen %4d: %s" % (qlen, " | ".join(out)), flush=True)
