"""Nucleotide query on both strands (swa_search2), kernel GCUPS by query length, 2,000,000-sequence database."""
import sys, numpy as np
np.seterr(over='ignore')
sys.path.insert(0, '.')
import swipe_amd
from swipe_amd import synth, blastdb
tab = synth.residue_table_nucleotide()
full = synth._random_residues(7, 1, 6000, tab)
res, off = swipe_amd.synth_db(3, 2_000_000, protein=False)
db = swipe_amd.Database.from_arrays(res, off, symtype=0)
db.set_scoring(swipe_amd.matrix_nucleotide(1, -3), 5, 2)
for qlen in map(int, sys.argv[1:]):
    q = full[:qlen]; qm = blastdb.revcomp_nt16(q)
    db.search2(q, qm)
    best = min((db.search2(q, qm)[2] for _ in range(3)), key=lambda c: c["kernel_ms"])
    print("qlen %4d: form %2d K=%2d kernel %6.0f GCUPS, search %6.0f GCUPS" % (qlen, best["narrow_shifted"], best["narrow_rows"],
          best["cells"] / best["kernel_ms"] / 1e6, best["cells"] / best["total_ms"] / 1e6), flush=True)
