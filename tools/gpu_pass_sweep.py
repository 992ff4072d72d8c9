"""Pass builds of the row-shifted kernel (long queries): GCUPS per rows-per-lane K for each (PIPE, DEFER) build.
Query length 32 K (two passes of exactly K rows per lane)."""
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
os.environ["SWA_PASS_KMAX"] = "56"
for K in map(int, sys.argv[1:]):
    q = full[:32 * K]
    out = []
    for pipe in (0, 1):
        for defer in (0, 1):
            os.environ["SWA_PASS_PIPE"] = str(pipe)
            os.environ["SWA_PASS_DEFER"] = str(defer)
            db.search(q, want_scores=False)
            best, c = 1e9, None
            for _ in range(3):
                _, c = db.search(q, want_scores=False)
                best = min(best, c["kernel_ms"])
            assert c["narrow_rows"] == K and c["narrow_shifted"] == 5
            out.append("p%dd%d %5.0f" % (pipe, defer, c["cells"] / best / 1e6))
    print("K=%2d  %s" % (K, "  ".join(out)), flush=True)
