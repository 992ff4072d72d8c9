"""Two equally long protein queries per pass, kernel GCUPS (aggregate over both) by query length: 2,000,000-sequence
database, thresholds of a 10 M-sequence database (80).  usage: python tools/gpu_pair_probe.py 260 300 375 384 ..."""
import sys, numpy as np
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
    q1, q2 = full[:qlen], full[3000:3000 + qlen][::-1].copy()
    db.search_pair_topk(q1, q2, keep=(250, 250), minscore=(80, 80))
    best = None
    for _ in range(3):
        c = db.search_pair_topk(q1, q2, keep=(250, 250), minscore=(80, 80))[2]
        if best is None or c["kernel_ms"] < best["kernel_ms"]: best = c
    print("qlen %4d: form %2d K=%2d kernel %6.0f GCUPS, search %6.0f GCUPS" % (qlen, best["narrow_shifted"], best["narrow_rows"],
          best["cells"] / best["kernel_ms"] / 1e6, best["cells"] / best["total_ms"] / 1e6), flush=True)
