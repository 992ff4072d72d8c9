"""Kernel knob sweep in one process (interleaved rounds): SWA_WAVES x SWA_BLOCKS_PER_CU."""
import os, sys, numpy as np
np.seterr(over='ignore')
sys.path.insert(0, '.')
import swipe_amd
from swipe_amd import synth, blastdb
q = blastdb.encode_protein(synth.QUERY_P07327)
nseq = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
res, off = swipe_amd.synth_db(1, nseq, query=q)
M = swipe_amd.matrix_builtin("blosum62")
db = swipe_amd.Database.from_arrays(res, off)
db.set_scoring(M, 11, 1)
base, _ = db.search(q)
configs = [(w, b) for w in (4, 3, 2) for b in (8, 4, 3, 2)]
best = {}
for rnd in range(3):
    for w, b in configs:
        os.environ["SWA_WAVES"] = str(w); os.environ["SWA_BLOCKS_PER_CU"] = str(b)
        s, c = db.search(q)
        assert np.array_equal(s, base)
        best.setdefault((w, b), []).append(c["kernel_ms"])
for k, v in best.items():
    print("waves %d blocks/CU %d: %.2f ms (min) %.0f GCUPS" % (k[0], k[1], min(v), c["cells"] / min(v) / 1e6))
