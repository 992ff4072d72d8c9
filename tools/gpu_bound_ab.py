"""A/B: exact first pass vs the bound build (sw_cb_kernels.hip) for a top-K search; identical hit lists required."""
import os, sys, numpy as np
np.seterr(over='ignore')
sys.path.insert(0, '.')
import swipe_amd
from swipe_amd import synth, blastdb
nseq = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
minscore = int(sys.argv[2]) if len(sys.argv) > 2 else 80
q = blastdb.encode_protein(synth.QUERY_P07327)
res, off = swipe_amd.synth_db(1, nseq, query=q)
db = swipe_amd.Database.from_arrays(res, off)
db.set_scoring(swipe_amd.matrix_builtin("BLOSUM62"), 11, 1)
for qlen in [int(x) for x in sys.argv[3:]] or [375]:
    qq = q[:qlen] if qlen <= len(q) else np.concatenate([q] * (qlen // len(q) + 1))[:qlen]
    ref = None
    for mode in ("0", "1", "0", "1"):
        db.set_option("bound", mode)
        hits, tot, obv, c = db.search_topk(qq, keep=250, minscore=minscore)
        best = min(db.search_topk(qq, keep=250, minscore=minscore)[3]["kernel_ms"] for _ in range(3))
        tbest = min(db.search_topk(qq, keep=250, minscore=minscore)[3]["total_ms"] for _ in range(3))
        if ref is None: ref = (hits, tot, obv)
        print("qlen %4d bound=%s: form %d K=%2d kernel %.2f ms (%.0f GCUPS) search %.2f ms requeued %d totalhits %d identical=%s" % (
            qlen, mode, c["narrow_shifted"], c["narrow_rows"], best, c["cells"] / best / 1e6, tbest, c["wide"], tot, (hits, tot, obv) == ref), flush=True)
