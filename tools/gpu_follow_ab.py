"""A/B of the re-queue follower (second stream, beside the first pass): blocks of the follower vs first-pass kernel time
and whole-step wall time, bench query on a shard of the bench database."""
import os, sys, time, numpy as np
np.seterr(over='ignore')
sys.path.insert(0, '.')
import swipe_amd
from swipe_amd import synth, blastdb
nseq = int(sys.argv[1]) if len(sys.argv) > 1 else 1_250_000
q = blastdb.encode_protein(synth.QUERY_P07327)
res, off = swipe_amd.synth_db(1, nseq, query=q)
db = swipe_amd.Database.from_arrays(res, off, total_seqcount=10_000_000, total_symcount=3_237_270_683)
db.set_scoring(swipe_amd.matrix_builtin("BLOSUM62"), 11, 1)
st = swipe_amd.stats_init(qlen=len(q), db_seqcount=10_000_000, db_symcount=3_237_270_683)
ref = None
for rnd in range(2):
    for follow in (0, 1, 32, 128, 256, 1024, 2048):
        db.set_option("requeue_follow", follow)
        db.search_topk_array(q, keep=250, minscore=st.scorethreshold)
        t = time.perf_counter(); k = []
        for _ in range(10):
            hits, tot, obv, c = db.search_topk_array(q, keep=250, minscore=st.scorethreshold)
            k.append(c["kernel_ms"])
        wall = (time.perf_counter() - t) / 10 * 1e3
        ref = hits if ref is None else ref
        print("follow %5d: kernel %.3f ms  step %.3f ms  overhead %.3f  same hits %s  requeued %d" % (
            follow, np.mean(k), wall, wall - np.mean(k), np.array_equal(hits, ref), c["wide"]), flush=True)
