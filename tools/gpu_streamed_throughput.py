"""Database larger than its HBM budget: throughput of the streamed shard (two device slots, PCIe double buffering) vs the
resident one, bench query and thresholds."""
import sys, time, numpy as np
np.seterr(over='ignore')
sys.path.insert(0, '.')
import swipe_amd
from swipe_amd import synth, blastdb
nseq = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
q = blastdb.encode_protein(synth.QUERY_P07327)
res, off = swipe_amd.synth_db(1, nseq, query=q)
nsym = int(off[-1])
st = swipe_amd.stats_init(qlen=len(q), db_seqcount=nseq, db_symcount=nsym)
M = swipe_amd.matrix_builtin("BLOSUM62")
out = {}
db = swipe_amd.Database.from_arrays(res, off)
full = db.info()["hbm_bytes"]
for label, budget in (("resident", 0), ("half", full // 2), ("quarter", full // 4), ("tenth", full // 10)):
    if budget:
        t = time.time()
        db = swipe_amd.Database.from_arrays(res, off, hbm_budget=budget)
        t_open = time.time() - t
    else:
        t_open = 0.0
    db.set_scoring(M, 11, 1)
    for mode in ("topk", "topk-exact"):
        db.set_option("bound", None if mode == "topk" else 0)
        hits = db.search_topk(q, keep=250, minscore=st.scorethreshold, maxscore=st.upperscorethreshold)
        t = time.perf_counter()
        for _ in range(4):
            hits = db.search_topk(q, keep=250, minscore=st.scorethreshold, maxscore=st.upperscorethreshold)
        dt = (time.perf_counter() - t) / 4
        out.setdefault(mode, hits[:3])
        print("%-9s %-10s budget %6.2f GB (device %5.2f GB, open %.1f s): %7.1f ms per search = %6.0f GCUPS, kernels %.1f ms, same hits %s" % (
            label, mode, budget / 1e9, db.info()["hbm_bytes"] / 1e9, t_open, dt * 1e3, nsym * len(q) / dt / 1e9, hits[3]["kernel_ms"],
            hits[:3] == out[mode]), flush=True)
    db.close()
