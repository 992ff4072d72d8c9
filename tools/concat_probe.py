"""Round 6's changes to the bound build on hardware: first-pass kernel time and step time of the bench search (375-aa
query, top-250 at E <= 10) on the 10 M-sequence database and on a 1.25 M-sequence shard (what one of 8 GPUs holds), for
concat = 1 / 4 / 8 / 16 / 32 sets per item x twin = 0 / 1, and of two 375-aa queries per pass for concat = 1 / 16.  Hit lists
must be identical throughout.  The defaults (concat = 16, twin = 1) were chosen on instruction counts alone; this table is what
should set them.  Run by tools/round6_gpu.sh concat.

    python tools/concat_probe.py [--quick]        (--quick: 1 M / 125 k sequences)
"""
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import numpy as np

np.seterr(over="ignore")
import swipe_amd
from swipe_amd import blastdb, synth


def timed(fn, reps):
    fn()
    walls, kms, last = [], [], None
    for _ in range(reps):
        t0 = time.perf_counter()
        last = fn()
        walls.append(time.perf_counter() - t0)
        kms.append(last[-1]["kernel_ms"])
    return float(np.median(walls)) * 1e3, float(np.median(kms)), last


def main():
    quick = "--quick" in sys.argv
    sizes = (1_000_000, 125_000) if quick else (10_000_000, 1_250_000)
    if os.environ.get("HIPSIM") == "1":
        sizes = (6_000,)                                  # (the interpreter: does the script run)
    q = blastdb.encode_protein(synth.QUERY_P07327)
    q2 = synth._random_residues(4242, 1, len(q), synth.residue_table_protein())
    rows = []
    for nseq in sizes:
        res, off = swipe_amd.synth_db(1, nseq, query=q, threads=os.cpu_count() or 1)
        db = swipe_amd.Database.from_arrays(res, off)
        db.set_scoring(swipe_amd.matrix_builtin("BLOSUM62"), 11, 1)
        db.set_option("bound", 1)                             # (the 10 M database takes the bound build by itself; the small shard must too)
        st = swipe_amd.stats_init(qlen=len(q), db_seqcount=nseq, db_symcount=int(off[-1]))
        cells = int(off[-1]) * len(q)
        want = None
        for twin in (0, 1):
            for concat in (1, 4, 8, 16, 32):
                db.set_option("twin", twin)
                db.set_option("concat", concat)
                wall, k, last = timed(lambda: db.search_topk(q, keep=250, minscore=st.scorethreshold, maxscore=st.upperscorethreshold), 9)
                if want is None:
                    want = last[:3]
                assert last[:3] == want, ("hit lists differ", nseq, twin, concat)
                rows.append((nseq, "375 aa", concat, twin, last[3]["narrow_rows"], last[3]["narrow_shifted"], last[3]["wide"], k, wall, cells / k / 1e6))
        db.set_option("twin", None)
        wantp = None
        for concat in (1, 16):
            db.set_option("concat", concat)
            wall, k, last = timed(lambda: db.search_pair_topk(q, q2, keep=250, minscore=(st.scorethreshold, st.scorethreshold)), 5)
            if wantp is None:
                wantp = last[:-1]
            assert last[:-1] == wantp, ("pair hit lists differ", nseq, concat)
            c = last[-1]
            rows.append((nseq, "2 x 375 aa", concat, -1, c["narrow_rows"], c["narrow_shifted"], c["wide"], k, wall, 2 * cells / k / 1e6))
        db.close()
        del res, off
    print("%10s  %-10s %6s %4s %4s %4s %9s %10s %10s %12s" % ("sequences", "query", "concat", "twin", "rows", "form", "requeued", "kernel ms", "step ms", "kernel GCUPS"))
    for r in rows:
        print("%10d  %-10s %6d %4d %4d %4d %9d %10.3f %10.3f %12.0f" % r)


if __name__ == "__main__":
    main()
