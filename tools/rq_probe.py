"""The re-queue's tail on hardware: per-search overhead = wall time of a top-K step - the first-pass kernel's
own time (HIP events), for the builds the verdict names - the 47-row bound build (375 aa), the 52-row build (416 aa) and the
63-row two-query nucleotide build (1 kb, both strands) - on the 10 M-sequence database and on a 1.25 M-sequence shard (what one
of 8 GPUs holds), with the re-queue worked off a wave per sequence (requeue_block=0) and a block of four waves per sequence
(requeue_block=1).  Hit lists of the two forms must be identical.  Run by tools/round6_gpu.sh rq.

    python tools/rq_probe.py [--quick]        (--quick: 1 M / 125 k sequences, for a first look)
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
        kms.append(last[3]["kernel_ms"])
    return float(np.median(walls)) * 1e3, float(np.median(kms)), last


def main():
    quick = "--quick" in sys.argv
    sizes = (1_000_000, 125_000) if quick else (10_000_000, 1_250_000)
    q375 = blastdb.encode_protein(synth.QUERY_P07327)
    q416 = np.concatenate([q375, q375[:41]])
    rows = []
    for nseq in sizes:
        res, off = swipe_amd.synth_db(1, nseq, query=q375, threads=os.cpu_count() or 1)
        db = swipe_amd.Database.from_arrays(res, off)
        db.set_scoring(swipe_amd.matrix_builtin("BLOSUM62"), 11, 1)
        for name, q in (("375 aa", q375), ("416 aa", q416)):
            st = swipe_amd.stats_init(qlen=len(q), db_seqcount=nseq, db_symcount=int(off[-1]))
            got = {}
            for form in ("0", "1"):
                db.set_option("requeue_block", form)
                wall, k, last = timed(lambda: db.search_topk(q, keep=250, minscore=st.scorethreshold, maxscore=st.upperscorethreshold), 9)
                got[form] = last[:3]
                rows.append((nseq, name, "block" if form == "1" else "wave", last[3]["narrow_rows"], last[3]["narrow_shifted"], last[3]["wide"], k, wall, wall - k))
            assert got["0"] == got["1"], ("hit lists differ", nseq, name)
        db.close()
        del res, off
        nres, noff = swipe_amd.synth_db(3, nseq * 5 // 2 if not quick else nseq, protein=False, threads=os.cpu_count() or 1)
        ndb = swipe_amd.Database.from_arrays(nres, noff, symtype=0)
        ndb.set_scoring(swipe_amd.matrix_nucleotide(1, -3), 5, 2)
        qn = synth._random_residues(99, 1, 1000, synth.residue_table_nucleotide())
        qm = blastdb.revcomp_nt16(qn)
        st = swipe_amd.stats_init(symtype=0, match=1, mismatch=-3, gapopen=5, gapextend=2, qlen=1000, db_seqcount=len(noff) - 1, db_symcount=int(noff[-1]))
        got = {}
        for form in ("0", "1"):
            ndb.set_option("requeue_block", form)
            wall, k, last = timed(lambda: ndb.search2_topk(qn, qm, keep=250, minscore=st.scorethreshold), 5)
            got[form] = last[:3]
            rows.append((len(noff) - 1, "1 kb nt x2", "block" if form == "1" else "wave", last[3]["narrow_rows"], last[3]["narrow_shifted"], last[3]["wide"], k, wall, wall - k))
        assert got["0"] == got["1"], ("hit lists differ", nseq, "nt")
        ndb.close()
        del nres, noff
    print("%10s  %-10s %-5s %4s %4s %9s %10s %10s %11s" % ("sequences", "query", "rq", "rows", "form", "requeued", "kernel ms", "step ms", "overhead ms"))
    for r in rows:
        print("%10d  %-10s %-5s %4d %4d %9d %10.3f %10.3f %11.3f" % r)


if __name__ == "__main__":
    main()
