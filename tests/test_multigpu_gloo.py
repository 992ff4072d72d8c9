"""CPU, world_size 2 over gloo: the N>1 path - shard by residues, per-rank top-K, ONE all_gather,
merge with the reference comparator - equals the single-list result.  Per-rank scores come from
the oracle here (no GPU in this container); on the GPU box the same code path runs over RCCL."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

import cases
import oracle
from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, keep, q_out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    import swipe_amd
    from swipe_amd import parallel
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        q = cases.Q375
        res, off = swipe_amd.synth_db(1, 3000, query=q)
        seqs = [res[off[i]:off[i + 1]] for i in range(3000)] + cases.case_p1k().seqs[1000:]
        r2, o2 = oracle.pack(seqs)
        lo, hi = parallel.shard_bounds(o2, world)[rank]
        M = oracle.matrix_builtin("BLOSUM62")
        local = oracle.search_all63(r2[o2[lo]:o2[hi]], o2[lo:hi + 1] - o2[lo], q, M, 12, 1)
        minscore = 40
        mine = sorted(((int(s), lo + i) for i, s in enumerate(local) if s >= minscore), key=lambda t: (-t[0], -t[1]))[:keep]
        hits, tot, obv = parallel.gather_topk([(i, s) for s, i in mine], keep, totalhits=int((local >= minscore).sum()))
        # the array form bench.py uses (one all_gather_into_tensor, merged straight from the gathered buffer)
        arr = np.array([(i, s) for s, i in mine], dtype=np.int64).reshape(-1, 2)
        h2, t2, o2b = parallel.gather_topk_array(arr, keep, totalhits=int((local >= minscore).sum()))
        assert [tuple(x) for x in h2.tolist()] == hits and (t2, o2b) == (tot, obv)

        class HostShard:
            """stands in for Database.align on a box without a GPU: end points from the oracle's search16s,
            then the PRODUCT's host traceback (swa_traceback), as swa_align_hits combines them"""
            def align(self, query, seqnos, dstrands=None, dframes=None):
                Mp = swipe_amd.matrix_builtin("BLOSUM62")
                out = []
                for s in seqnos:
                    assert lo <= s < hi
                    sc, bp, bq = oracle.search16s_lane(seqs[s], query, M, 12, 1)
                    a = swipe_amd.traceback(query, seqs[s], Mp, 11, 1, (sc, bq, bp) if (bq > 0 and bp != 0) else None)
                    a["seqno"] = s
                    out.append(a)
                return out

        al = parallel.align_sharded(HostShard(), q, hits[:20], lo, hi)
        if rank == 0:
            q_out.put((hits, tot, (lo, hi), [(a["seqno"], a["score"], a["q_start"], a["d_start"], a["q_end"], a["d_end"], a["cigar"]) for a in al]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("keep", [5, 250])
def test_sharded_topk_equals_single_list(keep):
    world = 2
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, keep, out)) for r in range(world)]
    for p in procs:
        p.start()
    hits, tot, bounds, aligned = out.get(timeout=900)
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    import swipe_amd
    q = cases.Q375
    res, off = swipe_amd.synth_db(1, 3000, query=q)
    seqs = [res[off[i]:off[i + 1]] for i in range(3000)] + cases.case_p1k().seqs[1000:]
    r2, o2 = oracle.pack(seqs)
    full = oracle.search_all63(r2, o2, q, oracle.matrix_builtin("BLOSUM62"), 12, 1, threads=2)
    h = oracle.HitList(descriptions=keep, alignments=0, minscore=40, expect=1e30, dbseqs=len(seqs), dbsyms=int(o2[-1]), qlen=375)
    for i, s in enumerate(full):
        h.enter(i, int(s))
    assert hits == [(x[0], x[1]) for x in h.hits()]
    assert tot == int((full >= 40).sum())
    assert 0 < bounds[1] < len(seqs)
    # alignment phase across shards: each hit aligned by the rank that holds it, gathered in hit order
    M = oracle.matrix_builtin("BLOSUM62")
    assert [a[0] for a in aligned] == [x[0] for x in hits[:20]]
    for seqno, score, qs, ds, qe, de, cigar in aligned:
        sc, bp, bq = oracle.search16s_lane(seqs[seqno], q, M, 12, 1)
        assert oracle.align(q, seqs[seqno], M, 11, 1, (sc, bq, bp) if (bq > 0 and bp != 0) else None) == (score, qs, ds, qe, de, cigar)
