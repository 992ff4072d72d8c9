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


# ------------------------------------------------------------------------------------------------ round 3: 4 and 8 ranks
def _scenarios():
    """(name, lengths, scores, keep, minscore, maxscore) - global arrays every rank rebuilds identically; the shards are cut
    by residues, so the lengths decide who owns what"""
    rng = np.random.default_rng(2026)
    out = []
    n = 4000
    lens = rng.integers(20, 400, n)
    out.append(("ties_across_boundaries", lens, rng.integers(50, 56, n), 100, 51, 1 << 62))          # 6 distinct scores, keep < hits
    out.append(("keep_1", lens, rng.integers(50, 56, n), 1, 40, 1 << 62))
    big = lens.copy()
    big[1234] = 3_000_000                                                                              # one sequence = most residues:
    out.append(("empty_shards", big, rng.integers(30, 90, n), 250, 60, 1 << 62))                       # several shards own nothing
    z = rng.integers(30, 90, n)
    z[n // 4: 3 * n // 4] = 0
    out.append(("shards_without_hits", lens, z, 250, 40, 1 << 62))
    e = np.zeros(n, dtype=np.int64)
    e[-40:] = rng.integers(100, 110, 40)
    out.append(("all_hits_in_the_last_shard", lens, e, 30, 1, 1 << 62))
    out.append(("fewer_sequences_than_ranks", np.array([10, 0, 25]), np.array([7, 7, 7]), 250, 1, 1 << 62))
    out.append(("nothing_at_all", np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.int64), 250, 1, 1 << 62))
    w = rng.integers(50, 60, n)
    out.append(("score_window", lens, w, 50, 52, 56))                                                  # obvious hits above the window
    return out


def _adversarial_worker(rank, world, port, q_out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from swipe_amd import parallel
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        results = {}
        for name, lens, scores, keep, lo_s, hi_s in _scenarios():
            off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
            lo, hi = parallel.shard_bounds(off, world)[rank]
            local = np.asarray(scores[lo:hi], dtype=np.int64)
            idx = [i for i in range(len(local)) if lo_s <= local[i] <= hi_s]
            idx.sort(key=lambda i: (-int(local[i]), -(lo + i)))
            mine = np.array([(lo + i, int(local[i])) for i in idx[:keep]], dtype=np.int64).reshape(-1, 2)
            tot, obv = int((local >= lo_s).sum()), int((local > hi_s).sum())
            h1, t1, o1 = parallel.gather_topk_array(mine, keep, tot, obv)
            h2, t2, o2 = parallel.gather_topk([tuple(x) for x in mine.tolist()], keep, tot, obv)
            assert [tuple(x) for x in h1.tolist()] == h2 and (t1, o1) == (t2, o2), name
            # frame-tagged form: two entries per sequence (frames 0 and 4) with the same score - ties on (score, seqno)
            fr = [(int(s), int(v), 0, f % 3, f // 3, f % 3) for s, v in mine.tolist() for f in (0, 4)][:keep]
            h6, t6, o6 = parallel.gather_topk(fr, keep, 2 * tot, 2 * obv, width=6)
            results[name] = (h2, t1, o1, h6, (lo, hi))
        if rank == 0:
            q_out.put(results)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [4, 8])
def test_gather_and_merge_at_four_and_eight_ranks(world):
    """the exchange of the 8-GPU run, de-risked on CPU: 4 and 8 gloo ranks, lists built to break a merge - equal scores
    straddling shard boundaries with keep < hits, keep = 1, shards that own no sequence, shards with no hit, every hit in the
    last shard, fewer sequences than ranks, an empty database, a score window with hits above it - against the reference's
    hits_enter fed the whole database in seqno order (oracle.HitList, hits.cc:163-222)"""
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_adversarial_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    results = out.get(timeout=900)
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    for name, lens, scores, keep, lo_s, hi_s in _scenarios():
        hits, tot, obv, fhits, bounds = results[name]
        h = oracle.HitList(descriptions=keep, alignments=0, minscore=lo_s, maxscore=hi_s, expect=1e30, dbseqs=max(1, len(lens)),
                           dbsyms=max(1, int(np.sum(lens))), qlen=100)
        for i, s in enumerate(scores):
            h.enter(i, int(s))
        assert hits == [(x[0], x[1]) for x in h.hits()], name
        assert tot == int((np.asarray(scores) >= lo_s).sum()) and obv == int((np.asarray(scores) > hi_s).sum()), name
        # frames: each sequence's two entries stay adjacent, frame 0 before frame 4 (insertion order for equal (score, seqno))
        want6 = [(s, v, 0, f % 3, f // 3, f % 3) for s, v in hits for f in (0, 4)][:keep]
        assert fhits == want6, name


def test_bench_divides_host_threads_by_world_and_keeps_single_rank_sections_out_of_multi_rank_runs():
    """bench.py at N > 1: generation and the oracle verification of every rank use cores // world host threads (8 ranks on one
    host must not oversubscribe it 8-fold inside the driver's timing), and the sections that only make sense once - the CPU
    baseline, the cold open, the secondary workloads, the pair section - are guarded by world == 1"""
    import ast
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "gen_threads = max(1, cores // max(1, world))" in src
    tree = ast.parse(src)
    main = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "main")
    parents = {}
    for node in ast.walk(main):
        for child in ast.iter_child_nodes(node):
            parents[child] = node

    def guarded(call):
        n = call
        while n in parents:
            n = parents[n]
            if isinstance(n, ast.If) and "world == 1" in ast.unparse(n.test):
                return True
        return False

    single = {"cpu_baseline", "cold_open", "protein100m_section", "search_pair_topk"}
    seen = set()
    for node in ast.walk(main):
        if isinstance(node, ast.Call):
            name = node.func.id if isinstance(node.func, ast.Name) else node.func.attr if isinstance(node.func, ast.Attribute) else ""
            if name in single:
                seen.add(name)
                assert guarded(node), f"{name} runs at N > 1"
            if name == "nucleotide_section" and "a.secondary_nt_nseq" in ast.unparse(node):
                seen.add(name)
                assert guarded(node), "the nucleotide secondary section runs at N > 1"
            if name in ("synth_db", "synth_offsets"):
                assert "gen_threads" in ast.unparse(node), f"{name} in main() must use gen_threads"
            if name == "verify_against_oracle":
                assert "gen_threads" in ast.unparse(node) and "// world" in ast.unparse(node)
    assert seen == single | {"nucleotide_section"}, seen
