"""GPU (-m gpu): the HIP path through the C ABI against (a) the committed reference outputs,
(b) the oracle on the same seeded inputs, (c) size-independent properties at larger sizes.
Integer scores: the bar is bit-exact."""
import os

import numpy as np
import pytest

import cases
import oracle
import swipe_amd
from conftest import under_interpreter, ROOT, case_matrix, load_golden
from swipe_amd import blastdb, synth

pytestmark = pytest.mark.gpu
NAMES = [f.__name__[5:] for f in cases.ALL]
THREADS = os.cpu_count() or 1


def strands(case):
    return [case.query] if case.protein else [case.query, blastdb.revcomp_nt16(case.query)]


def open_case(case):
    db = swipe_amd.Database.from_sequences(case.seqs, symtype=1 if case.protein else 0)
    db.set_scoring(case_matrix(case, swipe_amd), case.gapopen, case.gapextend)
    return db


@pytest.mark.parametrize("name", NAMES)
def test_scores_equal_reference_for_every_sequence(name):
    case, g = cases.get(name), load_golden(name)
    db = open_case(case)
    for strand, q in enumerate(strands(case)):
        scores, c = db.search(q)
        want = [r[7] for r in g["raw"] if r[1] == strand]
        assert list(scores) == want
        assert c["narrow"] + (c["wide"] if c["narrow"] == 0 else 0) == len(case.seqs)
    db.close()


@pytest.mark.parametrize("name", NAMES)
def test_hit_list_equals_reference_cli(name):
    """ranks, scores, E-value and bit-score strings of `swipe -m 8/-m 7` (1 and 8 threads agree)."""
    case, g = cases.get(name), load_golden(name)
    cli = g["cli"]["1"]
    nsym = int(sum(len(s) for s in case.seqs))
    st = swipe_amd.stats_init(symtype=1 if case.protein else 0, matrix=case.matrix, match=case.match,
                              mismatch=case.mismatch, gapopen=case.gapopen, gapextend=case.gapextend,
                              qlen=len(case.query), db_seqcount=len(case.seqs), db_symcount=nsym)
    db = open_case(case)
    entries = []
    total = 0
    for strand, q in enumerate(strands(case)):
        hits, tot, obv, _ = db.search_topk(q, keep=case.keep, minscore=st.scorethreshold, maxscore=st.upperscorethreshold)
        total += tot
        entries += [(seqno, score, strand) for seqno, score in hits]
    entries.sort(key=lambda e: (-e[1], -e[0]))        # stable: plus-strand entries were entered first
    entries = entries[:case.keep]
    assert [e[0] for e in entries] == cli["seqno"]
    assert [e[1] for e in entries] == cli["score"]
    if st.available:
        assert ["%.2g" % st.evalue(e[1]) for e in entries] == cli["evalue"]
        assert ["%.1f" % st.bits(e[1]) for e in entries] == cli["bits"]
    if cli["strand"]:
        assert ["-" if e[2] else "+" for e in entries] == cli["strand"]
    assert total == sum(r[7] >= st.scorethreshold for r in g["raw"])
    db.close()


def test_open_blast_volumes_and_shard_ranges(tmp_path):
    case, g = cases.get("multivol"), load_golden("multivol")
    base = str(tmp_path / "mv")
    blastdb.write_db(base, case.seqs, protein=True, volumes=case.volumes)
    want = [r[7] for r in g["raw"]]
    db = swipe_amd.Database.open(base)
    info = db.info()
    assert info["seqcount"] == 400 and info["total_seqcount"] == 400
    db.set_scoring(swipe_amd.matrix_builtin("BLOSUM62"), 11, 1)
    assert list(db.search(case.query)[0]) == want
    db.close()
    # a shard that straddles two volumes keeps global sequence numbers
    sh = swipe_amd.Database.open(base, first_seqno=100, last_seqno=299)
    assert sh.info()["first_seqno"] == 100 and sh.info()["seqcount"] == 200 and sh.info()["total_seqcount"] == 400
    sh.set_scoring(swipe_amd.matrix_builtin("BLOSUM62"), 11, 1)
    assert list(sh.search(case.query)[0]) == want[100:300]
    hits, tot, obv, _ = sh.search_topk(case.query, keep=5, minscore=1)
    best = sorted(((s, i) for i, s in enumerate(want) if 100 <= i < 300), key=lambda t: (-t[0], -t[1]))[:5]
    assert hits == [(i, s) for s, i in best]
    sh.close()


def test_width_escalation_16_to_32_to_64():
    """Scores past the f16-exact range are re-queued to the 32-bit kernel, past 2^31 to the 64-bit one."""
    rtab = synth.residue_table_protein()
    q = synth._random_residues(12345, 1, 700, rtab)
    mut = q.copy()
    mut[::9] = rtab[(np.arange(len(mut[::9])) * 131) % 4096]
    seqs = [synth._random_residues(777 + k, 1, 200 + 13 * k, rtab) for k in range(40)]
    seqs += [q, mut, q[:450], np.concatenate([seqs[0], q, seqs[1]]), q[100:]]
    M = swipe_amd.matrix_builtin("BLOSUM62")
    db = swipe_amd.Database.from_sequences(seqs)
    db.set_scoring(M, 11, 1)
    scores, c = db.search(q)
    res, off = oracle.pack(seqs)
    want = oracle.search_all63(res, off, q, oracle.matrix_builtin("BLOSUM62"), 12, 1, threads=THREADS)
    assert np.array_equal(scores, want)
    assert c["narrow"] == len(seqs) and c["wide"] == int((want >= 2048 - 11).sum()) > 0 and c["full"] == 0
    db.close()
    # 64-bit: a matrix with 10^7 on the diagonal pushes a 300-residue self hit past 2^31
    letters = "ARNDCQEGHILKMFPSTWYV"
    text = "   " + "  ".join(letters) + "\n" + "\n".join(
        a + " " + " ".join("10000000" if a == b else "-3000000" for b in letters) for a in letters) + "\n"
    q2 = q[:300]
    seqs2 = [q2, q2[:200], q2[:250], mut[:300], seqs[3], seqs[4]]
    db = swipe_amd.Database.from_sequences(seqs2)
    db.set_scoring(swipe_amd.matrix_parse(text), 200, 50)
    scores, c = db.search(q2)
    res, off = oracle.pack(seqs2)
    want = oracle.search_all63(res, off, q2, oracle.matrix_parse(text), 250, 50)
    assert np.array_equal(scores, want) and want.max() == 3_000_000_000
    assert c["narrow"] == 0 and c["wide"] == len(seqs2) and c["full"] == int((want >= 2**31 - 10_000_000).sum()) > 0
    hits, _, _, _ = db.search_topk(q2, keep=3, minscore=1)
    assert hits[0] == (0, 3_000_000_000)
    db.close()


@pytest.mark.parametrize("qlen", [1, 2, 15, 16, 17, 127, 128, 129, 255, 375, 384, 385, 511, 640, 767, 768, 769, 1000,
                                  1024, 1025, 1536, 2100, 5000])
def test_every_rows_per_lane_variant(qlen):
    """query lengths around every 16*K boundary of the kernel templates"""
    rtab = synth.residue_table_protein()
    q = synth._random_residues(4242 + qlen, 1, qlen, rtab)
    res, off = swipe_amd.synth_db(5, 600, query=q)
    extra = [q, q[: max(1, qlen // 2)], np.zeros(0, np.uint8)]
    r2, o2 = oracle.pack([res[off[i]:off[i + 1]] for i in range(600)] + extra)
    db = swipe_amd.Database.from_arrays(r2, o2)
    db.set_scoring(swipe_amd.matrix_builtin("BLOSUM62"), 11, 1)
    scores, _ = db.search(q)
    want = oracle.search_all63(r2, o2, q, oracle.matrix_builtin("BLOSUM62"), 12, 1, threads=THREADS)
    assert np.array_equal(scores, want)
    db.close()


@pytest.mark.parametrize("lanes", [16, 8, 4, 2, 1])
def test_every_instantiation_of_the_row_shifted_kernel(lanes, monkeypatch):
    """rows per lane K = ceil(qlen / lanes) for every K in 1..48 of all three forms (16 lanes per sequence pair, 8 for
    queries of at most 384 rows, 4 for at most 192): one query per instantiation, at both ends of its window"""
    monkeypatch.setenv("SWA_LANES", str(lanes))
    rtab = synth.residue_table_protein()
    full = synth._random_residues(99, 1, 928, rtab)
    res, off = swipe_amd.synth_db(6, 400, query=full)
    seqs = [res[off[i]:off[i + 1]] for i in range(400)] + [full, full[100:500].copy(), full[::-1].copy(), np.zeros(0, np.uint8)]
    r2, o2 = oracle.pack(seqs)
    db = swipe_amd.Database.from_arrays(r2, o2)
    db.set_scoring(swipe_amd.matrix_builtin("BLOSUM62"), 11, 1)
    Mo = oracle.matrix_builtin("BLOSUM62")
    for K in range(1, 59 if lanes == 16 else 49):
        for qlen in (lanes * K - lanes + 1, lanes * K):
            q = full[:qlen]
            scores, c = db.search(q)
            assert c["narrow_rows"] == K and c["narrow_shifted"] == {16: 1, 8: 2, 4: 3, 2: 7, 1: 11}[lanes]
            assert np.array_equal(scores, oracle.search_all63(r2, o2, q, Mo, 12, 1, threads=THREADS)), (K, qlen)
    db.close()


def _expected_topk(scores, keep, minscore, maxscore=1 << 62):
    order = sorted((i for i in range(len(scores)) if scores[i] >= minscore), key=lambda i: (-int(scores[i]), -i))
    kept = [i for i in order if scores[i] <= maxscore]     # hits.cc:174-178: "obvious" hits are counted, not listed
    return [(i, int(scores[i])) for i in kept[:keep]], len(order), int((scores > maxscore).sum())


@pytest.mark.parametrize("lanes", [16, 8, 4, 2, 1, -1])
def test_bound_build_of_the_first_pass_gives_the_same_hits(lanes, monkeypatch):
    """top-K searches may run the bound build of the row-shifted kernel (6.5 instructions per cell pair, result at most
    15 R above the score, everything at or above the threshold recomputed by the 32-bit kernel): every K = 25..48 (58)
    of every chain length (4 lanes from 11, 2 lanes from 5), hits planted at every distance from the threshold, thresholds from "everything comes back"
    to "nothing does", gap extension penalties 1..3 - hit list, totalhits and obvious must equal the exact ones"""
    other = lanes == -1                                      # the one-lane build in the column order it does not pick by itself
    lanes = abs(lanes)
    monkeypatch.setenv("SWA_LANES", str(lanes))
    monkeypatch.setenv("SWA_BOUND", "1")
    rtab = synth.residue_table_protein()
    full = synth._random_residues(99, 1, 930, rtab)
    rng = np.random.default_rng(lanes)
    res, off = swipe_amd.synth_db(6, 1500, query=full)
    seqs = [res[off[i]:off[i + 1]] for i in range(1500)]
    for k in range(120):                                   # pieces of the query of every length: scores across every threshold
        a = int(rng.integers(0, 700))
        piece = full[a:a + int(rng.integers(8, 90))].copy()
        mut = rng.random(len(piece)) < rng.random() * 0.4
        piece[mut] = rtab[rng.integers(0, len(rtab), int(mut.sum()))]
        seqs.append(np.concatenate([seqs[k][:int(rng.integers(0, 60))], piece, seqs[k + 1][:int(rng.integers(0, 60))]]))
    seqs += [full, full[::2].copy(), np.zeros(0, np.uint8)]
    r2, o2 = oracle.pack(seqs)
    db = swipe_amd.Database.from_arrays(r2, o2)
    Mo = oracle.matrix_builtin("BLOSUM62")
    n = 0
    for K in range({16: 25, 8: 25, 4: 11, 2: 5, 1: 1}[lanes], {16: 58, 8: 62, 4: 62, 2: 62, 1: 48 if other else 60}[lanes] + 1):   # (chains: 49..62 rows exist as the bound build only)
        go, ge = ((11, 1), (10, 2), (9, 3))[K % 3]       # (one lane: 49..60 rows exist as the bound build only)
        db.set_scoring(swipe_amd.matrix_builtin("BLOSUM62"), go, ge)
        q = full[:lanes * K - (K % lanes)]
        if other:
            db.set_option("pipe", 0 if K <= 28 else 1)       # two columns at a time up to 28 rows by default, one beyond
        want = oracle.search_all63(r2, o2, q, Mo, go + ge, ge, threads=THREADS)
        for minscore, maxscore in ((1, 1 << 62), (35, 90), (60, 1 << 62), (100, 300), (400, 1 << 62)):
            hits, tot, obv, c = db.search_topk(q, keep=40, minscore=minscore, maxscore=maxscore)
            assert c["narrow_rows"] == K and c["narrow_shifted"] == 8, (K, c)
            assert (hits, tot, obv) == _expected_topk(want, 40, minscore, maxscore), (K, minscore)
            n += c["wide"]
    assert n > 0
    # all scores are still exact when they are asked for
    scores, c = db.search(q)
    assert c["narrow_shifted"] in (1, 2, 3, 7, 11) and np.array_equal(scores, want)
    db.close()


@pytest.mark.late
@pytest.mark.parametrize("concat", [2, 5, 8])
@pytest.mark.parametrize("lanes", [16, 8, 4, 2])
def test_bound_build_with_sequences_back_to_back(lanes, concat, monkeypatch):
    """round 6: the bound build works through `concat` sets of batches per chain WITHOUT draining the chain or resetting its
    state in between (sw_cb_kernel.inc: what flows across a junction can only raise a bound, and what a bound sends back is
    recomputed exactly) - the skew of a chain is paid once per item instead of once per batch.  Hits, totalhits and obvious
    must be the exact ones at every threshold, for the shortest, a middle and the longest build of every chain length, with
    hits planted in front of ordinary sequences, sequences shorter than a period, empty ones, and a tail of the queue that
    is handed out one set at a time (here: the last quarter of the sets).  Builds at two waves per SIMD (30+ rows) run as blocks of
    8 waves with the profile twice, the second copy - N R, which step 0 of a period reads instead of renormalising H (option
    twin; beyond 56 rows the copy is out of reach of ds_read's immediate offset): same hits, same re-queue counts"""
    monkeypatch.setenv("SWA_LANES", str(lanes))
    monkeypatch.setenv("SWA_BOUND", "1")
    rtab = synth.residue_table_protein()
    full = synth._random_residues(99, 1, 1000, rtab)
    rng = np.random.default_rng(100 * lanes + concat)
    res, off = swipe_amd.synth_db(6, 1300, query=full)
    seqs = [res[off[i]:off[i + 1]] for i in range(1300)]
    for k in range(150):                                   # pieces of the query: scores across every threshold, hits of every size
        a = int(rng.integers(0, 700))
        piece = full[a:a + int(rng.integers(8, 200))].copy()
        mut = rng.random(len(piece)) < rng.random() * 0.4
        piece[mut] = rtab[rng.integers(0, len(rtab), int(mut.sum()))]
        seqs.append(np.concatenate([seqs[k][:int(rng.integers(0, 60))], piece, seqs[k + 1][:int(rng.integers(0, 3))]]))    # the hit ends at the junction
    seqs += [synth._random_residues(k, 1, int(rng.integers(1, 16)), rtab) for k in range(200)]     # shorter than a period
    seqs += [full, full[::2].copy(), np.zeros(0, np.uint8), np.zeros(0, np.uint8)]
    r2, o2 = oracle.pack(seqs)
    db = swipe_amd.Database.from_arrays(r2, o2)
    Mo = oracle.matrix_builtin("BLOSUM62")
    lo, hi = {16: (25, 58), 8: (25, 62), 4: (11, 62), 2: (5, 62)}[lanes]
    back = single = 0
    for K in (lo, (lo + hi) // 2 | 1, hi):
        go, ge = ((11, 1), (10, 2), (9, 3))[K % 3]
        db.set_scoring(swipe_amd.matrix_builtin("BLOSUM62"), go, ge)
        q = full[:lanes * K - (K % lanes)]
        want = oracle.search_all63(r2, o2, q, Mo, go + ge, ge, threads=THREADS)
        for minscore, maxscore in ((1, 1 << 62), (35, 90), (60, 1 << 62), (100, 300), (400, 1 << 62)):
            got = {}
            db.set_option("concat_tail", (len(seqs) // 8 // (16 // lanes)) // 4)      # the last quarter of the sets one at a time
            for m, twin in ((1, 1), (concat, 1), (concat, 0), (1, 0)):
                db.set_option("concat", m)
                db.set_option("twin", twin)       # builds at two waves per SIMD: 8-wave blocks with the profile twice / the round-3 form
                hits, tot, obv, c = db.search_topk(q, keep=40, minscore=minscore, maxscore=maxscore)
                assert c["narrow_rows"] == K and c["narrow_shifted"] == 8, (K, c)
                assert (hits, tot, obv) == _expected_topk(want, 40, minscore, maxscore), (K, minscore, m, twin)
                got[(m, twin)] = c["wide"]
            assert got[(concat, 1)] == got[(concat, 0)] and got[(1, 1)] == got[(1, 0)]     # the second copy of the profile changes no value
            back += got[(concat, 1)]
            single += got[(1, 1)]
    assert back >= single > 0            # what crosses a junction can only send MORE sequences back
    db.close()


def test_one_lane_bound_build_is_the_default_up_to_60_rows():
    """top-K searches of 49..60-row queries take the one-lane bound build by themselves (no option set), 61 rows and
    exact searches of the same queries go to 2-lane chains; same hits either way"""
    rtab = synth.residue_table_protein()
    full = synth._random_residues(5, 1, 80, rtab)
    res, off = swipe_amd.synth_db(8, 3000, query=full)
    db = swipe_amd.Database.from_arrays(res, off)
    db.set_scoring(swipe_amd.matrix_builtin("BLOSUM62"), 11, 1)
    Mo = oracle.matrix_builtin("BLOSUM62")
    for qlen, form, rows in ((49, 8, (49,)), (60, 8, (60,)), (61, 8, (31, 32, 33, 34))):   # 61: two lanes of ceil(61 / 2) rows,
        q = full[:qlen]                                                                    # or up to 3 more (kernel_choice.cpp)
        want = oracle.search_all63(res, off, q, Mo, 12, 1, threads=THREADS)
        hits, tot, obv, c = db.search_topk(q, keep=30, minscore=70)
        assert c["narrow_shifted"] == form and c["narrow_rows"] in rows, c
        assert (hits, tot, obv) == _expected_topk(want, 30, 70)
        scores, c = db.search(q)
        assert c["narrow_shifted"] == 7 and np.array_equal(scores, want)
    db.close()


def test_bound_build_under_a_translated_search(monkeypatch):
    """tblastn: a protein query against a nucleotide shard held as its six translations goes through swa_search_frames_topk
    - same frame-tagged hits with the bound build as with the exact first pass"""
    q = cases.Q375
    res, off = swipe_amd.synth_db(3, 3000, protein=False)
    db = swipe_amd.Database.from_arrays(res, off, translate_gencode=1)
    db.set_scoring(swipe_amd.matrix_builtin("BLOSUM62"), 11, 1)
    out = {}
    for mode in ("0", "1"):
        db.set_option("bound", mode)
        for minscore in (30, 45, 70):
            hits, tot, obv, c = db.search_frames_topk([q], keep=100, minscore=minscore)
            assert c["narrow_shifted"] == (8 if mode == "1" else 2)
            out[(mode, minscore)] = (hits, tot, obv)
    for minscore in (30, 45, 70):
        assert out[("0", minscore)] == out[("1", minscore)] and out[("0", minscore)][1] > 0 or minscore == 70
        assert out[("0", minscore)] == out[("1", minscore)]
    db.close()


@pytest.mark.parametrize("lanes", [16, 8, 4])
def test_bound_build_of_the_two_query_kernel(lanes, monkeypatch):
    """pairs of protein queries (frames of a translated search) with a score threshold: bound build of swa_dual_kernel,
    K = 17..32 rows per lane for chains of 16 and 8 lanes - the merged hit list of both queries, totalhits and obvious
    equal the exact ones for thresholds from "everything comes back" to "nothing does" """
    monkeypatch.setenv("SWA_LANES", str(lanes))
    monkeypatch.setenv("SWA_BOUND", "1")
    rtab = synth.residue_table_protein()
    full = synth._random_residues(4321, 1, 520, rtab)
    rng = np.random.default_rng(lanes + 1)
    res, off = swipe_amd.synth_db(16, 1200, query=full)
    seqs = [res[off[i]:off[i + 1]] for i in range(1200)]
    rev = full[::-1].copy()
    for k in range(100):
        src = full if k % 2 else rev
        a = int(rng.integers(0, 430))
        piece = src[a:a + int(rng.integers(8, 90))].copy()
        mut = rng.random(len(piece)) < rng.random() * 0.4
        piece[mut] = rtab[rng.integers(0, len(rtab), int(mut.sum()))]
        seqs.append(np.concatenate([seqs[k][:int(rng.integers(0, 50))], piece, seqs[k + 1][:int(rng.integers(0, 50))]]))
    seqs += [full, rev, np.zeros(0, np.uint8)]
    r2, o2 = oracle.pack(seqs)
    db = swipe_amd.Database.from_arrays(r2, o2)
    Mo = oracle.matrix_builtin("BLOSUM62")
    for K in range(17, 33 if lanes == 16 else 63):           # 4- and 8-lane chains: up to 62 rows (sw_cb_dual_g4 / _long.hip)
        go, ge = ((11, 1), (10, 2))[K % 2]
        db.set_scoring(swipe_amd.matrix_builtin("BLOSUM62"), go, ge)
        qlen = lanes * K - (K % lanes)
        q1 = full[:qlen]
        q2 = rev[:qlen].copy()
        w1 = oracle.search_all63(r2, o2, q1, Mo, go + ge, ge, threads=THREADS)
        w2 = oracle.search_all63(r2, o2, q2, Mo, go + ge, ge, threads=THREADS)
        for minscore, maxscore in ((1, 1 << 62), (40, 120), (70, 1 << 62), (500, 1 << 62)):
            hits, tot, obv, c = db.search2_topk(q1, q2, keep=30, minscore=minscore, maxscore=maxscore)
            assert c["narrow_rows"] == K and c["narrow_shifted"] == 10, (K, c)
            want = sorted([(int(s), i, 0) for i, s in enumerate(w1) if minscore <= s <= maxscore] +
                          [(int(s), i, 1) for i, s in enumerate(w2) if minscore <= s <= maxscore], key=lambda t: (-t[0], -t[1], t[2]))[:30]
            assert hits == [(i, s, w) for s, i, w in want], (K, minscore)
            assert tot == int((w1 >= minscore).sum() + (w2 >= minscore).sum()) and obv == int((w1 > maxscore).sum() + (w2 > maxscore).sum())
    db.close()


@pytest.mark.late
@pytest.mark.parametrize("lanes", [16, 8, 4])
def test_bound_build_of_the_two_query_kernel_with_sequences_back_to_back(lanes, monkeypatch):
    """round 6: the two-query bound build works through sets of batches back to back as the one-query one does
    (sw_cb_dual_kernel.inc): the shortest, a middle and the longest build of each chain length, 1 / 3 / 8 sets per item with the
    last quarter of the queue handed out singly, hits that end at a junction, sequences shorter than a period - both queries'
    merged hit list, totalhits and obvious equal the exact ones, and more sets never send fewer sequences back"""
    monkeypatch.setenv("SWA_LANES", str(lanes))
    monkeypatch.setenv("SWA_BOUND", "1")
    rtab = synth.residue_table_protein()
    full = synth._random_residues(4321, 1, 1000, rtab)
    rng = np.random.default_rng(lanes + 7)
    res, off = swipe_amd.synth_db(16, 1100, query=full)
    seqs = [res[off[i]:off[i + 1]] for i in range(1100)]
    rev = full[::-1].copy()
    for k in range(140):
        src = full if k % 2 else rev
        a = int(rng.integers(0, 430))
        piece = src[a:a + int(rng.integers(8, 160))].copy()
        mut = rng.random(len(piece)) < rng.random() * 0.4
        piece[mut] = rtab[rng.integers(0, len(rtab), int(mut.sum()))]
        seqs.append(np.concatenate([seqs[k][:int(rng.integers(0, 50))], piece, seqs[k + 1][:int(rng.integers(0, 3))]]))
    seqs += [synth._random_residues(k, 1, int(rng.integers(1, 16)), rtab) for k in range(150)]
    seqs += [full, rev, np.zeros(0, np.uint8), np.zeros(0, np.uint8)]
    r2, o2 = oracle.pack(seqs)
    db = swipe_amd.Database.from_arrays(r2, o2)
    Mo = oracle.matrix_builtin("BLOSUM62")
    per_wave = 1 if lanes == 16 else 8 // lanes
    db.set_option("concat_tail", (len(seqs) // (4 if lanes == 16 else 8) // per_wave) // 4)
    back = single = 0
    for K in (17, 25, 32 if lanes == 16 else 62):
        go, ge = ((11, 1), (10, 2))[K % 2]
        db.set_scoring(swipe_amd.matrix_builtin("BLOSUM62"), go, ge)
        qlen = lanes * K - (K % lanes)
        q1 = full[:qlen]
        q2 = rev[:qlen].copy()
        w1 = oracle.search_all63(r2, o2, q1, Mo, go + ge, ge, threads=THREADS)
        w2 = oracle.search_all63(r2, o2, q2, Mo, go + ge, ge, threads=THREADS)
        for minscore, maxscore in ((1, 1 << 62), (40, 120), (70, 1 << 62), (500, 1 << 62)):
            want = sorted([(int(s), i, 0) for i, s in enumerate(w1) if minscore <= s <= maxscore] +
                          [(int(s), i, 1) for i, s in enumerate(w2) if minscore <= s <= maxscore], key=lambda t: (-t[0], -t[1], t[2]))[:30]
            got = {}
            for m in (1, 3, 8):
                db.set_option("concat", m)
                hits, tot, obv, c = db.search2_topk(q1, q2, keep=30, minscore=minscore, maxscore=maxscore)
                assert c["narrow_rows"] == K and c["narrow_shifted"] == 10, (K, c)
                assert hits == [(i, s, w) for s, i, w in want], (K, minscore, m)
                assert tot == int((w1 >= minscore).sum() + (w2 >= minscore).sum()) and obv == int((w1 > maxscore).sum() + (w2 > maxscore).sum())
                got[m] = c["wide"]
            back += got[8]
            single += got[1]
    assert back >= single > 0
    db.close()


def test_bound_build_of_the_passes_of_long_queries(monkeypatch):
    """top-K searches of queries longer than 928 rows: passes of the bound build, 16 x K rows with K = 30..56, the hand-over
    stored without the step bias and re-biased on arrival; every K with two passes, then up to seven passes, hits that
    straddle the pass boundaries, several runs of batches - hit list, totalhits, obvious equal to the exact ones"""
    monkeypatch.setenv("SWA_BOUND", "1")
    rtab = synth.residue_table_protein()
    full = synth._random_residues(1234, 1, 5200, rtab)
    res, off = swipe_amd.synth_db(9, 600, query=full[:1200])
    base = [res[off[i]:off[i + 1]] for i in range(600)]
    Mo = oracle.matrix_builtin("BLOSUM62")
    rng = np.random.default_rng(3)
    lens = [32 * K for K in range(30, 57)] + [929, 1793, 48 * 40, 2688, 2689, 5200]
    for n, qlen in enumerate(lens):
        monkeypatch.setenv("SWA_BOUNDARY_MB", "1" if n % 3 == 0 else "4096")
        q = full[:qlen]
        npass = -(-qlen // (16 * 56))
        K = max(30, -(-qlen // (16 * npass)))
        edge = 16 * K
        planted = [q, q[edge - 120:edge + 130].copy(), q[:300].copy(), q[qlen - 320:].copy(), q[edge - 12:edge + 12].copy(),
                   q[edge * (npass - 1) - 200:edge * (npass - 1) + 60].copy(), np.concatenate([base[0], q[edge - 40:edge + 30], base[1]]),
                   q[100:qlen - 100:2].copy(), np.zeros(0, np.uint8)]
        for k in range(40):                                # short pieces: scores around every threshold
            a = int(rng.integers(0, qlen - 100))
            planted.append(np.concatenate([base[k][:30], q[a:a + int(rng.integers(8, 70))], base[k + 1][:30]]))
        r2, o2 = oracle.pack(base + planted)
        go, ge = ((11, 1), (10, 2))[n % 2]
        db = swipe_amd.Database.from_arrays(r2, o2)
        db.set_scoring(swipe_amd.matrix_builtin("BLOSUM62"), go, ge)
        want = oracle.search_all63(r2, o2, q, Mo, go + ge, ge, threads=THREADS)
        for minscore, maxscore in ((1, 1 << 62), (45, 200), (90, 1 << 62), (700, 1 << 62)):
            hits, tot, obv, c = db.search_topk(q, keep=40, minscore=minscore, maxscore=maxscore)
            assert c["narrow_shifted"] == 9 and c["narrow_rows"] == K, (qlen, c)
            assert (hits, tot, obv) == _expected_topk(want, 40, minscore, maxscore), (qlen, minscore)
        db.close()


@pytest.mark.parametrize("wave", ["0", "1"])
@pytest.mark.parametrize("qlen", [600, 1300, 2048, 2049, 4500])
def test_requeue_by_batches_and_by_wave(qlen, wave, monkeypatch):
    """sequences that leave the packed range are recomputed in 32 bits: a short list by one wave per sequence (in passes of
    2 048 rows), otherwise by the batch kernel (SWA_WAVE_REQUEUE=0 forces it) - same scores"""
    monkeypatch.setenv("SWA_WAVE_REQUEUE", wave)
    rtab = synth.residue_table_protein()
    q = synth._random_residues(808 + qlen, 1, qlen, rtab)
    res, off = swipe_amd.synth_db(12, 1200, query=q)
    seqs = [res[off[i]:off[i + 1]] for i in range(1200)]
    seqs += [q, q[::-1].copy(), np.concatenate([q, q]), q[: qlen // 2].copy(), synth._random_residues(5, 1, 9000, rtab), np.zeros(0, np.uint8)]
    r2, o2 = oracle.pack(seqs)
    db = swipe_amd.Database.from_arrays(r2, o2)
    db.set_scoring(swipe_amd.matrix_builtin("BLOSUM62"), 11, 1)
    scores, c = db.search(q)
    assert c["wide"] >= 2 and c["full"] == 0
    assert np.array_equal(scores, oracle.search_all63(r2, o2, q, oracle.matrix_builtin("BLOSUM62"), 12, 1, threads=THREADS))
    db.close()


def _requeue_heavy_protein(qlen):
    """a database with 130 relatives of the query that leave the packed range under BLOSUM62 x 5 whatever the query length
    (so that the re-queue has a real list), a subject of more than 6 000 columns that carries one more, and an empty one"""
    rtab = synth.residue_table_protein()
    q = synth._random_residues(9100 + qlen, 1, qlen, rtab)
    res, off = swipe_amd.synth_db(14, 500, query=q)
    seqs = [res[off[i]:off[i + 1]] for i in range(500)]
    rng = np.random.default_rng(qlen)
    for k in range(130):                                    # 2..10 % of the residues replaced, random flanks of random length
        rel = q.copy()
        m = rng.random(qlen) < 0.02 * (1 + k % 5)
        rel[m] = rtab[rng.integers(0, len(rtab), int(m.sum()))]
        seqs.insert(int(rng.integers(0, len(seqs))), np.concatenate([synth._random_residues(k, 1, int(rng.integers(0, 120)), rtab), rel,
                                                                     synth._random_residues(k + 300, 1, int(rng.integers(0, 120)), rtab)]))
    rel = q.copy()
    rel[::7] = rtab[(np.arange(len(rel[::7])) * 131) % len(rtab)]
    long_one = np.concatenate([synth._random_residues(6, 1, 3100, rtab), rel, synth._random_residues(7, 1, 3100, rtab)])
    seqs += [q, q[::-1].copy(), np.concatenate([q, q]), q[: qlen // 2].copy(), long_one, np.zeros(0, np.uint8), q[:1].copy()]
    return q, oracle.pack(seqs)


@pytest.mark.late
@pytest.mark.parametrize("qlen", [128, 129, 256, 257, 512, 513, 768, 769, 1024, 1025])
def test_both_device_requeue_forms_by_name(qlen):
    """VERDICT r5 item 3 / ADVICE r5: the two forms of the device-driven re-queue (search16's role, search16.cc:320-546, behind
    the escalation of swipe.cc:1492-1538) pinned by the option that selects them: requeue_block = 0 is
    swa_requeue_wave_kernel (one wave per re-queued sequence), requeue_block = 1 swa_requeue_block_kernel (four waves,
    queries of at most 1 024 rows; beyond that the wave form whatever the option says).  Every boundary of both kernels'
    rows-per-lane tables, at least 100 entries on the list, a subject of 6 000+ columns and an empty one: scores equal the
    oracle's, the two forms' hit lists are identical, and counters.requeue_form says which kernel ran."""
    q, (r2, o2) = _requeue_heavy_protein(qlen)
    M5 = swipe_amd.matrix_builtin("BLOSUM62") * 5
    want = oracle.search_all63(r2, o2, q, oracle.matrix_builtin("BLOSUM62") * 5, 12, 1, threads=THREADS)
    db = swipe_amd.Database.from_arrays(r2, o2)
    db.set_scoring(M5, 11, 1)
    lists = []
    for form in (0, 1):
        db.set_option("requeue_block", form)
        scores, c = db.search(q)
        assert c["requeue_form"] == (2 if form == 1 and qlen <= 1024 else 1), (qlen, form, c)
        assert c["wide"] >= 100 and c["full"] == 0, (qlen, form, c)
        assert np.array_equal(scores, want), (qlen, form, np.flatnonzero(scores != want)[:10])
        hits, tot, obv, ct = db.search_topk(q, keep=60, minscore=200, maxscore=int(want.max()) - 1)
        assert ct["requeue_form"] == c["requeue_form"]
        assert (hits, tot, obv) == _expected_topk(want, 60, 200, int(want.max()) - 1), (qlen, form)
        lists.append((hits, tot, obv))
    assert lists[0] == lists[1]
    db.set_option("requeue_block", None)
    assert db.search(q)[1]["requeue_form"] == 1          # the default is the form that has been measured on hardware
    db.close()


@pytest.mark.late
@pytest.mark.parametrize("qlen", [129, 1000, 1024])
def test_both_device_requeue_forms_two_query_path(qlen):
    """the same pin for the two-query path (both strands of a nucleotide query, swipe.cc:1403): each strand has a list of its
    own and a re-queue launch of its own - both lists at least 100 entries long, both forms, scores of both strands equal
    the oracle's"""
    rtab = synth.residue_table_nucleotide()
    q = synth._random_residues(77 + qlen, 1, qlen, rtab)
    qm = blastdb.revcomp_nt16(q)
    res, off = swipe_amd.synth_db(4, 300, protein=False)
    seqs = [res[off[i]:off[i + 1]] for i in range(300)]
    rng = np.random.default_rng(qlen)
    for k in range(220):                                    # relatives of either strand, inside random flanks of random length
        rel = (q if k % 2 == 0 else qm).copy()
        m = rng.random(qlen) < 0.01 * (1 + k % 5)
        rel[m] = rtab[rng.integers(0, 4, int(m.sum()))]
        seqs.append(np.concatenate([synth._random_residues(k, 1, int(rng.integers(0, 150)), rtab), rel, synth._random_residues(k + 500, 1, int(rng.integers(0, 150)), rtab)]))
    seqs += [np.concatenate([synth._random_residues(3, 1, 3000, rtab), q, synth._random_residues(4, 1, 3200, rtab)]), np.zeros(0, np.uint8), q[:1].copy()]
    r2, o2 = oracle.pack(seqs)
    db = swipe_amd.Database.from_arrays(r2, o2, symtype=0)
    db.set_scoring(swipe_amd.matrix_nucleotide(19, -20), 30, 8)        # a match of 19 takes a relative out of the packed range at 129 rows already
    Mo = oracle.matrix_nucleotide(19, -20)
    w1 = oracle.search_all63(r2, o2, q, Mo, 38, 8, threads=THREADS)
    w2 = oracle.search_all63(r2, o2, qm, Mo, 38, 8, threads=THREADS)
    want = sorted([(int(s), i, 0) for i, s in enumerate(w1) if s >= 300] + [(int(s), i, 1) for i, s in enumerate(w2) if s >= 300], key=lambda t: (-t[0], -t[1], t[2]))[:80]
    got = []
    for form in (0, 1):
        db.set_option("requeue_block", form)
        s1, s2, c = db.search2(q, qm)
        assert c["requeue_form"] == (2 if form == 1 else 1) and c["wide"] >= 200 and c["full"] == 0, (qlen, form, c)
        assert np.array_equal(s1, w1) and np.array_equal(s2, w2), (qlen, form)
        hits, tot, obv, _ = db.search2_topk(q, qm, keep=80, minscore=300)
        assert hits == [(i, s, w) for s, i, w in want], (qlen, form)
        got.append((hits, tot, obv))
    assert got[0] == got[1]
    db.close()


def test_bound_build_is_dropped_when_too_much_comes_back(monkeypatch):
    """auto mode: the bound build runs only for thresholds well above its slack, and a search that sends more than 2 % of
    the sequences back switches it off for that query length at thresholds up to that one, until the scoring system
    changes; results are exact either way"""
    monkeypatch.delenv("SWA_BOUND", raising=False)
    q = cases.Q375
    rtab = synth.residue_table_protein()
    rng = np.random.default_rng(5)
    res, off = swipe_amd.synth_db(6, 2000, query=q)
    seqs = [res[off[i]:off[i + 1]] for i in range(2000)]
    r2, o2 = oracle.pack(seqs)
    Mo = oracle.matrix_builtin("BLOSUM62")
    want = oracle.search_all63(r2, o2, q, Mo, 12, 1, threads=THREADS)
    db = swipe_amd.Database.from_arrays(r2, o2)
    db.set_scoring(swipe_amd.matrix_builtin("BLOSUM62"), 11, 1)
    hits, tot, obv, c = db.search_topk(q, keep=50, minscore=30)            # threshold below 4 x 16 R: exact kernel
    assert c["narrow_shifted"] == 2 and (hits, tot, obv) == _expected_topk(want, 50, 30)
    hits, tot, obv, c = db.search_topk(q, keep=50, minscore=80)
    assert c["narrow_shifted"] == 8 and (hits, tot, obv) == _expected_topk(want, 50, 80)
    # a family database: a tenth of the sequences are relatives of the query
    fam = list(seqs)
    for k in range(0, 2000, 10):
        piece = q.copy()
        mut = rng.random(len(piece)) < 0.5
        piece[mut] = rtab[rng.integers(0, len(rtab), int(mut.sum()))]
        fam[k] = piece
    r3, o3 = oracle.pack(fam)
    want3 = oracle.search_all63(r3, o3, q, Mo, 12, 1, threads=THREADS)
    db3 = swipe_amd.Database.from_arrays(r3, o3)
    db3.set_scoring(swipe_amd.matrix_builtin("BLOSUM62"), 11, 1)
    hits, tot, obv, c = db3.search_topk(q, keep=50, minscore=80)
    assert tot > 40 and c["narrow_shifted"] == 2 and (hits, tot, obv) == _expected_topk(want3, 50, 80)     # fell back
    hits, tot, obv, c = db3.search_topk(q, keep=50, minscore=80)
    assert c["narrow_shifted"] == 2 and (hits, tot, obv) == _expected_topk(want3, 50, 80)                  # and stays off
    hits, tot, obv, c = db3.search_topk(q, keep=50, minscore=900)          # ... for that query and threshold only
    assert c["narrow_shifted"] == 8 and (hits, tot, obv) == _expected_topk(want3, 50, 900)
    hits, tot, obv, c = db3.search_topk(q, keep=50, minscore=70)           # a lower threshold sends back even more
    assert c["narrow_shifted"] == 2 and (hits, tot, obv) == _expected_topk(want3, 50, 70)
    db3.set_scoring(swipe_amd.matrix_builtin("BLOSUM62"), 11, 1)
    hits, tot, obv, c = db3.search_topk(q, keep=50, minscore=900)
    assert c["narrow_shifted"] == 8 and (hits, tot, obv) == _expected_topk(want3, 50, 900)
    # round 3: the fallback is per QUERY, not per query length - another query of the same length, unrelated to the family,
    # still gets the bound build, whichever of the two is searched first (throughput does not depend on query order)
    other = synth._random_residues(99, 1, len(q), rtab)
    wanto = oracle.search_all63(r3, o3, other, Mo, 12, 1, threads=THREADS)
    hits, tot, obv, c = db3.search_topk(q, keep=50, minscore=80)
    assert c["narrow_shifted"] == 2
    hits, tot, obv, c = db3.search_topk(other, keep=50, minscore=80)
    assert c["narrow_shifted"] == 8 and (hits, tot, obv) == _expected_topk(wanto, 50, 80)
    hits, tot, obv, c = db3.search_topk(q, keep=50, minscore=80)
    assert c["narrow_shifted"] == 2 and (hits, tot, obv) == _expected_topk(want3, 50, 80)
    db.close()
    db3.close()


@pytest.mark.parametrize("lanes", [8, 4])
def test_pipelined_profile_build_of_the_split_kernel(lanes, monkeypatch):
    """the builds of swa_narrow_split_kernel: all profile units staged / pipelined one unit ahead (K = 30..36) /
    pipelined across steps (K = 45..48)"""
    monkeypatch.setenv("SWA_LANES", str(lanes))
    rtab = synth.residue_table_protein()
    full = synth._random_residues(98, 1, 300, rtab)
    res, off = swipe_amd.synth_db(8, 300, query=full)
    r2, o2 = oracle.pack([res[off[i]:off[i + 1]] for i in range(300)] + [full, full[50:250].copy()])
    db = swipe_amd.Database.from_arrays(r2, o2)
    db.set_scoring(swipe_amd.matrix_builtin("BLOSUM62"), 11, 1)
    Mo = oracle.matrix_builtin("BLOSUM62")
    for K in list(range(30, 37)) + [45, 46, 47, 48]:
        if lanes * K - 1 > len(full):
            full = np.concatenate([full, synth._random_residues(97, 1, lanes * 48 - len(full), rtab)])
        q = full[: lanes * K - 1]
        want = oracle.search_all63(r2, o2, q, Mo, 12, 1, threads=THREADS)
        for pipe in (("0", "1") if K < 40 else ("0", "2")):
            db.set_option("pipe", pipe)
            scores, c = db.search(q)
            assert c["narrow_rows"] == K and np.array_equal(scores, want), (K, pipe)
    db.close()


@pytest.mark.parametrize("variant", ["1", "2"])
@pytest.mark.parametrize("gaps", [(11, 1), (5, 2), (0, 3), (14, 4), (30, 20)])
def test_both_narrow_kernel_forms(monkeypatch, variant, gaps):
    """plain (SWA_NARROW_VARIANT=1) and row-shifted (=2) packed-f16 kernels, several gap systems"""
    monkeypatch.setenv("SWA_NARROW_VARIANT", variant)
    q = cases.Q375
    res, off = swipe_amd.synth_db(9, 4000, query=q)
    seqs = [res[off[i]:off[i + 1]] for i in range(4000)] + cases.case_p1k().seqs[1000:] + [q, q[:100], np.zeros(0, np.uint8)]
    r2, o2 = oracle.pack(seqs)
    db = swipe_amd.Database.from_arrays(r2, o2)
    db.set_scoring(swipe_amd.matrix_builtin("BLOSUM62"), *gaps)
    scores, c = db.search(q)
    want = oracle.search_all63(r2, o2, q, oracle.matrix_builtin("BLOSUM62"), gaps[0] + gaps[1], gaps[1], threads=THREADS)
    assert np.array_equal(scores, want)
    assert c["narrow"] == (len(seqs) if gaps[1] <= 16 else 0)      # extension 20: f16 range too small, 32 bit throughout
    db.close()


@pytest.mark.parametrize("name", NAMES)
def test_alignment_end_points_equal_search16s(name):
    """swa_search_endpoints against the reference's search16s for every sequence it does not saturate on"""
    case, g = cases.get(name), load_golden(name)
    db = open_case(case)
    for strand, q in enumerate(strands(case)):
        rows = [r for r in g["raw"] if r[1] == strand]
        sc, bp, bq = db.search_endpoints(q, [r[0] for r in rows])
        for r, a, b, c in zip(rows, sc, bp, bq):
            assert a == r[7]
            if r[8] < g["scorelimit16"]:
                assert (a, b, c) == (r[8], r[9], r[10])
    db.close()


def test_empty_inputs_and_errors():
    M = swipe_amd.matrix_builtin("BLOSUM62")
    db = swipe_amd.Database.from_sequences([cases.Q375, np.zeros(0, np.uint8)])
    with pytest.raises(swipe_amd.SwaError) as e:
        db.search(cases.Q375)
    assert "swa_set_scoring" in str(e.value)
    db.set_scoring(M, 11, 1)
    assert list(db.search(np.zeros(0, np.uint8))[0]) == [0, 0]
    assert list(db.search(cases.Q375)[0]) == [1957, 0]
    with pytest.raises(swipe_amd.SwaError):
        db.search(np.full(5, 40, np.uint8))
    db.close()
    empty = swipe_amd.Database.from_sequences([])
    empty.set_scoring(M, 11, 1)
    assert len(empty.search(cases.Q375)[0]) == 0
    assert empty.search_topk(cases.Q375, keep=5)[0] == []
    empty.close()


def test_properties_at_scale():
    """300 k sequences / ~10^8 residues: sample vs oracle, idempotence, shard == slice, permutation
    invariance, planted self copy, score bounds."""
    q = cases.Q375
    n = 300_000
    res, off = swipe_amd.synth_db(1, n, query=q)
    M = swipe_amd.matrix_builtin("BLOSUM62")
    db = swipe_amd.Database.from_arrays(res, off)
    db.set_scoring(M, 11, 1)
    s1, c1 = db.search(q)
    s2, _ = db.search(q)
    assert np.array_equal(s1, s2)                                     # idempotent
    assert s1.min() >= 0 and (s1 <= 11 * np.minimum(375, np.diff(off))).all()
    rng = np.random.default_rng(7)
    pick = np.unique(np.concatenate([rng.integers(0, n, 3000), np.argsort(s1)[-200:], np.argsort(np.diff(off))[-50:]]))
    r2, o2 = oracle.pack([res[off[i]:off[i + 1]] for i in pick])
    want = oracle.search_all63(r2, o2, q, oracle.matrix_builtin("BLOSUM62"), 12, 1, threads=THREADS)
    assert np.array_equal(s1[pick], want)
    lo, hi = 100_000, 180_000                                           # a shard is a slice
    sh = swipe_amd.Database.from_arrays(res[off[lo]:off[hi]], off[lo:hi + 1] - off[lo], first_seqno=lo)
    sh.set_scoring(M, 11, 1)
    assert np.array_equal(sh.search(q)[0], s1[lo:hi])
    hits_sh = sh.search_topk(q, keep=50, minscore=45)[0]
    sh.close()
    order = sorted(((int(s1[i]), i) for i in range(lo, hi) if s1[i] >= 45), key=lambda t: (-t[0], -t[1]))[:50]
    assert hits_sh == [(i, s) for s, i in order]
    perm = rng.permutation(20_000)                                      # order of the database is irrelevant
    seqs = [res[off[i]:off[i + 1]] for i in perm]
    pd = swipe_amd.Database.from_sequences(seqs)
    pd.set_scoring(M, 11, 1)
    assert np.array_equal(pd.search(q)[0], s1[perm])
    pd.close()
    hits, tot, obv, _ = db.search_topk(q, keep=250, minscore=50)
    order = sorted(((int(s), i) for i, s in enumerate(s1) if s >= 50), key=lambda t: (-t[0], -t[1]))
    assert hits == [(i, s) for s, i in order[:250]] and tot == len(order) and obv == 0
    db.close()


def test_all_scores_of_a_100k_slice_of_the_bench_database():
    """BASELINE config 3: EVERY score of a 100 000-sequence slice of the bench's database (seed 1, planted homologs, the
    375-aa query) against the scalar 63-bit oracle - and every sequence the reference would escalate (score >= 117 to 16
    bits, search.cc thresholds via oracle.score_limits) is among them with its exact score, whatever width ran here (packed f16 up to ~2 000, 32 bits beyond)"""
    q = cases.Q375
    n = 100_000
    res, off = swipe_amd.synth_db(1, n, query=q)
    db = swipe_amd.Database.from_arrays(res, off)
    db.set_scoring(swipe_amd.matrix_builtin("BLOSUM62"), 11, 1)
    Mo = oracle.matrix_builtin("BLOSUM62")
    want = oracle.search_all63(res, off, q, Mo, 12, 1, threads=THREADS)
    got, c = db.search(q)
    assert np.array_equal(got, want)
    lim7 = oracle.score_limits(Mo)[2]                      # 117: matrices.cc:574-578
    assert lim7 == 117 and int((want >= lim7).sum()) > 0
    # the same through the top-K entry point with the bound build: list, totalhits, obvious
    hits, tot, obv, c2 = db.search_topk(q, keep=250, minscore=int(lim7))
    assert c2["narrow_shifted"] == 8
    assert (hits, tot, obv) == _expected_topk(want, 250, int(lim7))
    db.close()


def test_permissive_threshold_takes_the_histogram_path():
    """minscore 1 on 1.3 M short sequences: more candidates than the device compaction buffer holds, so the top-K floor
    comes from the host-side histogram; list, totalhits and the obvious count must still be exact"""
    q = cases.Q375[:60]
    n = 1_300_000
    rtab = synth.residue_table_protein()
    res = synth._random_residues(4, 1, 12 * n, rtab)
    off = np.arange(n + 1, dtype=np.int64) * 12
    db = swipe_amd.Database.from_arrays(res, off)
    db.set_scoring(swipe_amd.matrix_builtin("BLOSUM62"), 11, 1)
    s, _ = db.search(q)
    for keep, lo, hi in ((100, 1, 1 << 40), (2000, 1, 28), (50, 0, 1 << 40)):
        hits, tot, obv, _ = db.search_topk(q, keep=keep, minscore=lo, maxscore=hi)
        order = sorted(((int(v), i) for i, v in enumerate(s) if lo <= v <= hi), key=lambda t: (-t[0], -t[1]))
        assert hits == [(i, v) for v, i in order[:keep]]
        assert tot == int((s >= lo).sum()) and obv == int((s > hi).sum())
    db.close()


@pytest.mark.parametrize("qlen", [1, 100, 128, 129, 300, 384, 385, 512, 513, 1000, 1024, 1025, 3000])
def test_dual_query_kernel_both_strands(qlen):
    """search2: plus strand and reverse complement in the two halves of one pass (single and multi pass)"""
    rtab = synth.residue_table_nucleotide()
    q = synth._random_residues(31 + qlen, 1, qlen, rtab)
    qm = blastdb.revcomp_nt16(q)
    res, off = swipe_amd.synth_db(4, 1500, protein=False)
    mut = q.copy()
    mut[::11] = rtab[(np.arange(len(mut[::11])) * 613) % 4096]
    seqs = [res[off[i]:off[i + 1]] for i in range(1500)] + [q, qm, mut, blastdb.revcomp_nt16(mut)[: max(1, qlen // 2)],
                                                            np.zeros(0, np.uint8), q[:1]]
    r2, o2 = oracle.pack(seqs)
    db = swipe_amd.Database.from_arrays(r2, o2, symtype=0)
    db.set_scoring(swipe_amd.matrix_nucleotide(1, -3), 5, 2)
    Mo = oracle.matrix_nucleotide(1, -3)
    s1, s2, c = db.search2(q, qm)
    assert np.array_equal(s1, oracle.search_all63(r2, o2, q, Mo, 7, 2, threads=THREADS))
    assert np.array_equal(s2, oracle.search_all63(r2, o2, qm, Mo, 7, 2, threads=THREADS))
    if qlen >= 2048:
        assert c["wide"] > 0           # the self hits leave the f16 range and are re-queued
    hits, tot, obv, _ = db.search2_topk(q, qm, keep=20, minscore=15)
    want = sorted([(int(s), i, 0) for i, s in enumerate(s1) if s >= 15] + [(int(s), i, 1) for i, s in enumerate(s2) if s >= 15],
                  key=lambda t: (-t[0], -t[1], t[2]))[:20]
    assert hits == [(i, s, w) for s, i, w in want]
    db.close()


@pytest.mark.parametrize("lanes", [16, 8, 4, 2, 1])
@pytest.mark.parametrize("protein", [False, True])
def test_every_instantiation_of_the_single_pass_dual_kernel(protein, lanes, monkeypatch):
    """two queries of equal length in one pass, K = ceil(qlen / lanes): every K of the nucleotide build (1..63 with
    16-lane chains, 1..60 with 8 / 4, 1..32 with 2) and of the protein build (1..32), and the multi-pass kernel as cross-check"""
    monkeypatch.setenv("SWA_LANES", str(lanes))
    tab = synth.residue_table_protein() if protein else synth.residue_table_nucleotide()
    full = synth._random_residues(55, 1, 1024, tab)
    res, off = swipe_amd.synth_db(9, 250, protein=protein)
    seqs = [res[off[i]:off[i + 1]] for i in range(250)] + [full, full[200:900].copy(), np.zeros(0, np.uint8)]
    if not protein:
        seqs += [blastdb.revcomp_nt16(full[100:800])]
    r2, o2 = oracle.pack(seqs)
    db = swipe_amd.Database.from_arrays(r2, o2, symtype=1 if protein else 0)
    if protein:
        db.set_scoring(swipe_amd.matrix_builtin("BLOSUM62"), 11, 1)
        Mo, goe, ge = oracle.matrix_builtin("BLOSUM62"), 12, 1
    else:
        db.set_scoring(swipe_amd.matrix_nucleotide(1, -3), 5, 2)
        Mo, goe, ge = oracle.matrix_nucleotide(1, -3), 7, 2
    kmax = 32 if protein else {1: 48, 2: 32, 4: 60, 8: 60, 16: 63}[lanes]     # nucleotide: long lanes on 4 / 8 too (sw_dual_long*.hip)
    for K in range(1, kmax + 1):
        qlen = lanes * K - (K % lanes)
        q1 = full[:qlen]
        q2 = q1[::-1].copy() if protein else blastdb.revcomp_nt16(q1)
        s1, s2, c = db.search2(q1, q2)
        assert c["narrow_rows"] == K and c["narrow_shifted"] == (12 if lanes == 1 else 4)   # 12: one lane per sequence
        assert np.array_equal(s1, oracle.search_all63(r2, o2, q1, Mo, goe, ge, threads=THREADS)), K
        assert np.array_equal(s2, oracle.search_all63(r2, o2, q2, Mo, goe, ge, threads=THREADS)), K
    db.set_option("dual_mp", 1)
    t1, t2, c = db.search2(q1, q2)
    assert c["narrow_shifted"] == 1 and np.array_equal(t1, s1) and np.array_equal(t2, s2)
    db.close()


@pytest.mark.parametrize("two", [False, True])
def test_multipass_hand_over_buffer_is_bounded(two, monkeypatch):
    """a database whose longest sequences do not fit the common allotment of the pass hand-over buffer: those batches
    run first on fewer waves, the rest with the bounded allotment - same scores (SWA_BOUNDARY_MB shrinks the budget)"""
    monkeypatch.setenv("SWA_BOUNDARY_MB", "8")
    tab = synth.residue_table_nucleotide() if two else synth.residue_table_protein()
    q = synth._random_residues(61, 1, 1300, tab)
    res, off = swipe_amd.synth_db(11, 3000, protein=not two)
    seqs = [res[off[i]:off[i + 1]] for i in range(3000)]
    seqs += [synth._random_residues(62 + k, 1, n, tab) for k, n in enumerate([20000, 9000, 9001, 4000])] + [np.concatenate([seqs[0], q[100:1200], seqs[1]])]
    r2, o2 = oracle.pack(seqs)
    db = swipe_amd.Database.from_arrays(r2, o2, symtype=0 if two else 1)
    if two:
        db.set_scoring(swipe_amd.matrix_nucleotide(1, -3), 5, 2)
        Mo, goe, ge = oracle.matrix_nucleotide(1, -3), 7, 2
        q2 = blastdb.revcomp_nt16(q)
        s1, s2, c = db.search2(q, q2)
        assert np.array_equal(s2, oracle.search_all63(r2, o2, q2, Mo, goe, ge, threads=THREADS))
    else:
        db.set_scoring(swipe_amd.matrix_builtin("BLOSUM62"), 11, 1)
        Mo, goe, ge = oracle.matrix_builtin("BLOSUM62"), 12, 1
        s1, c = db.search(q)
    assert np.array_equal(s1, oracle.search_all63(r2, o2, q, Mo, goe, ge, threads=THREADS))
    db.close()


def test_every_pass_build_of_the_row_shifted_kernel(monkeypatch):
    """queries longer than 928 rows run as passes of 16 x K rows, K = 30..56, one launch per pass with the last row handed
    over through HBM: every K with two passes, then three and more passes; the planted sequences straddle the pass
    boundaries, overflow the packed range in the first, a middle or the last pass, or stay exact across all of them.
    SWA_BOUNDARY_MB = 1 also cuts the batches into several runs"""
    rtab = synth.residue_table_protein()
    full = synth._random_residues(123, 1, 5200, rtab)
    res, off = swipe_amd.synth_db(9, 500, query=full[:1200])
    base = [res[off[i]:off[i + 1]] for i in range(500)]
    Mo = oracle.matrix_builtin("BLOSUM62")
    lens = [32 * K for K in range(30, 57)] + [32 * K - 31 for K in (30, 41, 56)] + [929, 1793, 48 * 40, 2688, 2689, 5200]
    for n, qlen in enumerate(lens):
        monkeypatch.setenv("SWA_BOUNDARY_MB", "1" if n % 3 == 0 else "4096")
        q = full[:qlen]
        npass = (qlen + 895) // 896
        K = max(30, -(-qlen // (16 * npass)))
        edge = 16 * K
        planted = [q, q[edge - 120:edge + 130].copy(), q[:300].copy(), q[qlen - 320:].copy(), q[edge - 5:edge + 5].copy(),
                   q[edge * (npass - 1) - 200:edge * (npass - 1) + 60].copy(), np.concatenate([base[0], q[edge - 60:edge + 40], base[1]]),
                   q[100:qlen - 100:2].copy(), np.zeros(0, np.uint8)]
        r2, o2 = oracle.pack(base + planted)
        db = swipe_amd.Database.from_arrays(r2, o2)
        db.set_scoring(swipe_amd.matrix_builtin("BLOSUM62"), 11, 1)
        scores, c = db.search(q)
        want = oracle.search_all63(r2, o2, q, Mo, 12, 1, threads=THREADS)
        assert c["narrow_shifted"] == 5 and c["narrow_rows"] == K, (qlen, c)
        assert np.array_equal(scores, want), (qlen, np.nonzero(scores != want)[0][:10])
        assert c["wide"] >= 1
        db.close()


@pytest.mark.parametrize("K", ["16", "24", "32"])
def test_multipass_pair_kernel_rows_per_lane(monkeypatch, K):
    """every rows-per-lane build of the multi-pass pair kernel (SWA_MP_K override) on a long protein query; since the
    tuned kernel runs long queries in passes of its own this one only serves gap-extension penalties too large for 30+
    rows per lane, so it is forced here"""
    monkeypatch.setenv("SWA_FORCE_MP", "1")
    monkeypatch.setenv("SWA_MP_K", K)
    rtab = synth.residue_table_protein()
    q = synth._random_residues(77, 1, 1300, rtab)
    res, off = swipe_amd.synth_db(8, 2500, query=q)
    seqs = [res[off[i]:off[i + 1]] for i in range(2500)] + [q, q[200:900], np.zeros(0, np.uint8)]
    r2, o2 = oracle.pack(seqs)
    db = swipe_amd.Database.from_arrays(r2, o2)
    db.set_scoring(swipe_amd.matrix_builtin("BLOSUM62"), 11, 1)
    scores, c = db.search(q)
    assert np.array_equal(scores, oracle.search_all63(r2, o2, q, oracle.matrix_builtin("BLOSUM62"), 12, 1, threads=THREADS))
    assert c["wide"] > 0
    db.close()


@pytest.mark.parametrize("protein", [False, True])
def test_every_pass_build_of_the_dual_kernel(protein, monkeypatch):
    """two queries longer than one pass of the dual kernel (1008 rows for nucleotides, 512 otherwise): passes of 16 x K rows,
    K = 32..56 (nucleotide) / 17..32, one launch per pass; every K with two passes, then more passes; sequences that straddle
    the pass boundaries and ones that overflow in one query only; SWA_BOUNDARY_MB = 1 cuts the batches into several runs"""
    tab = synth.residue_table_protein() if protein else synth.residue_table_nucleotide()
    full = synth._random_residues(321, 1, 4200, tab)
    res, off = swipe_amd.synth_db(19, 300, protein=protein)
    base = [res[off[i]:off[i + 1]] for i in range(300)]
    if protein:
        Mo, goe, ge, kmax, kmin = oracle.matrix_builtin("BLOSUM62"), 12, 1, 32, 17
    else:
        Mo, goe, ge, kmax, kmin = oracle.matrix_nucleotide(1, -3), 7, 2, 56, 32
    lens = [32 * K for K in range(kmin, kmax + 1)] + [32 * kmax + 1, 48 * (kmax - 3), 64 * kmax + 7, 4200]
    for n, qlen in enumerate(lens):
        monkeypatch.setenv("SWA_BOUNDARY_MB", "1" if n % 3 == 0 else "4096")
        q1 = full[:qlen]
        q2 = q1[::-1].copy() if protein else blastdb.revcomp_nt16(q1)
        npass = -(-qlen // (16 * kmax))
        K = max(kmin, -(-qlen // (16 * npass)))
        edge = 16 * K
        planted = [q1, q1[edge - 150:edge + 150].copy(), q2[edge - 150:edge + 150].copy(), q1[:400].copy(), q2[qlen - 400:].copy(),
                   q1[edge * (npass - 1) - 90:edge * (npass - 1) + 90].copy(), np.concatenate([base[0], q2[edge - 70:edge + 50], base[1]]),
                   q1[50:qlen - 50:2].copy(), np.zeros(0, np.uint8)]
        r2, o2 = oracle.pack(base + planted)
        db = swipe_amd.Database.from_arrays(r2, o2, symtype=1 if protein else 0)
        if protein:
            db.set_scoring(swipe_amd.matrix_builtin("BLOSUM62"), 11, 1)
        else:
            db.set_scoring(swipe_amd.matrix_nucleotide(1, -3), 5, 2)
        s1, s2, c = db.search2(q1, q2)
        assert c["narrow_shifted"] == 6 and c["narrow_rows"] == K, (qlen, c)
        w1 = oracle.search_all63(r2, o2, q1, Mo, goe, ge, threads=THREADS)
        w2 = oracle.search_all63(r2, o2, q2, Mo, goe, ge, threads=THREADS)
        assert np.array_equal(s1, w1), (qlen, np.nonzero(s1 != w1)[0][:10])
        assert np.array_equal(s2, w2), (qlen, np.nonzero(s2 != w2)[0][:10])
        db.close()


def test_dual_query_protein_and_custom_matrix():
    """the dual kernel is not nucleotide specific: two protein queries of equal length"""
    q1 = cases.Q375
    q2 = cases.Q375[::-1].copy()
    res, off = swipe_amd.synth_db(6, 3000, query=q1)
    db = swipe_amd.Database.from_arrays(res, off)
    db.set_scoring(swipe_amd.matrix_builtin("BLOSUM62"), 11, 1)
    s1, s2, _ = db.search2(q1, q2)
    Mo = oracle.matrix_builtin("BLOSUM62")
    assert np.array_equal(s1, oracle.search_all63(res, off, q1, Mo, 12, 1, threads=THREADS))
    assert np.array_equal(s2, oracle.search_all63(res, off, q2, Mo, 12, 1, threads=THREADS))
    assert np.array_equal(db.search(q1)[0], s1)
    db.close()


def test_nucleotide_scale_both_strands():
    rtab = synth.residue_table_nucleotide()
    q = synth._random_residues(99, 1, 1000, rtab)
    res, off = swipe_amd.synth_db(3, 20_000, protein=False)
    seqs = [res[off[i]:off[i + 1]] for i in range(20_000)] + [q[200:800], blastdb.revcomp_nt16(q[100:900])]
    r2, o2 = oracle.pack(seqs)
    db = swipe_amd.Database.from_arrays(r2, o2, symtype=0)
    db.set_scoring(swipe_amd.matrix_nucleotide(1, -3), 5, 2)
    Mo = oracle.matrix_nucleotide(1, -3)
    for qs in (q, blastdb.revcomp_nt16(q)):
        scores, _ = db.search(qs)
        want = oracle.search_all63(r2, o2, qs, Mo, 7, 2, threads=THREADS)
        assert np.array_equal(scores, want)
    db.close()


@pytest.mark.parametrize("name", ["p1k", "multivol", "nt", "asym", "edges"])
def test_cli_output_equals_reference_cli(tmp_path, name):
    """swipe_amd_cli -m 7 / -m 0 against the reference's own output for the same command line
    (XML identical except <len>, which the reference leaves unset without alignments; the plain hit
    list - names, bit scores, E-values, strands - identical line by line)."""
    import re
    import subprocess
    from conftest import ROOT
    case, g = cases.get(name), load_golden(name)
    base = str(tmp_path / name)
    blastdb.write_db(base, case.seqs, protein=case.protein, volumes=case.volumes)
    alpha = blastdb.NCBISTDAA if case.protein else blastdb.NCBI4NA
    qf = str(tmp_path / "q.fa")
    open(qf, "w").write(">query test\n" + "".join(alpha[c] for c in case.query) + "\n")
    exe = os.path.join(ROOT, "swipe_amd", "swipe_amd_cli")
    args = [exe, "-d", base, "-i", qf, "-p", "1" if case.protein else "0", "-G", str(case.gapopen), "-E", str(case.gapextend),
            "-v", str(case.keep), "-e", "10", "-b", "0"]
    if case.protein:
        mat = case.matrix
        if mat == "@text":
            mat = str(tmp_path / "matrix.txt")
            open(mat, "w").write(case.matrix_text)
        args += ["-M", mat]
    else:
        args += ["-r", str(case.match), "-q", str(case.mismatch)]
    xml = subprocess.run(args + ["-m", "7"], capture_output=True, text=True, check=True).stdout
    strip = lambda t: re.sub(r"\s*<len>\d+</len>", "", t)
    assert strip(xml) == strip(g["xml"])
    plain = subprocess.run(args + ["-m", "0"], capture_output=True, text=True, check=True).stdout.splitlines()
    k = next(i for i, l in enumerate(plain) if l.startswith("Sequences producing"))
    assert [l for l in plain[k + 2:] if l.strip()] == g["plain_hits"]


@pytest.mark.parametrize("name", ["p1k", "multivol", "nt", "asym", "edges", "limit16"])
def test_cli_alignment_output_equals_reference_cli(tmp_path, name):
    """-b > 0: the alignment phase end to end.  -m 7 (edit scripts, coordinates, the three alignment
    lines), -m 8 / -m 9 (identity, length, mismatches, gap openings, coordinates, E-value, bits) and
    the -m 0 pairwise blocks must equal the reference's output byte for byte."""
    import subprocess
    from conftest import ROOT
    case, g = cases.get(name), load_golden(name)
    base = str(tmp_path / name)
    blastdb.write_db(base, case.seqs, protein=case.protein, volumes=case.volumes)
    alpha = blastdb.NCBISTDAA if case.protein else blastdb.NCBI4NA
    qf = str(tmp_path / "q.fa")
    open(qf, "w").write(">query test\n" + "".join(alpha[c] for c in case.query) + "\n")
    exe = os.path.join(ROOT, "swipe_amd", "swipe_amd_cli")
    args = [exe, "-d", base, "-i", qf, "-p", "1" if case.protein else "0", "-G", str(case.gapopen), "-E", str(case.gapextend),
            "-v", str(case.keep), "-e", "10"]
    if case.protein:
        mat = case.matrix
        if mat == "@text":
            mat = str(tmp_path / "matrix.txt")
            open(mat, "w").write(case.matrix_text)
        args += ["-M", mat]
    else:
        args += ["-r", str(case.match), "-q", str(case.mismatch)]
    run = lambda extra: subprocess.run(args + extra, capture_output=True, text=True, check=True).stdout
    assert run(["-m", "7", "-b", str(g["nalign"])]) == g["xml_align"]
    assert run(["-m", "8", "-b", str(case.keep)]) == g["tsv"]
    # the same with the bound build of the first pass forced wherever a build exists for the query
    forced = lambda extra: subprocess.run(args + extra, capture_output=True, text=True, check=True,
                                          env=dict(os.environ, SWA_BOUND="1")).stdout
    assert forced(["-m", "7", "-b", str(g["nalign"])]) == g["xml_align"]
    assert forced(["-m", "8", "-b", str(case.keep)]) == g["tsv"]
    t9 = run(["-m", "9", "-b", str(case.keep)]).split("\n")[1:]
    want = g["tsv9"].split("\n")
    assert t9[0] == want[0] and t9[1].startswith("# Database: ") and t9[2:] == want[2:]
    plain = run(["-m", "0", "-b", str(g["nalign"])])
    assert plain[plain.index("Sequences producing"):] == g["plain_align"]


@pytest.mark.parametrize("name", NAMES)
def test_alignment_phase_matches_reference(name):
    """swa_align_hits (GPU end points + host traceback) for EVERY positive-scoring (sequence, strand) of the
    case against the reference's align() as hits_align runs it (with the search16s hint whenever
    hits.cc:587 honours it)."""
    case, g = cases.get(name), load_golden(name)
    db = open_case(case)
    rows = g["align"]
    got = db.align(case.query, [r[0] for r in rows], [r[1] for r in rows])
    assert len(got) == len(rows)
    for a, (seqno, ds, s16s, bp, bq, score, qs, dst, qe, de, cigar, hinted) in zip(got, rows):
        want = hinted if hinted is not None else [score, qs, dst, qe, de, cigar]
        assert [a["score"], a["q_start"], a["d_start"], a["q_end"], a["d_end"], a["cigar"]] == want, (seqno, ds)
        assert a["hinted"] == (hinted is not None) and a["seqno"] == seqno and a["dstrand"] == ds
        d = blastdb.revcomp_nt16(case.seqs[seqno]) if ds else case.seqs[seqno]
        assert np.array_equal(db.sequence(seqno, ds), d)
    e = db.search_endpoints(case.query, [r[0] for r in rows], [r[1] for r in rows])
    for k, r in enumerate(rows):
        if r[2] < g["scorelimit16"]:
            assert (int(e[0][k]), int(e[1][k]), int(e[2][k])) == (r[2], r[3], r[4])
    db.close()


TNAMES = [f.__name__[5:] for f in cases.TRANSLATED]


def open_translated_case(case):
    if case.sym in (3, 4):
        res, off = oracle.pack(case.seqs)
        db = swipe_amd.Database.from_arrays(res, off, translate_gencode=case.db_gencode)
    else:
        db = swipe_amd.Database.from_sequences(case.seqs, symtype=1)
    db.set_scoring(case_matrix(case, swipe_amd), case.gapopen, case.gapextend)
    return db


def query_frames(case):
    if case.sym in (2, 4):
        t = swipe_amd.translate_table(case.query_gencode)
        return [swipe_amd.translate(case.query, k // 3, k % 3, t) for k in range(6)]
    return [case.query]


@pytest.mark.parametrize("name", TNAMES)
def test_translated_scores_equal_reference(name):
    """-p 2/3/4: the GPU's six-frame translation of the database (swa_translate_frames) and the DP over every
    (query frame, database frame) pair against the reference's own 63-bit scores for all of them."""
    case, g = cases.get(name), load_golden(name)
    db = open_translated_case(case)
    info = db.info()
    nd = 6 if case.sym in (3, 4) else 1
    assert info["frames"] == nd and info["seqcount"] == len(case.seqs)
    assert info["symcount"] == sum(len(s) for s in case.seqs)
    dt = oracle.translate_table(case.db_gencode)
    for seqno in range(len(case.seqs)):
        for tag in range(nd):
            want = oracle.translate(case.seqs[seqno], tag // 3, tag % 3, dt) if nd == 6 else case.seqs[seqno]
            assert np.array_equal(db.sequence(seqno, tag // 3, tag % 3), want), (seqno, tag)
    for qt, q in enumerate(query_frames(case)):
        scores, c = db.search(q)
        want = np.zeros(len(case.seqs) * nd, dtype=np.int64)
        for r in g["raw"]:
            if r[1] == qt:
                want[r[0] * nd + r[2]] = r[8]
        assert np.array_equal(scores, want), qt
    db.close()


def test_six_frame_translation_at_awkward_shapes():
    """swa_translate_frames: thousands of empty / 1..5-base sequences in a row (more sequence boundaries per tile than
    the LDS window holds -> global-search fallback), one sequence spanning hundreds of tiles, ambiguity codes"""
    rng = np.random.default_rng(3)
    seqs = [rng.integers(1, 16, int(n)).astype(np.uint8) for n in rng.integers(0, 6, 9000)]
    seqs += [rng.integers(1, 16, 700_001).astype(np.uint8)]
    seqs += [rng.integers(1, 16, int(n)).astype(np.uint8) for n in rng.integers(0, 400, 300)]
    res, off = oracle.pack(seqs)
    db = swipe_amd.Database.from_arrays(res, off, translate_gencode=11)
    t = oracle.translate_table(11)
    for s in list(range(0, 9000, 97)) + [8998, 8999, 9000, 9001, 9150, len(seqs) - 1]:
        for tag in range(6):
            assert np.array_equal(db.sequence(s, tag // 3, tag % 3), oracle.translate(seqs[s], tag // 3, tag % 3, t)), (s, tag)
    db.close()


@pytest.mark.parametrize("name", TNAMES)
def test_translated_hit_list_and_alignments_equal_reference(name):
    case, g = cases.get(name), load_golden(name)
    db = open_translated_case(case)
    nsym = int(sum(len(s) for s in case.seqs))
    st = swipe_amd.stats_init(symtype=case.sym, matrix=case.matrix, gapopen=case.gapopen, gapextend=case.gapextend,
                              qlen=len(case.query), db_seqcount=len(case.seqs), db_symcount=nsym)
    qf = query_frames(case)
    per_seq = {2: 6, 3: 6, 4: 36}[case.sym]
    hits, tot, obv, c = db.search_frames_topk(qf, keep=min(case.keep, per_seq * len(case.seqs)),
                                              minscore=st.scorethreshold, maxscore=st.upperscorethreshold)
    cli = g["cli"]["1"]
    assert [h[0] for h in hits] == cli["seqno"] and [h[1] for h in hits] == cli["score"]
    lab = lambda s, f: "%s%d" % ("-" if s else "+", f + 1)
    got = [lab(h[2], h[3]) if case.sym == 2 else lab(h[4], h[5]) if case.sym == 3 else lab(h[2], h[3]) + "/" + lab(h[4], h[5])
           for h in hits]
    assert got == cli["strand"]
    # alignment phase for every positive pair of the fixture, grouped by query frame as align_chunk does
    for qt, q in enumerate(qf):
        rows = [r for r in g["align"] if r[1] == qt]
        if not rows:
            continue
        al = db.align(q, [r[0] for r in rows], [r[2] // 3 for r in rows], [r[2] % 3 for r in rows])
        for a, (seqno, _, dtag, s16s, bp, bq, score, qs, dst, qe, de, cigar, hinted) in zip(al, rows):
            want = hinted if hinted is not None else [score, qs, dst, qe, de, cigar]
            assert [a["score"], a["q_start"], a["d_start"], a["q_end"], a["d_end"], a["cigar"]] == want, (seqno, qt, dtag)
            assert a["dlennt"] == (len(case.seqs[seqno]) if case.sym in (3, 4) else 0)
    db.close()


@pytest.mark.parametrize("name", TNAMES)
def test_translated_cli_output_equals_reference_cli(tmp_path, name):
    import subprocess
    from conftest import ROOT
    case, g = cases.get(name), load_golden(name)
    base = str(tmp_path / name)
    blastdb.write_db(base, case.seqs, protein=case.protein, volumes=case.volumes)
    alpha = blastdb.NCBI4NA if case.query_is_nt else blastdb.NCBISTDAA
    qf = str(tmp_path / "q.fa")
    open(qf, "w").write(">query test\n" + "".join(alpha[c] for c in case.query) + "\n")
    exe = os.path.join(ROOT, "swipe_amd", "swipe_amd_cli")
    args = [exe, "-d", base, "-i", qf, "-p", str(case.sym), "-G", str(case.gapopen), "-E", str(case.gapextend),
            "-v", str(case.keep), "-e", "10", "-M", case.matrix, "-Q", str(case.query_gencode), "-D", str(case.db_gencode)]
    run = lambda extra: subprocess.run(args + extra, capture_output=True, text=True, check=True).stdout
    assert run(["-m", "7", "-b", str(g["nalign"])]) == g["xml_align"]
    assert run(["-m", "8", "-b", str(case.keep)]) == g["tsv"]
    plain = run(["-m", "0", "-b", str(g["nalign"])])
    assert plain[plain.index("Sequences producing"):] == g["plain_align"]


@pytest.mark.parametrize("variant", ["plain", "plain_gis", "plain_taxid", "plain_gis_taxid", "masked", "masked_gis_taxid",
                                     "taxlist", "taxlist_gis_taxid", "masked_taxlist"])
def test_cli_real_database_features_equal_reference_cli(tmp_path, variant):
    """an OID-mask alias (searches only members, statistics on the alias's NSEQ/LENGTH), a
    -x taxid list, -I / -H rendering of every Seq-id flavour and merged definition lines - output of
    -m 0 / 7 / 8 byte for byte against the reference CLI."""
    import subprocess
    from conftest import ROOT
    from test_host_cpu import build_headers_db, HEADER_VARIANTS
    case, vol, masked, tx = build_headers_db(tmp_path)
    ref = load_golden("headers")["variants"][variant]
    dbn, flags, taxlist = HEADER_VARIANTS[variant]
    qf = str(tmp_path / "q.fa")
    open(qf, "w").write(">query test\n" + "".join(blastdb.NCBISTDAA[c] for c in case.query) + "\n")
    exe = os.path.join(ROOT, "swipe_amd", "swipe_amd_cli")
    args = [exe, "-d", vol if dbn == "vol" else masked, "-i", qf, "-v", str(case.keep), "-e", "1e6"]
    args += (["-I"] if flags & 1 else []) + (["-H"] if flags & 2 else []) + (["-x", tx] if taxlist else [])
    run = lambda extra: subprocess.run(args + extra, capture_output=True, text=True, check=True).stdout
    assert run(["-m", "7", "-b", "5"]) == ref["m7"]
    assert run(["-m", "8", "-b", str(case.keep)]) == ref["m8"]
    plain = run(["-m", "0", "-b", "5"])
    assert plain[plain.index("Sequences producing"):] == ref["m0"]
    for line in ref["db_lines"]:
        assert line in plain.splitlines()


def test_inclusion_subset_scores(tmp_path):
    """swa_db_set_inclusion: excluded sequences are skipped (score -1), the rest score as before, and the
    subset can be changed and lifted again."""
    case = cases.get("p1k")
    db = open_case(case)
    full, _ = db.search(case.query)
    inc = (np.arange(len(case.seqs)) % 3 != 1).astype(np.uint8)
    db.set_inclusion(inc)
    part, c = db.search(case.query)
    assert np.array_equal(part[inc == 1], full[inc == 1]) and np.all(part[inc == 0] == -1)
    assert c["cells"] == len(case.query) * sum(len(s) for s, k in zip(case.seqs, inc) if k)
    hits, tot, obv, _ = db.search_topk(case.query, keep=50, minscore=30)
    want = sorted([(int(full[i]), i) for i in range(len(full)) if inc[i] and full[i] >= 30], reverse=True)[:50]
    assert [(h[1], h[0]) for h in hits] == want and tot == sum(1 for i in range(len(full)) if inc[i] and full[i] >= 30)
    db.set_inclusion(np.zeros(len(case.seqs), np.uint8))
    none, _ = db.search(case.query)
    assert np.all(none == -1)
    db.set_inclusion(None)
    again, _ = db.search(case.query)
    assert np.array_equal(again, full)
    db.close()


@pytest.mark.parametrize("qlen", [1, 63, 64, 257, 1025, 2048, 2049, 5000])
def test_endpoints_wave_kernel_all_row_counts_and_passes(qlen, monkeypatch):
    """search16s semantics from the wave-per-sequence kernel for every rows-per-lane instantiation, across the
    64*K-row pass boundary (2048 / 2049 / 5000 rows = 1 / 2 / 3 passes), and from the one-thread 64-bit form."""
    rtab = synth.residue_table_protein()
    q = synth._random_residues(777, 1, qlen, rtab)
    seqs = [synth._random_residues(800 + k, 2, n, rtab) for k, n in enumerate([0, 1, 2, 17, 64, 65, 127, 128, 129, 300, 1000, 2500])]
    seqs += [q.copy(), q[: max(1, qlen // 2)].copy(), q[qlen // 3:].copy(), np.concatenate([seqs[9], q[: min(qlen, 600)], seqs[6]])]
    M = oracle.matrix_builtin("BLOSUM62")
    want = [oracle.search16s_lane(d, q, M, 12, 1) for d in seqs]
    assert max(w[0] for w in want) < 65525
    db = swipe_amd.Database.from_sequences(seqs, symtype=1)
    db.set_scoring(swipe_amd.matrix_builtin("BLOSUM62"), 11, 1)
    ids = list(range(len(seqs)))
    for mode in ("wave", "thread"):
        db.set_option("endpoints_thread", mode)
        e = db.search_endpoints(q, ids)
        assert [(int(e[0][k]), int(e[1][k]), int(e[2][k])) for k in ids] == want, mode
    db.close()


def test_cli_multi_query_file_equals_reference_cli(tmp_path):
    """query_read (query.cc:265-366) through the CLI: several queries per file - no description on the first,
    wrapped and lower-case lines, blank lines, digits / '*' / CR LF inside, an empty query - and the
    per-query output loop; -m 7 / 8 / 9 / 0 byte for byte."""
    import subprocess
    from conftest import ROOT
    g = load_golden("multiquery")
    case = cases.get("edges")
    assert g["checksum"] == case.checksum()
    base = str(tmp_path / "db")
    blastdb.write_db(base, case.seqs, protein=True)
    qf = str(tmp_path / "q.fa")
    open(qf, "w").write(g["query_text"])
    exe = os.path.join(ROOT, "swipe_amd", "swipe_amd_cli")
    for m, b in (("7", "3"), ("8", "10"), ("9", "10"), ("0", "2")):
        r = subprocess.run([exe, "-d", base, "-i", qf, "-m", m, "-b", b, "-v", "12", "-e", "1000"], capture_output=True, text=True)
        assert r.returncode == g["rc" + m], r.stderr
        if m == "8":
            assert r.stdout == g["m8"]
        elif m == "7":
            # <len> of a hit WITHOUT an alignment is uninitialised memory in the reference (hits.cc:566 only fills
            # dlen for aligned hits; the list is re-malloc'ed per query, so later queries show stale values there)
            def norm(t):
                hits = t.split("<hit>")
                return "<hit>".join(h if "<alignment>" in h else re.sub(r"<len>\d+</len>", "<len>?</len>", h) for h in hits)
            import re
            assert norm(r.stdout) == norm(g["m7"])
        elif m == "9":        # first comment line of every query block names the program; the database path differs
            strip = lambda t: [l for l in t.split("\n") if not l.startswith("# SWIPE") and not l.startswith("# swipe_amd") and not l.startswith("# Database:")]
            assert strip(r.stdout) == strip(g["m9"])
        else:                 # the whole report from the first parameter block on, minus paths, times and speeds
            def body(t):
                lines = t.split("\n")
                k = next(i for i, l in enumerate(lines) if l.startswith("Database file:"))
                skip = ("Database file:", "Query file name:", "Search started:", "Search completed:", "Elapsed:", "Speed:")
                return [l for l in lines[k:] if not l.startswith(skip)]
            assert body(r.stdout) == body(g["m0"])


def test_cli_errors_like_the_reference(tmp_path):
    import subprocess
    from conftest import ROOT
    exe = os.path.join(ROOT, "swipe_amd", "swipe_amd_cli")
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 1 and "No database specified." in r.stderr
    r = subprocess.run([exe, "-d", str(tmp_path / "nosuch"), "-i", "/dev/null"], capture_output=True, text=True)
    assert r.returncode == 1 and "Unable to open file" in r.stderr


# ------------------------------------------------------------------------------------------------ round 2
def test_seeded_fuzz_slice():
    """a bounded, seeded slice of tools/gpu_fuzz.py (random scoring systems, alphabets, query lengths, one / two queries,
    top-K searches with the automatic first pass and the bound build forced, inclusion subsets) - every score, hit list
    and count against the oracle"""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import gpu_fuzz
    rng = np.random.default_rng(20260929)
    bad = []
    for it in range(30):
        desc, ok, ok2, ok4, ok3 = gpu_fuzz.one_config(rng, THREADS, max_qlen=1600, max_nseq=1500)
        if not (ok and ok2 and ok4 and ok3):
            bad.append((it, desc, ok, ok2, ok4, ok3))
    assert not bad, bad


def test_options_are_explicit_and_checked():
    """swa_set_option: unknown keys and unparsable values are errors; NULL restores the default; the search path does not
    read the environment (a variable set after the handle exists changes nothing)"""
    q = cases.Q375
    res, off = swipe_amd.synth_db(1, 3000, query=q)
    db = swipe_amd.Database.from_arrays(res, off)
    db.set_scoring(swipe_amd.matrix_builtin("BLOSUM62"), 11, 1)
    with pytest.raises(swipe_amd.SwaError):
        db.set_option("no_such_knob", 1)
    with pytest.raises(swipe_amd.SwaError):
        db.set_option("lanes", "eight")
    _, c = db.search(q)
    assert c["narrow_shifted"] == 2 and c["narrow_rows"] == 47          # 375 rows: 8-lane chains by default
    os.environ["SWA_LANES"] = "16"
    try:
        _, c = db.search(q)
        assert c["narrow_shifted"] == 2                                  # the handle already exists: no effect
        db2 = swipe_amd.Database.from_arrays(res, off)                   # read once, at creation
        db2.set_scoring(swipe_amd.matrix_builtin("BLOSUM62"), 11, 1)
        assert db2.search(q)[1]["narrow_shifted"] == 1
        db2.close()
    finally:
        del os.environ["SWA_LANES"]
    db.set_option("lanes", 16)
    assert db.search(q)[1]["narrow_shifted"] == 1
    db.set_option("lanes", None)
    assert db.search(q)[1]["narrow_shifted"] == 2
    db.close()


@pytest.mark.parametrize("host,follow", [(0, 1), (0, 0), (1, 0)])
def test_requeue_list_longer_than_the_device_driven_kernel_takes(host, follow):
    """the re-queue list normally never reaches the host: a persistent grid of waves works it off up to 65 536 entries
    and the search synchronises once.  A list beyond that (here: the bound build forced with a threshold of 1 - every
    sequence comes back) is taken over by the host after that synchronisation; requeue_host = 1 is the old
    host-driven path.  Same hits either way"""
    q = cases.Q375[:96]
    res, off = swipe_amd.synth_db(21, 70_000, query=q)
    lens = np.minimum(np.diff(off), 40)                      # sequences cut to at most 40 residues: a quick first pass
    o2 = np.zeros(len(lens) + 1, dtype=np.int64)
    np.cumsum(lens, out=o2[1:])
    src = np.repeat(off[:-1] - o2[:-1], lens) + np.arange(int(o2[-1]), dtype=np.int64)
    r2 = res[src]
    db = swipe_amd.Database.from_arrays(r2, o2)
    db.set_scoring(swipe_amd.matrix_builtin("BLOSUM62"), 11, 1)
    want = oracle.search_all63(r2, o2, q, oracle.matrix_builtin("BLOSUM62"), 12, 1, threads=THREADS)
    db.set_option("bound", 1)
    db.set_option("requeue_host", host)
    db.set_option("requeue_follow", follow)
    hits, tot, obv, c = db.search_topk(q, keep=100, minscore=1)
    assert c["narrow_shifted"] == 8 and c["wide"] >= 70_000 - 5
    assert (hits, tot, obv) == _expected_topk(want, 100, 1)
    db.close()


def test_second_query_takes_the_64_bit_hop():
    """two queries whose scores both leave 32 bits: each keeps its own 64-bit score array (the first version answered
    SWA_ERANGE for the second query)"""
    rtab = synth.residue_table_protein()
    q1 = synth._random_residues(77, 1, 40, rtab)
    q2 = q1[::-1].copy()
    res, off = swipe_amd.synth_db(5, 300)
    seqs = [res[off[i]:off[i + 1]] for i in range(300)] + [q1, q2, np.concatenate([q2, q1])]
    r2, o2 = oracle.pack(seqs)
    M = np.full(1024, -1, dtype=np.int64)
    for a in range(1, 28):
        M[(a << 5) | a] = 0x3fffffff // 4
    db = swipe_amd.Database.from_arrays(r2, o2)
    db.set_scoring(M, 11, 1)
    s1, s2, c = db.search2(q1, q2)
    assert np.array_equal(s1, oracle.search_all63(r2, o2, q1, M, 12, 1, threads=THREADS))
    assert np.array_equal(s2, oracle.search_all63(r2, o2, q2, M, 12, 1, threads=THREADS))
    assert c["full"] >= 4 and int(s2.max()) > (1 << 31)
    hits, tot, obv, _ = db.search2_topk(q1, q2, keep=5, minscore=1 << 31)
    want = sorted([(int(v), i, 0) for i, v in enumerate(s1) if v >= (1 << 31)] + [(int(v), i, 1) for i, v in enumerate(s2) if v >= (1 << 31)],
                  key=lambda t: (-t[0], -t[1], t[2]))[:5]
    assert hits == [(i, v, w) for v, i, w in want] and tot == len([1 for v in list(s1) + list(s2) if v >= (1 << 31)])
    db.close()


@pytest.mark.parametrize("lanes", [4, 8])
def test_overflow_to_infinity_does_not_poison_the_neighbour(lanes):
    """short chains isolate neighbouring sequences with a multiplication by zero; a self-hit under a matrix
    with scores of 400 drives its f16 state past 65504 = inf, and 0 x inf = NaN would zero the NEIGHBOUR's score without
    re-queueing it.  Such searches take 16-lane chains (zero fill by DPP); every score must be exact"""
    rtab = synth.residue_table_protein()
    q = synth._random_residues(5, 1, 180, rtab)      # 180 x 400 = 72 000 > 65 504 (2-lane chains end at 96 rows: out of reach)
    res, off = swipe_amd.synth_db(8, 64)
    base = [res[off[i]:off[i + 1]] for i in range(64)]
    seqs = []
    for k in range(32):                                      # self-hits interleaved with random neighbours of the same length
        seqs += [q.copy(), synth._random_residues(1000 + k, 1, len(q), rtab)]
    seqs += base
    r2, o2 = oracle.pack(seqs)
    M = swipe_amd.matrix_builtin("BLOSUM62").copy()
    for a in range(1, 28):
        M[(a << 5) | a] = 400
    db = swipe_amd.Database.from_arrays(r2, o2)
    db.set_option("lanes", lanes)
    db.set_scoring(M, 11, 1)
    scores, c = db.search(q)
    want = oracle.search_all63(r2, o2, q, M, 12, 1, threads=THREADS)
    assert int(want.max()) > 65504 and np.array_equal(scores, want), c
    db.close()


def test_database_residue_codes_are_validated():
    res, off = swipe_amd.synth_db(1, 50)
    bad = res.copy()
    bad[100] = 40
    with pytest.raises(swipe_amd.SwaError) as e:
        swipe_amd.Database.from_arrays(bad, off)
    assert "residue code" in str(e.value)
    nt, noff = swipe_amd.synth_db(3, 50, protein=False)
    nt = nt.copy()
    nt[7] = 17
    with pytest.raises(swipe_amd.SwaError):
        swipe_amd.Database.from_arrays(nt, noff, symtype=0)


def _bench_line(env_extra, *args):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, **env_extra)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), *args], env=env, capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, out.stderr[-2000:]
    return json.loads(out.stdout.strip().splitlines()[-1])


def test_bench_step_over_rccl_with_one_rank():
    """the N > 1 code path of bench.py on the one GPU there is: SWA_BENCH_FORCE_DIST=1 initialises the nccl (RCCL) process
    group with world size 1, every step goes through gather_topk_array (all_gather_into_tensor on device + merge), and the
    gathered hit list equals the local one; both runs verify their scores against the oracle inside bench.py"""
    args = ("--nseq", "200000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-secondary", "--verify-sample", "2000")
    local = _bench_line({}, *args)
    dist = _bench_line({"SWA_BENCH_FORCE_DIST": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29517", "RANK": "0",
                        "WORLD_SIZE": "1", "LOCAL_RANK": "0"}, *args)
    assert local["search"] == dist["search"] and local["hits_sha1"] == dist["hits_sha1"]
    assert dist["verified_vs_oracle"] >= 2000 and dist["scaling"] == "strong" and dist["n_gpus"] == 1
    assert "roofline" in dist and "overhead_ms" in dist


def _long_subject_db(q, rng, rtab, nbase, long_len, protein=True, seed=31):
    """random sequences + one very long one that carries copies of the query: one whole, one with a long insertion in
    the middle (an alignment that spans a long gap), pieces at both ends"""
    res, off = swipe_amd.synth_db(seed, nbase, protein=protein)
    seqs = [res[off[i]:off[i + 1]] for i in range(nbase)]
    body = rtab[rng.integers(0, len(rtab), long_len)].astype(np.uint8)
    half = len(q) // 2
    ins = rtab[rng.integers(0, len(rtab), 150)].astype(np.uint8)
    pos = [long_len // 7, long_len // 2, long_len - 3 * len(q)]
    body[pos[0]:pos[0] + len(q)] = q
    gapped = np.concatenate([q[:half], ins, q[half:]])
    body[pos[1]:pos[1] + len(gapped)] = gapped
    body[:half] = q[half:2 * half]
    body[long_len - half:] = q[:half]
    seqs.append(body)
    seqs.append(np.zeros(0, np.uint8))
    return seqs


def test_long_subject_is_searched_as_overlapping_windows():
    """a 35 000-residue protein among 2 000 ordinary ones: cut into windows that overlap by the longest span a
    positive-scoring alignment can have, each window searched as a sequence of its own, maximum per parent - every score
    equals the oracle's, for all-scores searches, top-K searches with the bound build, both queries of a pair, and the
    long sequence no longer sets the time of the pass"""
    rng = np.random.default_rng(77)
    rtab = synth.residue_table_protein()
    q = cases.Q375
    seqs = _long_subject_db(q, rng, rtab, 2000, 35_000)
    r2, o2 = oracle.pack(seqs)
    Mo = oracle.matrix_builtin("BLOSUM62")
    want = oracle.search_all63(r2, o2, q, Mo, 12, 1, threads=THREADS)
    db = swipe_amd.Database.from_arrays(r2, o2)
    db.set_scoring(swipe_amd.matrix_builtin("BLOSUM62"), 11, 1)
    db.set_option("window", 0)
    s0, c0 = db.search(q)
    db.set_option("window", None)
    s1, c1 = db.search(q)
    assert np.array_equal(s0, want) and np.array_equal(s1, want)
    assert int(want[2000]) >= 1957                                  # the whole copy of the query sits in it
    if not under_interpreter():                            # (tools/gfx950sim has no clock worth asserting on)
        assert c1["kernel_ms"] < 0.5 * c0["kernel_ms"], (c0["kernel_ms"], c1["kernel_ms"])   # 35 000 columns on one chain vs windows
    for bound in (0, 1):
        db.set_option("bound", bound)
        for minscore in (40, 80, 300):
            hits, tot, obv, c = db.search_topk(q, keep=50, minscore=minscore)
            assert (hits, tot, obv) == _expected_topk(want, 50, minscore), (bound, minscore)
    db.set_option("bound", None)
    q2 = q[::-1].copy()
    want2 = oracle.search_all63(r2, o2, q2, Mo, 12, 1, threads=THREADS)
    t1, t2, _ = db.search2(q, q2)
    assert np.array_equal(t1, want) and np.array_equal(t2, want2)
    # a longer query: passes of the kernel over the view (hand-over buffer addressed per region)
    ql = np.concatenate([q, synth._random_residues(3, 1, 700, rtab)])
    wantl = oracle.search_all63(r2, o2, ql, Mo, 12, 1, threads=THREADS)
    sl, cl = db.search(ql)
    assert cl["narrow_shifted"] == 5 and np.array_equal(sl, wantl)
    db.close()


def test_multipass_kernel_over_a_view_whose_windows_are_shorter_than_its_other_batches():
    """the block-synchronous multi-pass kernel sizes its per-wave pass hand-over by the longest batch of the set
    it runs over.  Over a window view that is NOT batch 0: the view is [window batches | the set's remaining batches], and
    with a large gap extension penalty the windows (W + O = 1 501 columns here) are far shorter than the longest sequences
    left whole (4 500).  Eight long sequences = exactly one batch of the pair stream, so no ordinary sequence shares the
    window batches either.  force_mp with 3 passes of a 600-row query; the query planted beyond column 1 600 of the
    4 500-residue sequences, where an undersized hand-over used to be overrun: every score against the oracle."""
    rng = np.random.default_rng(5)
    rtab = synth.residue_table_protein()
    q = synth._random_residues(23, 1, 600, rtab)
    seqs = []
    for k in range(8):                                             # windowed: longer than the threshold of 5 000
        body = rtab[rng.integers(0, len(rtab), 6000 + 37 * k)].astype(np.uint8)
        at = 700 + 450 * k                                         # straddling window starts (every 300 columns)
        body[at:at + 300] = q[150:450]                             # about 1 550: inside the exact f16 range, so the kernel's own
        seqs.append(body)
    for k in range(40):                                            # left whole, longer than any window
        body = rtab[rng.integers(0, len(rtab), 4500 - 30 * k)].astype(np.uint8)
        at = 1600 + 10 * k                                         # result is what is checked, not a re-queued recomputation
        body[at:at + 300 - 3 * k] = q[200: 500 - 3 * k]
        seqs.append(body)
    res, off = swipe_amd.synth_db(41, 600)
    seqs += [res[off[i]:off[i + 1]] for i in range(600)]
    r2, o2 = oracle.pack(seqs)
    Mo = oracle.matrix_builtin("BLOSUM62")
    want = oracle.search_all63(r2, o2, q, Mo, 16, 11, threads=THREADS)
    db = swipe_amd.Database.from_arrays(r2, o2)
    db.set_scoring(swipe_amd.matrix_builtin("BLOSUM62"), 5, 11)
    db.set_option("window", 5000)
    db.set_option("window_step", 300)
    db.set_option("force_mp", 1)
    for boundary_mb in (None, 1):                                  # the common allotment, and one too small for the long batches
        db.set_option("boundary_mb", boundary_mb)
        got, c = db.search(q)
        assert c["narrow_rows"] in range(9, 17) and np.array_equal(got, want), boundary_mb
    assert 1200 < int(want[8]) < 1800 and 1200 < int(want[0]) < 1800 and c["wide"] == 0      # nothing was recomputed
    # two queries (the dual policy of the same kernel) over the one-sequence-per-row view: 4 long sequences fill a batch there
    q2 = q[::-1].copy()
    want2 = oracle.search_all63(r2, o2, q2, Mo, 16, 11, threads=THREADS)
    db.set_option("dual_mp", 1)
    t1, t2, c2 = db.search2(q, q2)
    assert np.array_equal(t1, want) and np.array_equal(t2, want2)
    db.close()


def test_short_queries_window_a_titin_sized_subject_by_themselves():
    """round 3: whether a long sequence is cut into windows depends on how many sequences the chosen kernel keeps in flight.  A
    30-residue query runs one lane per sequence pair - 1 536 pairs per CU instead of the 64 of the 16-lane chains - so a
    35 000-residue protein, 5 ms on its one lane, would outlast the search of 300 000 ordinary sequences several times over
    (round 2's rule, tuned for 16-lane chains, left it whole).  Automatic windows: every score equal to the oracle's (exact and
    top-K with the bound build), and the first pass at least 1.5 x faster than with windows off."""
    rng = np.random.default_rng(21)
    rtab = synth.residue_table_protein()
    q = synth._random_residues(77, 1, 30, rtab)
    res, off = swipe_amd.synth_db(19, 300_000)
    body = rtab[rng.integers(0, len(rtab), 35_000)].astype(np.uint8)
    for at in (5, 17_000, 34_960):
        body[at:at + 30] = q                                      # copies at both ends and across a window start
    res = np.concatenate([res, body])
    off = np.concatenate([off, [off[-1] + len(body)]]).astype(np.int64)
    db = swipe_amd.Database.from_arrays(res, off)
    db.set_scoring(swipe_amd.matrix_builtin("BLOSUM62"), 11, 1)
    Mo = oracle.matrix_builtin("BLOSUM62")
    pick = np.unique(np.concatenate([rng.integers(0, 300_000, 3000), [300_000]]))
    sub = [res[off[i]:off[i + 1]] for i in pick]
    r2, o2 = oracle.pack(sub)
    want = oracle.search_all63(r2, o2, q, Mo, 12, 1, threads=THREADS)
    times = {}
    for mode in (None, 0):
        db.set_option("window", mode)
        best = 1e9
        for _ in range(3):
            scores, c = db.search(q)
            best = min(best, c["kernel_ms"])
        times[mode] = best
        assert c["narrow_shifted"] == 11 and np.array_equal(scores[pick], want), mode
    assert int(want[-1]) == int(oracle.search_all63(*oracle.pack([q]), q, Mo, 12, 1)[0])          # the long one carries the query whole
    assert under_interpreter() or times[None] * 1.5 < times[0], times
    db.set_option("window", None)
    full, _ = db.search(q)
    for bound in (0, 1):
        db.set_option("bound", bound)
        hits, tot, obv, c = db.search_topk(q, keep=40, minscore=70)
        assert (hits, tot, obv) == _expected_topk(full, 40, 70), bound
    db.close()


@pytest.mark.parametrize("step", [64, 333, 1000])
def test_windows_are_exact_for_alignments_that_span_long_gaps(step):
    """the overlap is the longest span a positive-scoring alignment can have, qlen (1 + hi / R): planted alignments with
    insertions of up to 400 columns, windows forced on every sequence longer than 300 with starts every `step` columns
    so that the alignments straddle window starts everywhere - scores equal to the oracle's under three gap systems"""
    rng = np.random.default_rng(step)
    rtab = synth.residue_table_protein()
    q = synth._random_residues(11, 1, 120, rtab)
    res, off = swipe_amd.synth_db(17, 400)
    seqs = [res[off[i]:off[i + 1]] for i in range(400)]
    for k in range(60):
        g = int(rng.integers(1, 400))
        ins = rtab[rng.integers(0, len(rtab), g)].astype(np.uint8)
        a = int(rng.integers(30, 90))
        left = rtab[rng.integers(0, len(rtab), int(rng.integers(0, 2500)))].astype(np.uint8)
        right = rtab[rng.integers(0, len(rtab), int(rng.integers(0, 900)))].astype(np.uint8)
        seqs.append(np.concatenate([left, q[:a], ins, q[a:], right]))
    r2, o2 = oracle.pack(seqs)
    db = swipe_amd.Database.from_arrays(r2, o2)
    db.set_option("window", 300)
    db.set_option("window_step", step)
    for go, ge in ((11, 1), (5, 2), (0, 1)):
        db.set_scoring(swipe_amd.matrix_builtin("BLOSUM62"), go, ge)
        want = oracle.search_all63(r2, o2, q, oracle.matrix_builtin("BLOSUM62"), go + ge, ge, threads=THREADS)
        got, c = db.search(q)
        assert np.array_equal(got, want), (go, ge)
        hits, tot, obv, c = db.search_topk(q, keep=30, minscore=60)
        assert (hits, tot, obv) == _expected_topk(want, 30, 60)
    db.close()


def test_chromosome_sized_nucleotide_subject_both_strands():
    """a 3 Mbp nucleotide sequence among short reads, 1 kb query on both strands: windows over the 4-bit stream of the
    two-query kernel; plus-strand and reverse-complement hits planted far apart; time no longer proportional to the
    longest sequence"""
    rng = np.random.default_rng(5)
    tab = synth.residue_table_nucleotide()
    q = synth._random_residues(99, 1, 1000, tab)
    qm = blastdb.revcomp_nt16(q)
    res, off = swipe_amd.synth_db(3, 3000, protein=False)
    seqs = [res[off[i]:off[i + 1]] for i in range(3000)]
    big = tab[rng.integers(0, len(tab), 3_000_000)].astype(np.uint8)
    big[500_000:501_000] = q
    big[2_200_000:2_201_000] = qm
    big[2_999_400:] = q[:600]
    seqs.append(big)
    r2, o2 = oracle.pack(seqs)
    Mo = oracle.matrix_nucleotide(1, -3)
    w1 = oracle.search_all63(r2, o2, q, Mo, 7, 2, threads=THREADS)
    w2 = oracle.search_all63(r2, o2, qm, Mo, 7, 2, threads=THREADS)
    db = swipe_amd.Database.from_arrays(r2, o2, symtype=0)
    db.set_scoring(swipe_amd.matrix_nucleotide(1, -3), 5, 2)
    s1, s2, c = db.search2(q, qm)
    assert np.array_equal(s1, w1) and np.array_equal(s2, w2) and int(w1[3000]) == 1000 and int(w2[3000]) == 1000
    assert under_interpreter() or c["kernel_ms"] < 1000, c                       # one chain over 3 M columns would take about 4 s
    hits, tot, obv, _ = db.search2_topk(q, qm, keep=10, minscore=100)
    assert hits == [(3000, 1000, 0), (3000, 1000, 1)] and tot == 2      # one score per (sequence, strand): the maximum over its windows
    db.close()


@pytest.mark.parametrize("lens", [(375, 375), (375, 300), (300, 375), (100, 90), (40, 33), (600, 520), (1300, 1100), (2300, 2000)])
def test_two_different_queries_in_one_pass(lens):
    """swa_search_pair_topk: two queries of a multi-query file in the two halves of the packed lanes, the shorter padded
    with rows that never score; each keeps its own thresholds, hit list and counts - equal to two separate searches and
    to the oracle, with the exact first pass and with the bound build"""
    rtab = synth.residue_table_protein()
    qa = synth._random_residues(1001, 1, lens[0], rtab)
    qb = synth._random_residues(2002, 1, lens[1], rtab)
    res, off = swipe_amd.synth_db(41, 1500)
    seqs = [res[off[i]:off[i + 1]] for i in range(1500)]
    rng = np.random.default_rng(lens[0] * 7 + lens[1])
    for k in range(60):                                      # relatives of either query, of both in one sequence
        src = (qa, qb)[k % 2]
        a = int(rng.integers(0, max(1, len(src) - 20)))
        piece = src[a:a + int(rng.integers(10, 120))].copy()
        mut = rng.random(len(piece)) < rng.random() * 0.3
        piece[mut] = rtab[rng.integers(0, len(rtab), int(mut.sum()))]
        seqs.append(np.concatenate([seqs[k][:30], piece, seqs[k + 1][:30]]))
    seqs += [qa, qb, np.concatenate([qb, qa]), np.zeros(0, np.uint8)]
    r2, o2 = oracle.pack(seqs)
    Mo = oracle.matrix_builtin("BLOSUM62")
    wa = oracle.search_all63(r2, o2, qa, Mo, 12, 1, threads=THREADS)
    wb = oracle.search_all63(r2, o2, qb, Mo, 12, 1, threads=THREADS)
    db = swipe_amd.Database.from_arrays(r2, o2)
    db.set_scoring(swipe_amd.matrix_builtin("BLOSUM62"), 11, 1)
    for bound in (0, 1, None):
        db.set_option("bound", bound)
        for (lo_a, hi_a), (lo_b, hi_b) in (((1, 1 << 62), (1, 1 << 62)), ((70, 1 << 62), (45, 200)), ((45, 150), (90, 1 << 62))):
            (h1, t1, o1), (h2, t2, o2b), c = db.search_pair_topk(qa, qb, keep=(40, 25), minscore=(lo_a, lo_b), maxscore=(hi_a, hi_b))
            assert (h1, t1, o1) == _expected_topk(wa, 40, lo_a, hi_a), (bound, lo_a)
            assert (h2, t2, o2b) == _expected_topk(wb, 25, lo_b, hi_b), (bound, lo_b)
            assert c["cells"] == int(o2[-1]) * (len(qa) + len(qb))
    db.close()


def test_blast_volumes_opened_with_an_hbm_budget_are_streamed(tmp_path):
    """swa_db_open_streamed: two BLAST v4 volumes behind an alias, a slice of them opened with a budget below what it
    needs resident - same hits, counts and scores as swa_db_open of the same slice"""
    q = cases.Q375
    res, off = swipe_amd.synth_db(5, 40_000, query=q)
    cut = 22_000
    swipe_amd.write_blastdb(str(tmp_path / "v0"), res[:off[cut]], off[:cut + 1], first_id=0)
    swipe_amd.write_blastdb(str(tmp_path / "v1"), res[off[cut]:], off[cut:] - off[cut], first_id=cut)
    blastdb.write_alias(str(tmp_path / "both"), [str(tmp_path / "v0"), str(tmp_path / "v1")])
    M = swipe_amd.matrix_builtin("BLOSUM62")
    for first, last in ((0, -1), (3_000, 36_999)):
        resident = swipe_amd.Database.open(str(tmp_path / "both"), first_seqno=first, last_seqno=last)
        streamed = swipe_amd.Database.open(str(tmp_path / "both"), first_seqno=first, last_seqno=last, hbm_budget=20 << 20)
        assert streamed.info()["hbm_bytes"] < 0.8 * resident.info()["hbm_bytes"]
        assert streamed.info()["seqcount"] == resident.info()["seqcount"]
        for db in (resident, streamed):
            db.set_scoring(M, 11, 1)
        assert np.array_equal(resident.search(q)[0], streamed.search(q)[0])
        for minscore in (45, 80):
            a, b = resident.search_topk(q, keep=100, minscore=minscore), streamed.search_topk(q, keep=100, minscore=minscore)
            assert a[:3] == b[:3] and a[1] > 0
        resident.close()
        streamed.close()


def test_database_larger_than_its_hbm_budget_is_streamed():
    """a shard that may use less device memory than its resident form needs stays in page-locked host memory and is
    walked through two device slots, part by part, double-buffered: same scores, hit lists and counts as the resident
    shard for every budget, in both walking directions, with the bound build and with score windows"""
    q = cases.Q375
    res, off = swipe_amd.synth_db(1, 40_000, query=q)
    Mo = oracle.matrix_builtin("BLOSUM62")
    want = oracle.search_all63(res, off, q, Mo, 12, 1, threads=THREADS)
    resident = swipe_amd.Database.from_arrays(res, off, first_seqno=1000)
    resident.set_scoring(swipe_amd.matrix_builtin("BLOSUM62"), 11, 1)
    full = resident.info()["hbm_bytes"]
    for budget_mb in (34, 26, 18.5):                         # 2 slots x (budget / 2 - 8 MB of fixed allowance): 4, 6 and ~24 parts
        db = swipe_amd.Database.from_arrays(res, off, first_seqno=1000, hbm_budget=int(budget_mb * (1 << 20)))
        info = db.info()
        assert info["seqcount"] == 40_000 and info["symcount"] == int(off[-1]) and info["hbm_bytes"] < 0.8 * full
        db.set_scoring(swipe_amd.matrix_builtin("BLOSUM62"), 11, 1)
        for rnd in range(3):                                 # the walk alternates direction
            scores, c = db.search(q)
            assert np.array_equal(scores, want) and c["cells"] == int(off[-1]) * len(q)
        for bound in (None, 1, 0):
            db.set_option("bound", bound)
            for minscore, maxscore in ((40, 1 << 62), (80, 1 << 62), (45, 120)):
                got = db.search_topk(q, keep=60, minscore=minscore, maxscore=maxscore)
                ref = resident.search_topk(q, keep=60, minscore=minscore, maxscore=maxscore)
                assert got[:3] == ref[:3]
                exp = _expected_topk(want, 60, minscore, maxscore)
                assert ([(s - 1000, v) for s, v in got[0]], got[1], got[2]) == exp
        # two different queries per pass over the parts
        q2 = cases.Q375[::-1][:330].copy()
        (h1, t1, o1), (h2, t2, o2), _ = db.search_pair_topk(q, q2, keep=(30, 20), minscore=(50, 45))
        (r1, rt1, ro1), (r2, rt2, ro2), _ = resident.search_pair_topk(q, q2, keep=(30, 20), minscore=(50, 45))
        assert (h1, t1, o1, h2, t2, o2) == (r1, rt1, ro1, r2, rt2, ro2) and t1 > 0 and t2 > 0
        db.close()
    resident.close()
    # a nucleotide database, both strands: parts carry the 4-bit one-sequence-per-row tables
    tab = synth.residue_table_nucleotide()
    qn = synth._random_residues(99, 1, 400, tab)
    qm = blastdb.revcomp_nt16(qn)
    res, off = swipe_amd.synth_db(3, 30_000, protein=False)
    res = res.copy()
    res[off[777]:off[777] + 300] = qn[:300]
    res[off[20_001]:off[20_001] + 250] = qm[100:350]
    resident = swipe_amd.Database.from_arrays(res, off, symtype=0)
    resident.set_scoring(swipe_amd.matrix_nucleotide(1, -3), 5, 2)
    want = resident.search2_topk(qn, qm, keep=50, minscore=25)
    db = swipe_amd.Database.from_arrays(res, off, symtype=0, hbm_budget=int(19 * (1 << 20)))
    assert db.info()["hbm_bytes"] < resident.info()["hbm_bytes"]
    db.set_scoring(swipe_amd.matrix_nucleotide(1, -3), 5, 2)
    for rnd in range(2):
        got = db.search2_topk(qn, qm, keep=50, minscore=25)
        assert got[:3] == want[:3] and got[1] >= 2 and {h[2] for h in got[0]} == {0, 1}      # hits on both strands
    db.close()
    resident.close()


@pytest.mark.late
def test_streamed_shard_answers_every_entry_point():
    """a budgeted shard answers what a resident one does (round 4): two queries in one walk, and - the owning part bound to
    a slot for the call - end points, alignments, sequence fetches; only inclusion masks need a resident shard"""
    q = cases.Q375
    res, off = swipe_amd.synth_db(1, 40_000, query=q)
    want = oracle.search_all63(res, off, q, oracle.matrix_builtin("BLOSUM62"), 12, 1, threads=THREADS)
    resident = swipe_amd.Database.from_arrays(res, off, first_seqno=1000)
    resident.set_scoring(swipe_amd.matrix_builtin("BLOSUM62"), 11, 1)
    for budget_mb in (34, 18.5):
        db = swipe_amd.Database.from_arrays(res, off, first_seqno=1000, hbm_budget=int(budget_mb * (1 << 20)))
        db.set_scoring(swipe_amd.matrix_builtin("BLOSUM62"), 11, 1)
        s1, s2, _ = db.search2(q, q[::-1].copy())
        r1s, r2s, _ = resident.search2(q, q[::-1].copy())
        assert np.array_equal(s1, r1s) and np.array_equal(s2, r2s) and np.array_equal(s1, want)
        top = [s for s, _ in resident.search_topk(q, keep=25, minscore=40)[0]]
        assert [list(map(int, a)) for a in db.search_endpoints(q, top)] == [list(map(int, a)) for a in resident.search_endpoints(q, top)]
        assert db.align(q, top) == resident.align(q, top)
        assert all(np.array_equal(db.sequence(s), resident.sequence(s)) for s in top[:5])
        # ... and, since round 5, inclusion sets (every part re-plans its tables): the odd sequences only, then all again
        odd = (np.arange(40_000) % 2).astype(np.uint8)
        db.set_inclusion(odd)
        resident.set_inclusion(odd)
        assert db.search_topk(q, keep=25, minscore=40)[:3] == resident.search_topk(q, keep=25, minscore=40)[:3]
        db.set_inclusion(None)
        resident.set_inclusion(None)
        assert db.search_topk(q, keep=25, minscore=40)[:3] == resident.search_topk(q, keep=25, minscore=40)[:3]
        db.close()
    resident.close()


def test_bench_shards_one_database_over_two_ranks():
    """bench.py --gpus 2 as the driver launches it (torch.distributed.run, one process per rank), on the one GPU of this box:
    both ranks use device 0 and the collectives run over gloo (RCCL refuses two ranks on one GPU).  The ONE database is cut by
    parallel.shard_bounds, each rank generates and searches only its residue-balanced slice, the gathered hit list, totalhits
    and every verified score equal the single-rank run's"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    args = ["--nseq", "400000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-secondary", "--verify-sample", "2000"]
    one = _bench_line({}, *args)
    env = dict(os.environ, SWA_BENCH_DEVICE="0", SWA_BENCH_BACKEND="gloo")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                          "127.0.0.1", "--master-port", "29531", os.path.join(root, "bench.py"), "--gpus", "2", *args],
                         env=env, capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, out.stderr[-3000:]
    two = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert two["n_gpus"] == 2 and two["scaling"] == "strong" and two["collectives"] == "gloo"
    assert two["hits_sha1"] == one["hits_sha1"] and two["search"]["totalhits"] == one["search"]["totalhits"]
    assert two["config"]["residues_total"] == one["config"]["residues_total"]
    assert abs(two["config"]["residues_rank0"] * 2 - two["config"]["residues_total"]) < 40_000      # balanced by residues
    assert two["verified_vs_oracle"] >= 2000


def test_bench_workload_protein100m_over_two_ranks():
    """BASELINE.json configs[4] (`--workload protein100M`) through the N-rank code path at a size the test box takes: 400 000
    sequences over two ranks (gloo, both on device 0) against the same workload on one rank"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    args = ["--workload", "protein100M", "--nseq", "400000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-secondary",
            "--verify-sample", "2000"]
    one = _bench_line({}, *args)
    env = dict(os.environ, SWA_BENCH_DEVICE="0", SWA_BENCH_BACKEND="gloo")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                          "127.0.0.1", "--master-port", "29533", os.path.join(root, "bench.py"), "--gpus", "2", *args],
                         env=env, capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, out.stderr[-3000:]
    two = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert two["n_gpus"] == 2 and two["hits_sha1"] == one["hits_sha1"]
    assert two["search"]["totalhits"] == one["search"]["totalhits"] and two["search"]["top_hit"] == one["search"]["top_hit"]
    assert two["verified_vs_oracle"] >= 2000 and "roofline" in two


def test_bench_default_line_carries_configs_3_and_4_as_secondary_sections():
    """the default N = 1 run: headline + exact first pass + `secondary` = [nucleotide (configs[3]), the big protein database
    on one GPU (configs[4]'s database), two queries per pass], each with its own roofline and oracle verification - here at
    sizes the test box takes (the driver's run uses 50 M and 100 M sequences)"""
    line = _bench_line({}, "--nseq", "300000", "--steps", "2", "--warmup", "1", "--no-cold", "--no-live-traffic", "--verify-sample", "1500",
                       "--secondary-nt-nseq", "100000", "--secondary-protein-nseq", "500000")
    assert line["metric"] == "GCUPS, 375-aa query vs 10M-seq protein db at 1/2/4/8 GPUs; bit-exact scores"      # BASELINE.json's string
    assert "exact_first_pass.value" in line["config"]["workload"] and line["verified_vs_oracle"] >= 1500
    nt, big, pair = line["secondary"]
    assert "configs[3]" in nt["metric"] and nt["verified_vs_oracle"] >= 300 and nt["roofline"]["bound"] == "hbm" and nt["value"] > 0
    assert nt["cpu_baseline"]["kind"] in ("reference", "port")
    assert "configs[4]" in big["metric"] and big["verified_vs_oracle"] >= 375 and big["roofline"]["frac"] > 0 and big["value"] > 0
    assert "500000 synthetic protein sequences" in big["config"]["workload"]
    assert pair["hits_identical"] is True
    assert line["cpu_baseline"]["value"] and line["roofline"]["achieved"] > 0


@pytest.mark.late
@pytest.mark.parametrize("golden", ["ntamb", "ntamb_overlap"])
@pytest.mark.parametrize("how", ["resident", "streaming in", "budget"])
def test_cli_nucleotide_ambiguity_tables_equal_reference_cli(tmp_path, how, golden):
    """tests/golden/ntamb.json (make_ntamb_golden.py): .nsq ambiguity tables in both forms of the format (32-bit entries in one
    volume, 64-bit entries in the other; database.cc:1284-1323), the query planted on both strands across the ambiguous runs -
    the unmodified reference's -m 8 / -m 0 / -m 7 output, byte for byte, from the old reader, from the pipelined open with the
    device-side unpack (chunks of 4 KiB: entries cut everywhere a chunk may end) and from a shard over its HBM budget.
    ntamb_overlap (round 6): the same database with its tables written back to front and runs that overlap inside the planted
    query - the reference applies entries in file order, the last writer wins (database.cc:1296-1321)"""
    import subprocess
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_ntamb_golden as G
    g = load_golden(golden)
    base, qf, sha = G.build(str(tmp_path), overlap=golden == "ntamb_overlap")
    assert sha == g["sha1_of_volumes"]                              # the same bytes the reference searched
    env = dict(os.environ)
    extra = []
    if how == "resident":
        env["SWA_PIPELINED"] = "0"
    elif how == "streaming in":
        env.update(SWA_LOAD_PART="16384", SWA_LOAD_CHUNK="4096")
    else:
        env["SWA_STREAM_RESERVE"] = "4096"
        extra = ["--hbm-budget", "140000"]
    exe = os.path.join(ROOT, "swipe_amd", "swipe_amd_cli")
    for m in ("8", "0", "7"):
        r = subprocess.run([exe, "-d", base, "-i", qf, "-m", m] + g["args"] + extra, capture_output=True, text=True, env=env)
        assert r.returncode == 0, r.stderr[-600:]
        text = r.stdout
        if m == "0":
            text = text[text.index("Sequences producing"):]
        assert text == g["m" + m], (how, m)


@pytest.mark.late
def test_bench_traffic_source_is_measured_or_an_explicit_fallback():
    """roofline.traffic of the default line comes from `rocprofv3 --pmc` passes run inside bench.py itself; whatever
    happens to them - no rocprofv3, a csv it cannot read, an overrun - the line must come out with a traffic_source that
    says which of the two it is, and a measured figure must be in the neighbourhood of the algorithmic bytes"""
    line = _bench_line({}, "--nseq", "400000", "--steps", "2", "--warmup", "1", "--traffic-only")
    roof = line["roofline"]
    src = roof.get("traffic_source") or ""
    measured = src.startswith("measured in this run")
    assert measured or "live pass not available" in src or src.startswith("not measured"), src
    if measured:
        algo = 400000 * 12 + line["config"]["residues_rank0"]
        assert 0.8 * algo < roof["traffic"] < 3.0 * algo, (roof["traffic"], algo)
    assert roof["achieved"] > 0 and line["value"] > 0


def test_bench_predicts_the_scaling_curve_from_one_gpu():
    """--predict-scaling: shards of N = 1, 2, 4, 8 timed one by one; merged lists identical at every N; efficiencies sane"""
    r = _bench_line({}, "--predict-scaling", "--nseq", "400000", "--steps", "2")
    rows = r["rows"]
    assert [x["n_gpus"] for x in rows] == [1, 2, 4, 8] and r["merged_lists_identical"] is True
    assert all(len(x["shard_ms"]) == x["n_gpus"] for x in rows)
    assert rows[0]["predicted_efficiency"] == 1.0 and all(0.2 < x["predicted_efficiency"] <= 1.1 for x in rows)


def test_windows_compose_with_subsets_translation_and_streaming():
    """long sequences cut into windows inside the other ways a shard can be held: with an inclusion subset (excluded long
    sequences are not windowed at all, included ones are), as six translated frames of a long nucleotide sequence
    (tblastn), and in a streamed database whose parts each hold a long sequence"""
    rng = np.random.default_rng(11)
    rtab = synth.residue_table_protein()
    q = cases.Q375[:200]
    Mo = oracle.matrix_builtin("BLOSUM62")
    res, off = swipe_amd.synth_db(23, 600)
    seqs = [res[off[i]:off[i + 1]] for i in range(600)]
    for k in range(3):                                       # three long sequences, each carrying the query somewhere
        body = rtab[rng.integers(0, len(rtab), 20_000 + 3_000 * k)].astype(np.uint8)
        at = int(rng.integers(0, len(body) - len(q)))
        body[at:at + len(q)] = q
        seqs.insert(200 * k + 7, body)
    r2, o2 = oracle.pack(seqs)
    want = oracle.search_all63(r2, o2, q, Mo, 12, 1, threads=THREADS)
    db = swipe_amd.Database.from_arrays(r2, o2)
    db.set_scoring(swipe_amd.matrix_builtin("BLOSUM62"), 11, 1)
    got, _ = db.search(q)
    assert np.array_equal(got, want)
    inc = np.ones(len(seqs), dtype=np.uint8)
    inc[7] = 0                                               # one of the long ones is excluded
    inc[::5] = 0
    db.set_inclusion(inc)
    got, _ = db.search(q)
    assert np.array_equal(got[inc == 1], want[inc == 1]) and np.all(got[inc == 0] == -1)
    hits, tot, obv, _ = db.search_topk(q, keep=20, minscore=60)
    sub = np.where(inc == 1, want, -1)
    assert (hits, tot, obv) == _expected_topk(sub, 20, 60)
    db.close()
    # streamed: two slots, parts of a few hundred sequences
    sdb = swipe_amd.Database.from_arrays(r2, o2, hbm_budget=int(17.5 * (1 << 20)))
    sdb.set_scoring(swipe_amd.matrix_builtin("BLOSUM62"), 11, 1)
    got, _ = sdb.search(q)
    assert np.array_equal(got, want)
    sdb.close()
    # tblastn: a 90 kb nucleotide sequence among short ones, six frames each; windows are cut in the translated frames
    tab = synth.residue_table_nucleotide()
    nres, noff = swipe_amd.synth_db(3, 300, protein=False)
    nseqs = [nres[noff[i]:noff[i + 1]] for i in range(300)]
    nseqs.insert(50, tab[rng.integers(0, len(tab), 90_000)].astype(np.uint8))
    n2, no2 = oracle.pack(nseqs)
    tdb = swipe_amd.Database.from_arrays(n2, no2, translate_gencode=1)
    tdb.set_scoring(swipe_amd.matrix_builtin("BLOSUM62"), 11, 1)
    tdb.set_option("window", 3000)                           # 30 000-residue frames: windowed
    s_win, _ = tdb.search(q)
    tdb.set_option("window", 0)
    s_whole, _ = tdb.search(q)
    assert np.array_equal(s_win, s_whole) and len(s_win) == 6 * len(nseqs)
    table = oracle.translate_table(1)
    for frame in range(6):
        prot = oracle.translate(nseqs[50], frame // 3, frame % 3, table)
        assert int(s_win[6 * 50 + frame]) == oracle.fullsw(prot, q, Mo, 12, 1)
    tdb.close()


_FOLLOWER_SCRIPT = r"""
import os, sys, numpy as np
sys.path.insert(0, %r)
os.environ["SWA_WATCHDOG_S"] = "30"          # a relapse is an error message with the device's control block, not a hang
import swipe_amd
from swipe_amd import blastdb, synth
q = blastdb.encode_protein(synth.QUERY_P07327)
res, off = swipe_amd.synth_db(1, %d, query=q)
lens = np.diff(off)
cand = np.nonzero((lens > 330) & (lens < 420))[0]
pick = cand[:: max(1, len(cand) // 16)][:16]          # database sequences with families of their own: the lists are not empty
db = swipe_amd.Database.from_arrays(res, off)
db.set_scoring(swipe_amd.matrix_builtin("BLOSUM62"), 11, 1)
rows, ref = [], {}
for forced in (None,):
    db.set_option("requeue_follow", forced)          # accepted and ignored since round 4 (the follower is gone)
    for k in (0, 2, 2, 4, 14, 2, 0, 2):
        a, b = (res[off[i]:off[i + 1]] for i in pick[k:k + 2])
        r = db.search_pair_topk(a, b, keep=250, minscore=(80, 80))
        rows.append((r[2]["narrow_shifted"], r[2]["narrow_rows"]))
        if k not in ref:
            one = [db.search_topk(x, keep=250, minscore=80) for x in (a, b)]
            ref[k] = (one[0][0], one[0][1], one[1][0], one[1][1])
        assert (r[0][0], r[0][1], r[1][0], r[1][1]) == ref[k], (forced, k)
print("OK", rows)
"""


def test_pairs_of_queries_one_after_the_other_on_one_handle_do_not_wait_for_each_other():
    """Round 3's CLI hung on the second pair of a query file against the 10 M-sequence database (found by tools/probe.py
    dropin): beside the 52-row two-query bound build - 512-thread blocks, 223 registers a wave, so that a block needs the
    whole register file of its CU - 256 follower waves of 68 registers reached the device together with the producer on a
    warm handle, and the producer's queue head froze for good (control block via option watchdog_s: all 256 blocks
    "started", none finished, every follower alive).  Three defences, each tested here: the host starts a follower only
    beside builds it can share a SIMD with (at most 48 rows per lane); a follower leaves a producer that is not on the
    device in full or whose queue head stands still (requeue_follow = 128 forces the follower beside the 49..52-row builds:
    the searches must still come back, with the same hits); the producer's end flag no longer waits for blocks that never
    started.  Round 4 took the follower out altogether (no kernel waits for another kernel); the test stays as
    the regression for pairs of queries one after the other on a warm handle of the 10 M-sequence database.  In a child
    process with the watchdog on, so that a hang is a failed test and not a hung suite"""
    import subprocess
    import sys
    nseq = 40_000 if under_interpreter() else 10_000_000      # (interpreted kernels: the same builds meet on a smaller shard)
    r = subprocess.run([sys.executable, "-c", _FOLLOWER_SCRIPT % (ROOT, nseq)], capture_output=True, text=True, timeout=1500 if under_interpreter() else 400)
    assert r.returncode == 0 and r.stdout.startswith("OK"), (r.stdout[-500:], r.stderr[-2500:])
    assert under_interpreter() or ("(10, 52)" in r.stdout and "(10, 48)" in r.stdout and "(10, 49)" in r.stdout)   # (which builds meet depends on the shard's own sequences)


_WARM_SCRIPT = r"""
import os, sys, numpy as np
sys.path.insert(0, %r)
os.environ["SWA_WATCHDOG_S"] = "30"
import swipe_amd
from swipe_amd import blastdb, synth
q0 = blastdb.encode_protein(synth.QUERY_P07327)
res, off = swipe_amd.synth_db(1, %d, query=q0)
lens = np.diff(off)
rng = np.random.default_rng(%d)
db = swipe_amd.Database.from_arrays(res, off)
db.set_scoring(swipe_amd.matrix_builtin("BLOSUM62"), 11, 1)
ref = swipe_amd.Database.from_arrays(res, off)            # the same shard, exact first pass, no follower: what must come out
ref.set_scoring(swipe_amd.matrix_builtin("BLOSUM62"), 11, 1)
ref.set_option("bound", 0); ref.set_option("requeue_follow", 0)
def query():
    n = int(rng.choice([rng.integers(5, 64), rng.integers(64, 520), rng.integers(520, 1300)]))
    if rng.random() < 0.7:                                  # a database sequence (it has itself, and perhaps a family, as hits) ...
        c = np.nonzero(lens == n)[0]
        if len(c):
            i = int(rng.choice(c)); return res[off[i]:off[i + 1]].copy()
    return synth._random_residues(int(rng.integers(1 << 30)), 1, n, synth.residue_table_protein())   # ... or a random one
forms, prev = set(), None
for it in range(%d):
    a = query()
    if prev is not None and rng.random() < 0.5 and 4 * min(len(a), len(prev)) >= 3 * max(len(a), len(prev)):
        r = db.search_pair_topk(prev, a, keep=100, minscore=(70, 70))
        w = [ref.search_topk(x, keep=100, minscore=70) for x in (prev, a)]
        assert (r[0][0], r[0][1], r[1][0], r[1][1]) == (w[0][0], w[0][1], w[1][0], w[1][1]), ("pair", it, len(prev), len(a))
        forms.add((r[2]["narrow_shifted"], r[2]["narrow_rows"]))
    else:
        r = db.search_topk(a, keep=100, minscore=70)
        w = ref.search_topk(a, keep=100, minscore=70)
        assert r[:3] == w[:3], ("one", it, len(a))
        forms.add((r[3]["narrow_shifted"], r[3]["narrow_rows"]))
    prev = a
print("OK", len(forms), sorted(forms))
"""


def test_a_query_file_of_mixed_lengths_on_one_warm_handle():
    """the fuzz tools open a fresh handle per configuration, and a fresh handle is exactly what hid round 3's follower hang
    (kernels whose first launch is slow never meet on the device).  Here ONE handle of a 3 M-sequence database takes 90
    searches back to back - database sequences and random queries of 5..1300 residues, singly and two per pass, whatever
    build the table picks for each (one lane, chains of 2..16 lanes, long lanes, passes) - and every hit list must equal the
    one a second handle computes with the exact first pass and no follower.  Child process, watchdog on"""
    import subprocess
    import sys
    nseq = 20_000 if under_interpreter() else 3_000_000
    r = subprocess.run([sys.executable, "-c", _WARM_SCRIPT % (ROOT, nseq, 20260929, 90)], capture_output=True, text=True, timeout=2400 if under_interpreter() else 600)
    assert r.returncode == 0 and r.stdout.startswith("OK"), (r.stdout[-500:], r.stderr[-2500:])
    assert int(r.stdout.split()[1]) >= (15 if under_interpreter() else 25)                    # that many different builds met on the one handle
