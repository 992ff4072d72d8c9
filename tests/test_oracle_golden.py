"""CPU: the oracle (oracle/sw_oracle.c) against the committed outputs of the compiled reference
(tests/golden/*.json, produced by tests/golden/make_golden.py from oracle/_ref).  This is what
pins the oracle - every later parity claim rests on it."""
import numpy as np
import pytest

import cases
import oracle
from conftest import case_matrix, load_golden
from swipe_amd import blastdb

NAMES = [f.__name__[5:] for f in cases.ALL]


def strands(case):
    if case.protein:
        return [case.query]
    return [case.query, blastdb.revcomp_nt16(case.query)]


@pytest.mark.parametrize("name", NAMES)
def test_case_regenerates_identically(name):
    assert cases.get(name).checksum() == load_golden(name)["checksum"]


@pytest.mark.parametrize("name", NAMES)
def test_lane_kernels_match_reference_raw_outputs(name):
    """7-bit (both builds), 16-bit + bestpos and 63-bit values of EVERY sequence, saturated ones included."""
    case, g = cases.get(name), load_golden(name)
    M = case_matrix(case, oracle)
    lo, hi, l7, l16 = oracle.score_limits(M)
    assert (l7, l16) == (g["scorelimit7"], g["scorelimit16"])
    goe, ge = case.gapopen + case.gapextend, case.gapextend
    qs = strands(case)
    assert len(g["raw"]) == len(case.seqs) * len(qs)
    for seqno, strand, length, s7a, s7b, s16, bp16, s63, s16s, bp16s, bq16s in g["raw"]:
        d, q = case.seqs[seqno], qs[strand]
        assert len(d) == length
        assert oracle.search7_lane(d, q, M, goe, ge) == s7a == s7b
        assert oracle.search16_lane(d, q, M, goe, ge) == (s16, bp16)
        assert oracle.search16s_lane(d, q, M, goe, ge) == (s16s, bp16s, bq16s)
        assert oracle.fullsw(d, q, M, goe, ge) == s63


@pytest.mark.parametrize("name", NAMES)
def test_escalation_delivers_exact_scores(name):
    """search_chunk's 7 -> 16 -> 63 ladder ends at the 63-bit score whenever a width saturates."""
    case, g = cases.get(name), load_golden(name)
    M = case_matrix(case, oracle)
    res, off = oracle.pack(case.seqs)
    goe, ge = case.gapopen + case.gapextend, case.gapextend
    for strand, q in enumerate(strands(case)):
        rows = [r for r in g["raw"] if r[1] == strand]
        scores, (c7, c16, c63) = oracle.search_chunk(res, off, q, M, goe, ge)
        assert list(scores) == [r[7] for r in rows]
        assert c7 == len(rows)
        assert c16 == sum(r[3] >= g["scorelimit7"] for r in rows)
        assert c63 == sum(r[3] >= g["scorelimit7"] and r[5] >= g["scorelimit16"] for r in rows)
        assert list(oracle.search_all63(res, off, q, M, goe, ge, threads=2)) == [r[7] for r in rows]


@pytest.mark.parametrize("name", NAMES)
def test_hit_list_rank_evalue_bits_match_cli(name):
    """hits_init thresholds + hits_enter order + E-value / bit-score strings of the reference CLI."""
    case, g = cases.get(name), load_golden(name)
    assert g["cli"]["1"] == g["cli"]["8"]          # thread count never changes the reference's answer
    cli = g["cli"]["1"]
    nsym = int(sum(len(s) for s in case.seqs))
    kw = dict(descriptions=case.keep, alignments=0, symtype=1 if case.protein else 0, matrix=case.matrix,
              match=case.match, mismatch=case.mismatch, gapopen=case.gapopen, gapextend=case.gapextend,
              qlen=len(case.query), dbseqs=len(case.seqs), dbsyms=nsym)
    h = oracle.HitList(**kw)
    for row in g["raw"]:
        seqno, strand, s63 = row[0], row[1], row[7]
        h.enter(seqno, s63, 0, 0, strand, 0)       # nucleotide minus strand enters as dstrand 1 (swipe.cc:1470)
    got = h.hits()
    assert [x[0] for x in got] == cli["seqno"]
    assert [x[1] for x in got] == cli["score"]
    if cli["strand"]:
        assert ["-" if x[3] else "+" for x in got] == cli["strand"]
    if h.c.stats_available:
        assert ["%.2g" % h.expect(x[1]) for x in got] == cli["evalue"]
        assert ["%.1f" % h.bits(x[1]) for x in got] == cli["bits"]
    else:
        assert [str(x[1]) for x in got] == cli["bits"]      # no statistics: the TSV shows the raw score


@pytest.mark.parametrize("name", NAMES)
def test_alignments_match_reference_align(name):
    """align() (align.cc) for every positive-scoring sequence, started both ways hits_align starts it:
    from scratch and from the search16s end point (hits.cc:587-616)."""
    case, g = cases.get(name), load_golden(name)
    M = case_matrix(case, oracle)
    assert g["align"]
    for seqno, ds, s16s, bp, bq, score, qs, dst, qe, de, cigar, hinted in g["align"]:
        d = blastdb.revcomp_nt16(case.seqs[seqno]) if ds else case.seqs[seqno]
        assert oracle.align(case.query, d, M, case.gapopen, case.gapextend) == (score, qs, dst, qe, de, cigar)
        assert (hinted is not None) == (s16s < g["scorelimit16"] and bq > 0 and bp != 0)
        if hinted is not None:
            assert oracle.align(case.query, d, M, case.gapopen, case.gapextend, (s16s, bq, bp)) == tuple(hinted)


def test_known_answers_from_baseline_md():
    """self hit 1957 = 758.4 bits; ties rank by descending sequence number."""
    M = oracle.matrix_builtin("BLOSUM62")
    assert oracle.fullsw(cases.Q375, cases.Q375, M, 12, 1) == 1957
    h = oracle.HitList(qlen=375, dbseqs=1000, dbsyms=344448)
    assert "%.1f" % h.bits(1957) == "758.4"
    assert "%.2g" % h.expect(1957) == "3.9e-221"
    for s in (5, 9, 7):
        h.enter(s, 100)
    assert [x[0] for x in h.hits()] == [9, 7, 5]


def test_stats_tables():
    assert oracle.default_gaps("BLOSUM62") == (11, 1)
    assert oracle.stats_protein("BLOSUM62", 11, 1)[:2] == (0.267, 0.041)
    assert oracle.stats_nucleotide(1, -3, 5, 2)[:2] == (1.374, 0.711)     # both >= maxima -> the (0,0) row
    assert oracle.stats_protein("BLOSUM62", 3, 3) is None


TNAMES = [f.__name__[5:] for f in cases.TRANSLATED]


def frame_sets(case):
    """(query frames, per-sequence database frames) the reference searches for -p 2/3/4 (swipe.cc:289-324)"""
    qt = oracle.translate_table(case.query_gencode)
    dt = oracle.translate_table(case.db_gencode)
    qf = oracle.frames(case.query, qt) if case.sym in (2, 4) else [case.query]
    df = [oracle.frames(s, dt) if case.sym in (3, 4) else [s] for s in case.seqs]
    return qf, df


@pytest.mark.parametrize("name", TNAMES)
def test_translated_case_regenerates_identically(name):
    assert cases.get(name).checksum() == load_golden(name)["checksum"]


@pytest.mark.parametrize("name", TNAMES)
def test_translated_lanes_and_alignments_match_reference(name):
    """genetic-code tables + six-frame translation (query.cc:377-506, database.cc:1182-1218) feeding the same
    kernels: every (query frame, database frame) pair of every sequence, all lane widths, and align()."""
    case, g = cases.get(name), load_golden(name)
    M = case_matrix(case, oracle)
    goe, ge = case.gapopen + case.gapextend, case.gapextend
    qf, df = frame_sets(case)
    assert len(g["raw"]) == len(case.seqs) * len(qf) * len(df[0])
    for seqno, qtag, dtag, length, s7a, s7b, s16, bp16, s63, s16s, bp16s, bq16s in g["raw"]:
        d, q = df[seqno][dtag], qf[qtag]
        assert len(d) == length
        assert oracle.search7_lane(d, q, M, goe, ge) == s7a == s7b
        assert oracle.search16_lane(d, q, M, goe, ge) == (s16, bp16)
        assert oracle.search16s_lane(d, q, M, goe, ge) == (s16s, bp16s, bq16s)
        assert oracle.fullsw(d, q, M, goe, ge) == s63
    assert g["align"]
    for seqno, qtag, dtag, s16s, bp, bq, score, qs, dst, qe, de, cigar, hinted in g["align"]:
        d, q = df[seqno][dtag], qf[qtag]
        assert oracle.align(q, d, M, case.gapopen, case.gapextend) == (score, qs, dst, qe, de, cigar)
        if hinted is not None:
            assert oracle.align(q, d, M, case.gapopen, case.gapextend, (s16s, bq, bp)) == tuple(hinted)


@pytest.mark.parametrize("name", TNAMES)
def test_translated_hit_list_matches_cli(name):
    """hits_init for symtype 2/3/4 (lengths in codons, ungapped statistics for -p 4, maxhits per frame pair)
    and the insertion order (query frame outer, database frame inner, swipe.cc:1403-1470)."""
    case, g = cases.get(name), load_golden(name)
    assert g["cli"]["1"] == g["cli"]["8"]
    cli = g["cli"]["1"]
    nsym = int(sum(len(s) for s in case.seqs))
    h = oracle.HitList(descriptions=case.keep, alignments=0, symtype=case.sym, matrix=case.matrix, gapopen=case.gapopen,
                       gapextend=case.gapextend, qlen=len(case.query), dbseqs=len(case.seqs), dbsyms=nsym)
    qtags = sorted({r[1] for r in g["raw"]})
    for qt in qtags:                                     # one search_chunk pass per query frame
        for r in g["raw"]:
            if r[1] == qt:
                h.enter(r[0], r[8], qt // 3, qt % 3, r[2] // 3, r[2] % 3)
    got = h.full()
    assert [x[0] for x in got] == cli["seqno"]
    assert [x[1] for x in got] == cli["score"]
    label = lambda s, f: "%s%d" % ("-" if s else "+", f + 1)
    if case.sym == 2:
        assert [label(x[2], x[3]) for x in got] == cli["strand"]
    elif case.sym == 3:
        assert [label(x[4], x[5]) for x in got] == cli["strand"]
    else:
        assert [label(x[2], x[3]) + "/" + label(x[4], x[5]) for x in got] == cli["strand"]
    assert h.c.stats_available
    assert ["%.2g" % h.expect(x[1]) for x in got] == cli["evalue"]
    assert ["%.1f" % h.bits(x[1]) for x in got] == cli["bits"]


def _option_runs():
    g = load_golden("options")
    return [(name, i) for name in ("p1k", "nt") for i in range(len(g[name]["runs"]))]


@pytest.mark.parametrize("name,i", _option_runs())
def test_oracle_hits_init_under_the_reference_s_options(name, i):
    """tests/golden/options.json (the reference CLI under -c -u -e -k -z -v -b -S, other rewards / matrices / gap systems):
    the oracle's scalar scores + hits_init / hits_enter restatement reproduce the reference's hit list (sequence numbers
    and scores of the -m 7 view) and the E-value / bit-score columns of its -m 8 view."""
    import re
    from swipe_amd import blastdb
    g = load_golden("options")[name]
    case = cases.get(name)
    assert g["checksum"] == case.checksum()
    rec = g["runs"][i]
    opt = dict(zip(rec["options"][::2], rec["options"][1::2]))
    protein = case.protein
    matrix = opt.get("-M", "BLOSUM62")
    match, mismatch = int(opt.get("-r", 1)), int(opt.get("-q", -3))
    go = int(opt.get("-G", 11 if protein else 5))
    ge = int(opt.get("-E", 1 if protein else 2))
    strands = {"plus": 1, "minus": 2, "both": 3}.get(opt.get("-S", "3"), None) or int(opt.get("-S", 3))
    M = oracle.matrix_builtin(matrix) if protein else oracle.matrix_nucleotide(match, mismatch)
    res, off = oracle.pack(case.seqs)
    q = np.asarray(case.query, dtype=np.uint8)
    queries = [(q, 0)] if protein else [(x, s) for x, s in ((q, 0), (blastdb.revcomp_nt16(q), 1)) if (s + 1) & strands]
    kw = dict(descriptions=int(opt.get("-v", 250)), alignments=0, minscore=int(opt.get("-c", 1)), maxscore=int(opt.get("-u", 1 << 62)),
              minexpect=float(opt.get("-k", 0.0)), expect=float(opt.get("-e", 10.0)), symtype=1 if protein else 0, querystrands=strands,
              matrix=matrix, match=match, mismatch=mismatch, gapopen=go, gapextend=ge, qlen=len(q), dbseqs=len(case.seqs),
              dbsyms=int(off[-1]), effdbsize=int(opt.get("-z", 0)))
    h = oracle.HitList(**kw)
    for x, s in queries:
        scores = oracle.search_all63(res, off, x, M, go + ge, ge, threads=2)
        for seqno, sc in enumerate(scores):
            h.enter(seqno, int(sc), 0, 0, s, 0)
    got = h.hits()
    assert [x[0] for x in got] == list(map(int, re.findall(r"<track>(\d+)</track>", rec["m7"])))
    assert [x[1] for x in got] == list(map(int, re.findall(r"<score>(-?\d+)</score>", rec["m7"])))
    # -m 8 (its own list: alignments = -b default 100, descriptions -v): E-value and bit score columns of the rows it prints
    rows = [l.split("\t") for l in rec["m8"].splitlines()]
    h8 = oracle.HitList(**dict(kw, alignments=int(opt.get("-b", 100))))
    for x, s in queries:
        scores = oracle.search_all63(res, off, x, M, go + ge, ge, threads=2)
        for seqno, sc in enumerate(scores):
            h8.enter(seqno, int(sc), 0, 0, s, 0)
    top = h8.hits()[: len(rows)]
    if h8.c.stats_available:
        assert [r[10] for r in rows] == ["%.2g" % h8.expect(x[1]) for x in top]
        assert [r[11] for r in rows] == ["%.1f" % h8.bits(x[1]) for x in top]
    else:
        assert [r[10] for r in rows] == [str(x[1]) for x in top]
