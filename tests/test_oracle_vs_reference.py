"""CPU, wherever oracle/_ref/ exists (the build container, and the GPU box, where the built
binaries travel): the oracle against the COMPILED REFERENCE on freshly generated inputs that are
not among the committed goldens - other matrices, gap systems, query lengths and both alphabets.
Skipped when the reference binaries are absent."""
import os
import subprocess

import numpy as np
import pytest

import oracle
from conftest import ROOT
from swipe_amd import blastdb, synth

HARNESS = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
pytestmark = pytest.mark.skipif(not os.path.exists(HARNESS), reason="oracle/_ref/ref_harness not built")


def run_harness(tmp_path, seqs, query, protein, matrix, go, ge, match=1, mismatch=-3, align=False, symtype=None, qgc=1, dgc=1):
    base = str(tmp_path / "db")
    blastdb.write_db(base, seqs, protein=protein)
    sym = symtype if symtype is not None else (1 if protein else 0)
    alpha = blastdb.NCBI4NA if sym in (0, 2, 4) else blastdb.NCBISTDAA
    qf = str(tmp_path / "q.fa")
    open(qf, "w").write(">q\n" + "".join(alpha[c] for c in query) + "\n")
    out = subprocess.run([HARNESS, base, qf, str(sym), matrix if sym else "-", str(go), str(ge),
                          str(match), str(mismatch), "align" if align else "-", str(qgc), str(dgc)],
                         capture_output=True, text=True, check=True).stdout
    rows, arows = [], []
    for l in out.splitlines():
        if l.startswith("#"):
            continue
        f = l.split("\t")
        if f[0] == "A":
            k = f.index("|")
            head = [int(x) for x in f[1:k]]
            first = [int(x) for x in f[k + 1:k + 6]] + [f[k + 6]]
            rest = f[k + 8:]
            arows.append((head, tuple(first), None if rest[0] == "-" else tuple([int(x) for x in rest[:5]] + [rest[5]])))
        elif f[0] == "T":
            rows.append([int(x) for x in f[1:]])
        else:
            rows.append([int(x) for x in f])
    return (rows, arows) if align else rows


@pytest.mark.parametrize("matrix,go,ge,qlen,seed", [
    ("BLOSUM62", 11, 1, 375, 1), ("BLOSUM45", 14, 2, 90, 2), ("BLOSUM80", 10, 1, 211, 3),
    ("PAM30", 9, 1, 47, 4), ("PAM250", 14, 2, 600, 5), ("BLOSUM50", 13, 2, 33, 6), ("BLOSUM90", 10, 1, 128, 7),
])
def test_protein_lanes_match_compiled_reference(tmp_path, matrix, go, ge, qlen, seed):
    rtab = synth.residue_table_protein()
    q = synth._random_residues(1000 + seed, 3, qlen, rtab)
    seqs = synth.make_db(40 + seed, 150, query=q)
    mut = q.copy()
    mut[::7] = rtab[(np.arange(len(mut[::7])) * 37 + seed) % 4096]
    seqs += [q, mut, q[: qlen // 2], np.concatenate([seqs[0], mut[5:], seqs[1]]), np.zeros(0, np.uint8), q[:1]]
    M = oracle.matrix_builtin(matrix)
    rows = run_harness(tmp_path, seqs, q, True, matrix, go, ge)
    assert len(rows) == len(seqs)
    goe = go + ge
    for seqno, strand, length, s7a, s7b, s16, bp16, s63, s16s, bp16s, bq16s in rows:
        d = seqs[seqno]
        assert oracle.search7_lane(d, q, M, goe, ge) == s7a == s7b
        assert oracle.search16_lane(d, q, M, goe, ge) == (s16, bp16)
        assert oracle.search16s_lane(d, q, M, goe, ge) == (s16s, bp16s, bq16s)
        assert oracle.fullsw(d, q, M, goe, ge) == s63


@pytest.mark.parametrize("match,mismatch,go,ge,qlen,seed", [(1, -3, 5, 2, 400, 1), (2, -3, 5, 2, 150, 2), (1, -2, 2, 1, 77, 3),
                                                            (5, -4, 10, 6, 260, 4)])
def test_nucleotide_lanes_match_compiled_reference(tmp_path, match, mismatch, go, ge, qlen, seed):
    rtab = synth.residue_table_nucleotide()
    q = synth._random_residues(2000 + seed, 3, qlen, rtab)
    seqs = synth.make_db(60 + seed, 120, protein=False)
    amb = q.copy()
    amb[3:6] = 15
    seqs += [q, blastdb.revcomp_nt16(q), amb, q[10: qlen - 10], np.zeros(0, np.uint8)]
    M = oracle.matrix_nucleotide(match, mismatch)
    rows = run_harness(tmp_path, seqs, q, False, "-", go, ge, match, mismatch)
    qs = [q, blastdb.revcomp_nt16(q)]
    assert len(rows) == 2 * len(seqs)
    goe = go + ge
    for seqno, strand, length, s7a, s7b, s16, bp16, s63, s16s, bp16s, bq16s in rows:
        d = seqs[seqno]
        assert oracle.search7_lane(d, qs[strand], M, goe, ge) == s7a == s7b
        assert oracle.search16_lane(d, qs[strand], M, goe, ge) == (s16, bp16)
        assert oracle.search16s_lane(d, qs[strand], M, goe, ge) == (s16s, bp16s, bq16s)
        assert oracle.fullsw(d, qs[strand], M, goe, ge) == s63


@pytest.mark.parametrize("matrix,go,ge,qlen,seed", [("BLOSUM62", 11, 1, 200, 11), ("PAM70", 10, 1, 90, 12), ("BLOSUM45", 10, 3, 333, 13)])
def test_alignments_match_compiled_reference(tmp_path, matrix, go, ge, qlen, seed):
    """align() with and without the search16s hint on fresh inputs with gappy homologs"""
    rtab = synth.residue_table_protein()
    q = synth._random_residues(3000 + seed, 3, qlen, rtab)
    seqs = synth.make_db(70 + seed, 60, query=q)
    gappy = np.concatenate([q[: qlen // 3], synth._random_residues(seed, 9, 7, rtab), q[qlen // 3: qlen // 2], q[qlen // 2 + 5:]])
    seqs += [q, gappy, np.concatenate([seqs[0][:40], gappy[10:], seqs[1][:25]]), q[::-1].copy()]
    M = oracle.matrix_builtin(matrix)
    rows, arows = run_harness(tmp_path, seqs, q, True, matrix, go, ge, align=True)
    assert len(arows) > 20
    for (seqno, ds, s16s, bp, bq), plain, hinted in arows:
        assert oracle.align(q, seqs[seqno], M, go, ge) == (plain[0], plain[1], plain[2], plain[3], plain[4], plain[5])
        if hinted is not None:
            assert oracle.align(q, seqs[seqno], M, go, ge, (s16s, bq, bp)) == hinted


@pytest.mark.parametrize("symtype,qgc,dgc,seed", [(2, 11, 1, 1), (3, 1, 11, 2), (4, 4, 5, 3), (4, 1, 1, 4)])
def test_translated_lanes_match_compiled_reference(tmp_path, symtype, qgc, dgc, seed):
    """-p 2/3/4 with other genetic codes: the oracle's tables and frames against the reference's, through every
    (query frame, database frame) score"""
    ntab, rtab = synth.residue_table_nucleotide(), synth.residue_table_protein()
    q = synth._random_residues(4000 + seed, 3, 240, ntab) if symtype in (2, 4) else synth._random_residues(4000 + seed, 3, 90, rtab)
    if symtype == 2:
        seqs = synth.make_db(80 + seed, 25)
    else:
        seqs = [s[:300] for s in synth.make_db(80 + seed, 25, protein=False)]
        seqs[3][5:8] = 15
        seqs += [np.zeros(0, np.uint8), seqs[0][:2].copy(), seqs[1][:4].copy()]
    M = oracle.matrix_builtin("BLOSUM62")
    rows = run_harness(tmp_path, seqs, q, symtype == 2, "BLOSUM62", 11, 1, symtype=symtype, qgc=qgc, dgc=dgc)
    qt, dt = oracle.translate_table(qgc), oracle.translate_table(dgc)
    qf = oracle.frames(q, qt) if symtype in (2, 4) else [q]
    assert len(rows) == len(seqs) * len(qf) * (6 if symtype in (3, 4) else 1)
    for seqno, qtag, dtag, length, s7a, s7b, s16, bp16, s63, s16s, bp16s, bq16s in rows:
        d = oracle.translate(seqs[seqno], dtag // 3, dtag % 3, dt) if symtype in (3, 4) else seqs[seqno]
        assert len(d) == length
        assert oracle.search7_lane(d, qf[qtag], M, 12, 1) == s7a == s7b
        assert oracle.fullsw(d, qf[qtag], M, 12, 1) == s63
        assert oracle.search16s_lane(d, qf[qtag], M, 12, 1) == (s16s, bp16s, bq16s)


def test_reference_prints_the_same_for_a_nucleotide_database_cut_into_volumes(tmp_path):
    """the fixture the GPU test test_cli_nucleotide_database_in_three_volumes_behind_an_alias rests on: the compiled
    reference on the `nt` sequences written as three .nin/.nsq volumes behind a .nal alias prints the golden -m 8 text of
    the single-volume database"""
    import cases
    from conftest import load_golden
    exe = os.path.join(ROOT, "oracle", "_ref", "swipe")
    case, g = cases.get("nt"), load_golden("nt")
    base = str(tmp_path / "ntv")
    blastdb.write_db(base, case.seqs, protein=False, volumes=3)
    qf = str(tmp_path / "q.fa")
    open(qf, "w").write(">query test\n" + "".join(blastdb.NCBI4NA[c] for c in case.query) + "\n")
    r = subprocess.run([exe, "-d", base, "-i", qf, "-p", "0", "-G", str(case.gapopen), "-E", str(case.gapextend), "-v", str(case.keep),
                        "-e", "10", "-r", str(case.match), "-q", str(case.mismatch), "-m", "8", "-b", str(case.keep)],
                       capture_output=True, text=True, check=True)
    assert r.stdout == g["tsv"]
