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


def run_harness(tmp_path, seqs, query, protein, matrix, go, ge, match=1, mismatch=-3):
    base = str(tmp_path / "db")
    blastdb.write_db(base, seqs, protein=protein)
    alpha = blastdb.NCBISTDAA if protein else blastdb.NCBI4NA
    qf = str(tmp_path / "q.fa")
    open(qf, "w").write(">q\n" + "".join(alpha[c] for c in query) + "\n")
    out = subprocess.run([HARNESS, base, qf, "1" if protein else "0", matrix if protein else "-", str(go), str(ge),
                          str(match), str(mismatch)], capture_output=True, text=True, check=True).stdout
    return [list(map(int, l.split())) for l in out.splitlines() if not l.startswith("#")]


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
