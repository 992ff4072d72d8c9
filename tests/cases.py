"""Deterministic test cases shared by the golden-fixture generator (tests/golden/make_golden.py,
run in the build container against the compiled reference) and the parity tests (run anywhere).
A case is regenerated from code; only the reference's OUTPUTS are committed under tests/golden/.
Case list follows SURVEY.md section 8(c)."""
from __future__ import annotations

import hashlib
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

from swipe_amd import blastdb, synth

np.seterr(over="ignore")

Q375 = blastdb.encode_protein(synth.QUERY_P07327)
PLANTED_IN_100K = [10216, 13988, 28018, 33513, 34273, 48653, 50171, 51976, 59305, 73391, 80694, 96384, 98022]

ASYM_MATRIX = """# asymmetric test matrix: row = database letter, column = query letter
   A  R  N  D  C  Q  E  G
A  5 -1 -2 -2  0 -1 -1  0
R -2  6  0 -2 -3  1  0 -2
N -1  1  7  1 -3  0  0  0
D -3 -2  2  8 -3  0  2 -1
C  1 -3 -3 -3  9 -3 -4 -3
Q -1  2  0  0 -3  6  2 -2
E -1  0  0  1 -4  3  7 -2
G  1 -2  0 -1 -3 -2 -3  6
"""

# every cell 100 on the diagonal: a 750-aa self hit scores 75000 > SCORELIMIT_16
BIG_MATRIX = "   " + "  ".join("ARNDCQEGHILKMFPSTWYV") + "\n" + "\n".join(
    a + " " + " ".join("100" if a == b else "-50" for b in "ARNDCQEGHILKMFPSTWYV") for a in "ARNDCQEGHILKMFPSTWYV") + "\n"


@dataclass
class Case:
    name: str
    protein: bool
    seqs: List[np.ndarray]
    query: np.ndarray
    matrix: str = "BLOSUM62"          # builtin name, or "@text" for a custom matrix body
    matrix_text: Optional[str] = None
    gapopen: int = 11
    gapextend: int = 1
    match: int = 1
    mismatch: int = -3
    volumes: int = 1
    keep: int = 250
    extra: dict = field(default_factory=dict)
    symtype: Optional[int] = None     # reference -p: 0 nt, 1 aa, 2 translated query, 3 translated db, 4 both
    query_gencode: int = 1
    db_gencode: int = 1

    @property
    def sym(self) -> int:
        return self.symtype if self.symtype is not None else (1 if self.protein else 0)

    @property
    def query_is_nt(self) -> bool:
        return self.sym in (0, 2, 4)

    def checksum(self) -> str:
        h = hashlib.sha1()
        for s in self.seqs:
            h.update(np.asarray(s, dtype=np.uint8).tobytes())
            h.update(b"|")
        h.update(np.asarray(self.query, dtype=np.uint8).tobytes())
        return h.hexdigest()


def _rng_seq(seed, n, table):
    return synth._random_residues(synth.seq_key(seed, 7), 99, n, table)


def case_p1k() -> Case:
    ltab, rtab = synth.length_table(), synth.residue_table_protein()
    seqs = [synth.make_sequence(1, s, ltab, rtab, Q375) for s in range(1000)]
    seqs += [synth.make_sequence(1, s, ltab, rtab, Q375) for s in PLANTED_IN_100K]
    return Case("p1k", True, seqs, Q375)


def case_edges() -> Case:
    """zero-length, 1-residue, lengths 0..3 mod 4, >16 sequences ending in the same block,
    rare letters (B Z X U * O J), prefixes of the query whose self score walks across 117."""
    rtab = synth.residue_table_protein()
    seqs = [np.zeros(0, np.uint8), Q375[:1], Q375[:2], Q375[:3], Q375[:4], Q375[:5]]
    seqs += [_rng_seq(3 + k, 37, rtab) for k in range(20)]          # 20 sequences, same length
    seqs += [_rng_seq(40 + k, 60 + k, rtab) for k in range(8)]      # lengths 60..67
    seqs += [Q375[:n] for n in range(14, 40)]                       # self scores straddle SCORELIMIT_7
    seqs += [blastdb.encode_protein("BZXU*OJ-" * 5), blastdb.encode_protein("ACDEFGHIKLMNPQRSTVWYBZX*")]
    seqs += [np.zeros(0, np.uint8), Q375, Q375[::-1].copy(), np.concatenate([Q375, Q375])]
    return Case("edges", True, seqs, Q375)


def case_limit16() -> Case:
    """custom matrix with 100 on the diagonal: crosses SCORELIMIT_16 = 65436 -> fullsw."""
    rtab = synth.residue_table_protein()
    q = _rng_seq(5, 750, rtab)
    seqs = [q, q[:654], q[:655], q[:656], q[:300], _rng_seq(6, 500, rtab), q[100:], np.concatenate([q[:400], q[350:]])]
    return Case("limit16", True, seqs, q, matrix="@text", matrix_text=BIG_MATRIX, gapopen=40, gapextend=10, keep=20)


def case_asym() -> Case:
    tab = np.array([1, 16, 13, 4, 3, 15, 5, 7], dtype=np.uint8)    # A R N D C Q E G in NCBIstdaa
    def rs(seed, n):
        h = synth.splitmix64(np.arange(n, dtype=np.uint64) + np.uint64(seed * 7919))
        return tab[(h >> np.uint64(20)).astype(np.int64) % 8]
    q = rs(1, 120)
    seqs = [rs(10 + k, 50 + 7 * k) for k in range(30)] + [q, q[10:90], np.concatenate([rs(77, 20), q[30:100], rs(78, 9)])]
    return Case("asym", True, seqs, q, matrix="@text", matrix_text=ASYM_MATRIX, gapopen=6, gapextend=2, keep=40)


def case_nt() -> Case:
    """1 kb DNA query, both strands, planted plus- and minus-strand hits, ambiguity codes."""
    rtab = synth.residue_table_nucleotide()
    ltab = synth.length_table()
    q = _rng_seq(11, 1000, rtab)
    seqs = [synth.make_sequence(2, s, ltab, rtab, None) for s in range(300)]
    mut = q.copy()
    mut[::13] = rtab[(np.arange(len(mut[::13])) * 977) % 4096]
    seqs += [np.concatenate([_rng_seq(12, 40, rtab), mut[100:700], _rng_seq(13, 25, rtab)])]        # plus-strand hit
    seqs += [np.concatenate([_rng_seq(14, 33, rtab), blastdb.revcomp_nt16(mut[200:900]), _rng_seq(15, 8, rtab)])]  # minus
    amb = q[300:500].copy()
    amb[50:53] = 15      # NNN
    amb[100] = 5         # R = A|G
    seqs += [amb, np.zeros(0, np.uint8), q[:1], q[:2], q[:3], q[:4], q[:5], q[:7]]
    return Case("nt", False, seqs, q, gapopen=5, gapextend=2, match=1, mismatch=-3, keep=60)


def case_multivol() -> Case:
    c = case_p1k()
    return Case("multivol", True, c.seqs[:400], Q375, volumes=3, keep=50)


STANDARD_CODE = "FFLLSSSSYY**CC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG"   # codons in T,C,A,G order


def back_translate(protein: np.ndarray, salt: int = 0) -> np.ndarray:
    """NCBIstdaa codes -> one-hot nucleotide codes (A=1 C=2 G=4 T=8), picking among synonymous codons of the
    standard code deterministically."""
    base = [8, 2, 1, 4]     # T C A G
    out = []
    for k, aa in enumerate(protein):
        letter = blastdb.NCBISTDAA[int(aa)]
        codons = [c for c in range(64) if STANDARD_CODE[c] == letter] or [c for c in range(64) if STANDARD_CODE[c] == "A"]
        c = codons[(k * 7 + salt) % len(codons)]
        out += [base[c >> 4], base[(c >> 2) & 3], base[c & 3]]
    return np.array(out, dtype=np.uint8)


def case_blastx() -> Case:
    """-p 2: nucleotide query translated in six frames against a protein database."""
    rtab, ntab = synth.residue_table_protein(), synth.residue_table_nucleotide()
    ltab = synth.length_table()
    gene = back_translate(Q375[40:160], 1)
    other = blastdb.revcomp_nt16(back_translate(Q375[200:290], 2))
    q = np.concatenate([_rng_seq(21, 2, ntab), gene, _rng_seq(22, 31, ntab), other, _rng_seq(23, 4, ntab)])
    q[100] = 15          # N inside a codon
    q[203] = 5           # R
    seqs = [synth.make_sequence(3, s, ltab, rtab, None) for s in range(120)]
    seqs += [Q375, Q375[30:170], Q375[190:300], Q375[::-1].copy(), np.zeros(0, np.uint8), Q375[:1], Q375[50:53]]
    return Case("blastx", True, seqs, q, keep=40, symtype=2)


def case_tblastn() -> Case:
    """-p 3: protein query against a nucleotide database translated in six frames (genetic code 4 for the
    database: TGA = W)."""
    rtab, ntab = synth.residue_table_protein(), synth.residue_table_nucleotide()
    ltab = synth.length_table()
    q = Q375[:150]
    seqs = [synth.make_sequence(4, s, ltab, ntab, None) for s in range(100)]
    gene = back_translate(q[10:140], 3)
    seqs += [np.concatenate([_rng_seq(31, 7, ntab), gene, _rng_seq(32, 11, ntab)]),          # frame +2
             np.concatenate([_rng_seq(33, 3, ntab), blastdb.revcomp_nt16(gene), _rng_seq(34, 5, ntab)]),   # minus strand
             np.concatenate([gene[:150], _rng_seq(35, 1, ntab), gene[150:]])]                # frame shift inside
    amb = gene.copy()
    amb[30:33] = 15
    amb[61] = 3           # M = A|C
    seqs += [amb, np.zeros(0, np.uint8), gene[:1], gene[:2], gene[:3], gene[:4], gene[:5], gene[:6], gene[:7]]
    return Case("tblastn", False, seqs, q, keep=40, symtype=3, db_gencode=4)


def case_tblastx() -> Case:
    """-p 4: both translated (36 frame pairs); query code 1, database code 2."""
    ntab = synth.residue_table_nucleotide()
    ltab = synth.length_table()
    gene = back_translate(Q375[100:200], 5)
    q = np.concatenate([_rng_seq(41, 1, ntab), gene, _rng_seq(42, 20, ntab)])
    seqs = [synth.make_sequence(5, s, ltab, ntab, None)[:400] for s in range(40)]
    seqs += [np.concatenate([_rng_seq(43, 5, ntab), gene[30:270], _rng_seq(44, 9, ntab)]),
             blastdb.revcomp_nt16(np.concatenate([_rng_seq(45, 10, ntab), gene[:200]])), q.copy(), np.zeros(0, np.uint8), q[:4]]
    return Case("tblastx", False, seqs, q, keep=40, symtype=4, query_gencode=1, db_gencode=2)


def case_headers() -> Case:
    """Real-database header features: every Seq-id flavour the reference renders, several
    definition lines per entry, missing titles, taxids / membership bits / links, an OID mask behind an alias
    (as NCBI ships swissprot) and a taxid list."""
    ltab, rtab = synth.length_table(), synth.residue_table_protein()
    seqs = [synth.make_sequence(6, s, ltab, rtab, None)[:300] for s in range(48)]
    frag = [Q375[20 * k: 20 * k + 150] for k in range(12)]
    seqs += frag
    n = len(seqs)
    flavours = [
        lambda i: [("gi", 1000 + i), ("sp", "P%05d" % i, "PROT%d_HUMAN" % i, 2)],
        lambda i: [("gi", 2000 + i), ("ref", "NP_%06d" % i, "", 1)],
        lambda i: [("gb", "AAA%05d" % i, "", 3)],
        lambda i: [("emb", "CAA%05d" % i, "LOCUS%d" % i)],
        lambda i: [("pdb", "1AB%d" % (i % 10), 65 + i % 26)],
        lambda i: [("pdb", "2XY%d" % (i % 10), 97 + i % 26)],      # lower-case chain -> doubled upper case
        lambda i: [("gnl", "mydb", "tag%d" % i)],
        lambda i: [("gnl", "BL_ORD_ID", i)],
        lambda i: [("lcl", i)],
        lambda i: [("pat", "US", "650%04d" % i, i % 7 + 1, True)],
        lambda i: [("pat", "EP", "11%04d" % i, 2, False)],
        lambda i: [("sp", "Q%05d" % i, "UNREV%d" % i, 0, "unreviewed")],
        lambda i: [("bbs", 70000 + i)],
        lambda i: [("gim", 5 + i)],
        lambda i: [("dbj", "BAA%05d" % i, "", 1), ("gi", 3000 + i)],
        lambda i: [("pir", "", "S%05d" % i)],
    ]
    headers = []
    for i in range(n):
        ids = flavours[i % len(flavours)](i)
        d = dict(title=None if i % 11 == 5 else "protein number %d with a description long enough to be cut in the hit list and wrapped above the alignment, repeated: protein number %d" % (i, i) if i % 5 == 0 else "protein %d" % i,
                 ids=ids, taxid=9600 + i % 7, memb=(1 if i % 2 == 0 else 2) if i % 3 else None, links=4 if i % 13 == 0 else None)
        entry = [d]
        if i % 4 == 1:        # a second, merged definition line with another taxid and membership
            entry.append(dict(title="identical twin of %d" % i, ids=[("gi", 9000 + i), ("gb", "TWIN%04d" % i, "", 1)],
                              taxid=9700 + i % 3, memb=1))
        headers.append(entry)
    include = [any(((d.get("memb") or 0) & 1) == 1 for d in h) for h in headers]
    extra = dict(headers=headers, include=include, taxids=[9601, 9603, 9700, 12345])
    return Case("headers", True, seqs, Q375, keep=30, extra=extra)


ALL = [case_p1k, case_edges, case_limit16, case_asym, case_nt, case_multivol]
TRANSLATED = [case_blastx, case_tblastn, case_tblastx]


def get(name: str) -> Case:
    for f in ALL + TRANSLATED + [case_headers]:
        if f.__name__ == "case_" + name:
            return f()
    raise KeyError(name)
