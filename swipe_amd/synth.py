"""Deterministic synthetic sequence databases (counter-based, integer-only).

The benchmark database of BASELINE.json (10 M proteins, lengths ~ lognormal(5.6, 0.6)
clipped to [10, 35000], Robinson-Robinson residue background, ~0.01 % planted query
homologs) must be producible on the GPU box without shipping gigabytes, and the small
test databases must be *the same function* at smaller ``nseq``.  Everything is a pure
function of ``(seed, seqno, position)`` through splitmix64, so the numpy implementation
here and the multi-threaded C++ one in ``csrc/synth.cpp`` (which receives the two
quantile tables computed here) are bit-identical; ``tests/test_synth.py`` checks that.
"""
from __future__ import annotations

import numpy as np

from .blastdb import NCBISTDAA

MASK64 = (1 << 64) - 1
GOLDEN = 0x9E3779B97F4A7C15
MIX_SEQ = 0xD1B54A32D192ED03
PLANT_PERIOD = 8192          # one sequence in 8192 is a planted homolog (~0.012 %)

# Robinson & Robinson (1991) amino-acid background frequencies
_RR = {"A": 0.07805, "R": 0.05129, "N": 0.04487, "D": 0.05364, "C": 0.01925, "Q": 0.04264,
       "E": 0.06295, "G": 0.07377, "H": 0.02199, "I": 0.05142, "L": 0.09019, "K": 0.05744,
       "M": 0.02243, "F": 0.03856, "P": 0.05203, "S": 0.07120, "T": 0.05841, "W": 0.01330,
       "Y": 0.03216, "V": 0.06441}


def splitmix64(x):
    """splitmix64 finaliser; works on Python ints and on numpy uint64 arrays."""
    if isinstance(x, np.ndarray):
        with np.errstate(over="ignore"):
            z = x + np.uint64(GOLDEN)
            z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
            return z ^ (z >> np.uint64(31))
    z = (x + GOLDEN) & MASK64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK64
    return z ^ (z >> 31)


def length_table(mu: float = 5.6, sigma: float = 0.6, lo: int = 10, hi: int = 35000) -> np.ndarray:
    """4096 quantiles of lognormal(mu, sigma), clipped to [lo, hi] (int32)."""
    from scipy.special import ndtri
    p = (np.arange(4096, dtype=np.float64) + 0.5) / 4096.0
    q = np.exp(mu + sigma * ndtri(p))
    return np.clip(np.rint(q), lo, hi).astype(np.int32)


def residue_table_protein() -> np.ndarray:
    """4096-entry table: uniform 12-bit draw -> NCBIstdaa code with R-R frequencies."""
    letters = list(_RR)
    w = np.array([_RR[c] for c in letters], dtype=np.float64)
    edges = np.rint(np.cumsum(w / w.sum()) * 4096).astype(np.int64)
    tab = np.empty(4096, dtype=np.uint8)
    start = 0
    for c, e in zip(letters, edges):
        tab[start:e] = NCBISTDAA.index(c)
        start = e
    tab[start:] = NCBISTDAA.index(letters[-1])
    return tab


def residue_table_nucleotide() -> np.ndarray:
    """4096-entry table -> one-hot base masks 1,2,4,8 (uniform)."""
    return np.repeat(np.array([1, 2, 4, 8], dtype=np.uint8), 1024)


def seq_key(seed: int, seqno: int) -> int:
    return splitmix64(splitmix64(seed & MASK64) ^ ((seqno * MIX_SEQ) & MASK64))


def _random_residues(key: int, salt: int, n: int, rtab: np.ndarray) -> np.ndarray:
    """n residues from the stream keyed (key, salt): 5 twelve-bit draws per hash."""
    nh = (n + 4) // 5
    ctr = (np.arange(nh, dtype=np.uint64) + np.uint64((key + salt) & MASK64))
    h = splitmix64(ctr)
    draws = np.stack([(h >> np.uint64(12 * k)) & np.uint64(4095) for k in range(5)], axis=1).reshape(-1)
    return rtab[draws[:n].astype(np.int64)]


def make_sequence(seed: int, seqno: int, ltab: np.ndarray, rtab: np.ndarray,
                  query: np.ndarray | None = None) -> np.ndarray:
    """Residue codes of synthetic sequence ``seqno`` (see module docstring)."""
    key = seq_key(seed, seqno)
    if query is not None and len(query) and key % PLANT_PERIOD == 0:
        kind = (key >> 13) & 7
        copies, rate = (3, 5) if kind == 0 else (1, kind * 20)      # rate in 1/256 per position
        left = int((key >> 20) & 63)
        right = int((key >> 28) & 63)
        parts = [_random_residues(key, 1, left, rtab)]
        qlen = len(query)
        for c in range(copies):
            idx = np.arange(qlen, dtype=np.uint64) + np.uint64((key + (2 + c) * (1 << 24)) & MASK64)
            h = splitmix64(idx)
            sub = (h & np.uint64(0xFF)).astype(np.int64) < rate
            repl = rtab[((h >> np.uint64(20)) & np.uint64(4095)).astype(np.int64)]
            ev = ((h >> np.uint64(40)) & np.uint64(0x3FF)).astype(np.int64)
            body = np.where(sub, repl, query).astype(np.uint8)
            keep = ev >= 6                       # ~0.6 % deletions
            ins = (ev >= 6) & (ev < 12)          # ~0.6 % insertions after the residue
            insres = rtab[((h >> np.uint64(50)) & np.uint64(4095)).astype(np.int64)]
            out = np.empty(2 * qlen, dtype=np.uint8)
            pos = np.cumsum(keep.astype(np.int64) + ins.astype(np.int64)) - (keep.astype(np.int64) + ins.astype(np.int64))
            out[pos[keep]] = body[keep]
            out[pos[ins] + 1] = insres[ins]
            parts.append(out[: int(keep.sum() + ins.sum())])
        parts.append(_random_residues(key, 1 << 30, right, rtab))
        return np.concatenate(parts)
    n = int(ltab[(key >> 40) & 4095])
    return _random_residues(key, 1, n, rtab)


def make_db(seed: int, nseq: int, *, protein: bool = True, query: np.ndarray | None = None,
            first: int = 0, ltab: np.ndarray | None = None):
    """List of residue arrays for seqnos [first, first + nseq)."""
    ltab = length_table() if ltab is None else ltab
    rtab = residue_table_protein() if protein else residue_table_nucleotide()
    return [make_sequence(seed, s, ltab, rtab, query) for s in range(first, first + nseq)]


# fixed 375-aa ADH1A-like query used throughout (BASELINE.json "375-aa query (P07327)")
QUERY_P07327 = (
    "MSTAGKVIKCKAAVLWELKKPFSIEEVEVAPPKAHEVRIKMVAVGICGTDDHVVSGTMVTPLPVILGHEAAGIVESVGEGVTTVKPGDKVIPL"
    "AIPQCGKCRICKNPESNYCLKNDVSNPQGTLQDGTSRFTCRRKPIHHFLGISTFSQYTVVDENAVAKIDAASPLEKVCLIGCGFSTGYGSAVN"
    "VAKVTPGSTCAVFGLGGVGLSAIMGCKAAGAARIIAVDINKDKFAKAKELGATECINPQDYKKPIQEVLKEMTDGGVDFSFEVIGRLDTMMASL"
    "LCCHEACGTSVIVGVPPDSQNLSMNPMLLLTGRTWKGAILGGFKSKECVPKLVADFMAKKFSLDALITHVLPFEKINEGFDLLHSGKSIRTILMF"
)
