"""TEST INFRASTRUCTURE: ctypes view of oracle/liboracle.so (sw_oracle.c), the CPU
restatement of the reference's hot path.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg import this package - never the product in swipe_amd/.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False) -> None:
    """Compile liboracle.so (and, when /root/reference exists, oracle/_ref/)."""
    if force or not os.path.exists(os.path.join(_HERE, "liboracle.so")) or \
            os.path.getmtime(os.path.join(_HERE, "liboracle.so")) < os.path.getmtime(os.path.join(_HERE, "sw_oracle.c")):
        subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])


def lib():
    global _LIB
    if _LIB is None:
        build()
        L = C.CDLL(os.path.join(_HERE, "liboracle.so"))
        lp, u8p, i64p = C.POINTER(C.c_long), C.c_void_p, C.c_void_p
        L.swo_matrix_builtin.argtypes = [C.c_char_p, lp]
        L.swo_matrix_builtin.restype = C.c_int
        L.swo_matrix_nucleotide.argtypes = [C.c_long, C.c_long, lp]
        L.swo_matrix_parse.argtypes = [C.c_char_p, lp]
        L.swo_matrix_parse.restype = C.c_int
        L.swo_score_limits.argtypes = [lp, lp, lp, lp, lp]
        for f in (L.swo_fullsw, L.swo_search7_lane):
            f.argtypes = [u8p, C.c_long, u8p, C.c_long, lp, C.c_ubyte, C.c_ubyte]
            f.restype = C.c_long
        L.swo_search16_lane.argtypes = [u8p, C.c_long, u8p, C.c_long, lp, C.c_ushort, C.c_ushort, lp]
        L.swo_search16_lane.restype = C.c_long
        L.swo_search16s_lane.argtypes = [u8p, C.c_long, u8p, C.c_long, lp, C.c_ushort, C.c_ushort, lp, lp]
        L.swo_search16s_lane.restype = C.c_long
        L.swo_search_chunk.argtypes = [u8p, i64p, C.c_long, u8p, C.c_long, lp, C.c_long, C.c_long, lp, C.c_void_p]
        L.swo_search_all63.argtypes = [u8p, i64p, C.c_long, u8p, C.c_long, lp, C.c_long, C.c_long, lp, C.c_int]
        L.swo_length_adjustment.argtypes = [C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, C.c_long, C.c_int, C.POINTER(C.c_int)]
        L.swo_length_adjustment.restype = C.c_int
        L.swo_stats_protein.argtypes = [C.c_char_p, C.c_long, C.c_long, C.c_void_p]
        L.swo_stats_protein.restype = C.c_int
        L.swo_stats_nucleotide.argtypes = [C.c_long, C.c_long, C.c_long, C.c_long, C.c_void_p]
        L.swo_stats_nucleotide.restype = C.c_int
        L.swo_stats_default_gaps.argtypes = [C.c_char_p, lp, lp]
        L.swo_stats_default_gaps.restype = C.c_int
        L.swo_hits_new.argtypes = [C.c_long, C.c_long, C.c_long, C.c_long, C.c_double, C.c_double, C.c_int, C.c_int,
                                   C.c_char_p, C.c_long, C.c_long, C.c_long, C.c_long, C.c_long, C.c_long, C.c_long, C.c_long]
        L.swo_hits_new.restype = C.POINTER(Hits)
        L.swo_hits_enter.argtypes = [C.POINTER(Hits), C.c_long, C.c_long, C.c_long, C.c_long, C.c_long, C.c_long]
        L.swo_hits_expect.argtypes = [C.POINTER(Hits), C.c_long]
        L.swo_hits_expect.restype = C.c_double
        L.swo_hits_bits.argtypes = [C.POINTER(Hits), C.c_long]
        L.swo_hits_bits.restype = C.c_double
        L.swo_hits_free.argtypes = [C.POINTER(Hits)]
        L.swo_align.argtypes = [u8p, C.c_long, u8p, C.c_long, lp, C.c_long, C.c_long, C.c_long, C.c_long, C.c_long,
                                C.c_void_p, C.c_char_p, C.c_long]
        L.swo_align.restype = C.c_long
        L.swo_translate_table.argtypes = [C.c_int, u8p]
        L.swo_translate_table.restype = C.c_int
        L.swo_translate.argtypes = [u8p, C.c_long, C.c_int, C.c_int, u8p, u8p]
        L.swo_translate.restype = C.c_long
        _LIB = L
    return _LIB


class Hit(C.Structure):
    _fields_ = [(n, C.c_long) for n in ("seqno", "score", "qstrand", "qframe", "dstrand", "dframe")]


class Hits(C.Structure):
    _fields_ = [(n, C.c_long) for n in ("keephits", "count", "scorethreshold", "upperscorethreshold",
                                        "init_threshold", "totalhits", "obvious")] + \
               [("stats_available", C.c_int)] + \
               [(n, C.c_double) for n in ("lam", "K", "Kmn", "logK", "lambda_d_log2", "logK_d_log2")] + \
               [(n, C.c_long) for n in ("lenadj", "m", "n")] + [("list", C.POINTER(Hit))]


class KA(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("lam", "K", "H", "alpha", "beta")]


class Counters(C.Structure):
    _fields_ = [(n, C.c_long) for n in ("compute7", "compute16", "compute63")]


def _lp(a):
    return a.ctypes.data_as(C.POINTER(C.c_long))


def matrix_builtin(name: str) -> np.ndarray:
    M = np.empty(1024, dtype=np.int64)
    if not lib().swo_matrix_builtin(name.encode(), _lp(M)):
        raise KeyError(name)
    return M


def matrix_nucleotide(match: int = 1, mismatch: int = -3) -> np.ndarray:
    M = np.empty(1024, dtype=np.int64)
    lib().swo_matrix_nucleotide(match, mismatch, _lp(M))
    return M


def matrix_parse(text: str) -> np.ndarray:
    M = np.empty(1024, dtype=np.int64)
    if not lib().swo_matrix_parse(text.encode(), _lp(M)):
        raise ValueError("Problem parsing score matrix file.")
    return M


def score_limits(M: np.ndarray):
    v = [C.c_long() for _ in range(4)]
    lib().swo_score_limits(_lp(M), *[C.byref(x) for x in v])
    return tuple(x.value for x in v)   # lo, hi, limit7, limit16


def _u8(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    return a, a.ctypes.data


def fullsw(d, q, M, goe, ge) -> int:
    d, dp = _u8(d)
    q, qp = _u8(q)
    return lib().swo_fullsw(dp, len(d), qp, len(q), _lp(M), goe & 0xFF, ge & 0xFF)


def search7_lane(d, q, M, goe, ge) -> int:
    d, dp = _u8(d)
    q, qp = _u8(q)
    return lib().swo_search7_lane(dp, len(d), qp, len(q), _lp(M), goe & 0xFF, ge & 0xFF)


def search16_lane(d, q, M, goe, ge):
    d, dp = _u8(d)
    q, qp = _u8(q)
    bp = C.c_long()
    s = lib().swo_search16_lane(dp, len(d), qp, len(q), _lp(M), goe & 0xFFFF, ge & 0xFFFF, C.byref(bp))
    return s, bp.value


def search16s_lane(d, q, M, goe, ge):
    d, dp = _u8(d)
    q, qp = _u8(q)
    bp, bq = C.c_long(), C.c_long()
    s = lib().swo_search16s_lane(dp, len(d), qp, len(q), _lp(M), goe & 0xFFFF, ge & 0xFFFF, C.byref(bp), C.byref(bq))
    return s, bp.value, bq.value


def align(q, d, M, gapopen, gapextend, hint=None):
    """align() of the reference as hits_align calls it; hint = (score, q_end, d_end) or None.
    Returns (score, q_start, d_start, q_end, d_end, cigar) or None for the internal-error case."""
    d, dp = _u8(d)
    q, qp = _u8(q)
    res = (C.c_long * 5)()
    room = 16 * (len(q) + len(d)) + 64
    buf = C.create_string_buffer(room)
    hs, hq, hd = hint if hint else (0, 0, 0)
    n = lib().swo_align(qp, len(q), dp, len(d), _lp(M), gapopen, gapextend, hs, hq, hd, C.addressof(res), buf, room)
    if n < 0:
        return None
    return res[4], res[0], res[1], res[2], res[3], buf.value.decode()


def translate_table(gencode: int) -> np.ndarray:
    t = np.zeros(4096, dtype=np.uint8)
    if not lib().swo_translate_table(gencode, t.ctypes.data):
        raise ValueError("Illegal genetic code specified.")
    return t


def translate(dna, strand: int, frame: int, table: np.ndarray) -> np.ndarray:
    """frame (0..2) of strand (0/1) of a nucleotide sequence in nibble codes -> NCBIstdaa codes"""
    dna, dp = _u8(dna)
    out = np.zeros(max(len(dna) // 3, 1), dtype=np.uint8)
    n = lib().swo_translate(dp, len(dna), strand, frame, table.ctypes.data, out.ctypes.data)
    return out[:n].copy()


def frames(dna, table: np.ndarray):
    """the six translations in the reference's order 3*strand + frame"""
    return [translate(dna, t // 3, t % 3, table) for t in range(6)]


def pack(seqs):
    """list of residue arrays -> (concatenated uint8, int64 offsets[n+1])"""
    lens = np.fromiter((len(s) for s in seqs), dtype=np.int64, count=len(seqs))
    off = np.zeros(len(seqs) + 1, dtype=np.int64)
    np.cumsum(lens, out=off[1:])
    res = np.concatenate([np.asarray(s, dtype=np.uint8) for s in seqs]) if len(seqs) else np.zeros(0, np.uint8)
    return np.ascontiguousarray(res), off


def search_chunk(residues, offsets, q, M, goe, ge):
    """Scores as the reference's escalation loop delivers them + (compute7, compute16, compute63)."""
    residues, rp = _u8(residues)
    q, qp = _u8(q)
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    n = len(offsets) - 1
    scores = np.zeros(n, dtype=np.int64)
    c = Counters()
    lib().swo_search_chunk(rp, offsets.ctypes.data, n, qp, len(q), _lp(M), goe, ge, _lp(scores), C.byref(c))
    return scores, (c.compute7, c.compute16, c.compute63)


def search_all63(residues, offsets, q, M, goe, ge, threads: int = 1) -> np.ndarray:
    residues, rp = _u8(residues)
    q, qp = _u8(q)
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    n = len(offsets) - 1
    scores = np.zeros(n, dtype=np.int64)
    lib().swo_search_all63(rp, offsets.ctypes.data, n, qp, len(q), _lp(M), goe, ge, _lp(scores), threads)
    return scores


def stats_protein(matrix: str, go: int, ge: int):
    p = KA()
    ok = lib().swo_stats_protein(matrix.encode(), go, ge, C.byref(p))
    return (p.lam, p.K, p.H, p.alpha, p.beta) if ok else None


def stats_nucleotide(match: int, mismatch: int, go: int, ge: int):
    p = KA()
    ok = lib().swo_stats_nucleotide(match, mismatch, go, ge, C.byref(p))
    return (p.lam, p.K, p.H, p.alpha, p.beta) if ok else None


def default_gaps(matrix: str):
    a, b = C.c_long(), C.c_long()
    return (a.value, b.value) if lib().swo_stats_default_gaps(matrix.encode(), C.byref(a), C.byref(b)) else None


class HitList:
    """hits_init + hits_enter + E-value/bit-score arithmetic of the reference (hits.cc)."""

    def __init__(self, *, descriptions=250, alignments=100, minscore=1, maxscore=(1 << 62), minexpect=0.0,
                 expect=10.0, symtype=1, querystrands=3, matrix="BLOSUM62", match=1, mismatch=-3,
                 gapopen=11, gapextend=1, qlen=0, dbseqs=0, dbsyms=0, effdbsize=0):
        self._h = lib().swo_hits_new(descriptions, alignments, minscore, maxscore, minexpect, expect, symtype,
                                     querystrands, matrix.encode(), match, mismatch, gapopen, gapextend,
                                     qlen, dbseqs, dbsyms, effdbsize)

    def enter(self, seqno, score, qstrand=0, qframe=0, dstrand=0, dframe=0):
        lib().swo_hits_enter(self._h, int(seqno), int(score), qstrand, qframe, dstrand, dframe)

    @property
    def c(self):
        return self._h.contents

    def hits(self):
        h = self.c
        return [(h.list[i].seqno, h.list[i].score, h.list[i].qstrand, h.list[i].dstrand) for i in range(h.count)]

    def full(self):
        """(seqno, score, qstrand, qframe, dstrand, dframe) per kept hit"""
        h = self.c
        return [(h.list[i].seqno, h.list[i].score, h.list[i].qstrand, h.list[i].qframe, h.list[i].dstrand, h.list[i].dframe)
                for i in range(h.count)]

    def expect(self, score):
        return lib().swo_hits_expect(self._h, int(score))

    def bits(self, score):
        return lib().swo_hits_bits(self._h, int(score))

    def __del__(self):
        try:
            lib().swo_hits_free(self._h)
        except Exception:
            pass
