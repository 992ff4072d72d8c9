"""Host-side mirror of the reference's interface around the hot path, over the C ABI.

  reference                                   here
  db_open / db_mapsequences                   Database.open / Database.from_sequences
  score_matrix_init (+ gap penalties)         Database.set_scoring, matrix_builtin, ...
  search_chunk -> hits_enter per sequence     Database.search (all scores) / search_topk
  hits_init thresholds, E-value, bit score    stats_init -> Stats.evalue / .bits
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import numpy as np

from . import _lib


class SwaError(RuntimeError):
    pass


def _check(rc: int):
    if rc != 0:
        raise SwaError(f"[{rc}] {_lib.load().swa_last_error().decode()}")


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def matrix_builtin(name: str) -> np.ndarray:
    M = np.empty(1024, dtype=np.int64)
    _check(_lib.load().swa_matrix_builtin(name.encode(), M.ctypes.data))
    return M


def matrix_nucleotide(match: int = 1, mismatch: int = -3) -> np.ndarray:
    M = np.empty(1024, dtype=np.int64)
    _check(_lib.load().swa_matrix_nucleotide(match, mismatch, M.ctypes.data))
    return M


def matrix_parse(text: str) -> np.ndarray:
    M = np.empty(1024, dtype=np.int64)
    _check(_lib.load().swa_matrix_parse(text.encode(), M.ctypes.data))
    return M


def default_gaps(matrix: str):
    a, b = C.c_int64(), C.c_int64()
    _check(_lib.load().swa_default_gaps(matrix.encode(), C.byref(a), C.byref(b)))
    return a.value, b.value


class Stats:
    def __init__(self, raw):
        self.raw = raw
        for f, _ in raw._fields_:
            setattr(self, f, getattr(raw, f))

    def evalue(self, score: int) -> float:
        return _lib.load().swa_evalue(C.byref(self.raw), int(score))

    def bits(self, score: int) -> float:
        return _lib.load().swa_bits(C.byref(self.raw), int(score))


def stats_init(*, symtype=1, matrix="BLOSUM62", match=1, mismatch=-3, gapopen=11, gapextend=1, qlen=0,
               db_seqcount=0, db_symcount=0, effdbsize=0, minscore=1, maxscore=(1 << 62), minexpect=0.0,
               expect=10.0) -> Stats:
    s = _lib.Stats()
    _check(_lib.load().swa_stats_init(symtype, matrix.encode(), match, mismatch, gapopen, gapextend, qlen,
                                      db_seqcount, db_symcount, effdbsize, minscore, maxscore, minexpect, expect,
                                      C.byref(s)))
    return Stats(s)


def merge_hits(lists: Sequence[Sequence[tuple]], keep: int):
    """Global top-`keep` of per-shard ordered hit lists (score desc, seqno desc)."""
    n = len(lists)
    stride = max((len(l) for l in lists), default=0)
    buf = (_lib.Hit * max(1, n * stride))()
    counts = (C.c_int64 * max(1, n))()
    for i, l in enumerate(lists):
        counts[i] = len(l)
        for j, (seqno, score) in enumerate(l):
            buf[i * stride + j] = _lib.Hit(int(seqno), int(score))
    out = (_lib.Hit * max(1, keep))()
    nout = C.c_int64()
    _check(_lib.load().swa_hits_merge(buf, counts, n, stride, keep, out, C.byref(nout)))
    return [(out[i].seqno, out[i].score) for i in range(nout.value)]


def merge_hit_arrays(lists: np.ndarray, counts: np.ndarray, keep: int) -> np.ndarray:
    """merge_hits on arrays: lists int64 [nlists, stride, 2] of (seqno, score) rows, counts int64 [nlists];
    returns int64 [n, 2].  No Python-level loop: the buffers go straight to swa_hits_merge."""
    lists = np.ascontiguousarray(lists, dtype=np.int64)
    counts = np.ascontiguousarray(counts, dtype=np.int64)
    n, stride = lists.shape[0], lists.shape[1]
    out = np.empty((max(1, keep), 2), dtype=np.int64)
    nout = C.c_int64()
    _check(_lib.load().swa_hits_merge(C.cast(lists.ctypes.data, C.POINTER(_lib.Hit)), C.cast(counts.ctypes.data, C.POINTER(C.c_int64)),
                                      n, stride, keep, C.cast(out.ctypes.data, C.POINTER(_lib.Hit)), C.byref(nout)))
    return out[: nout.value]


def merge_frame_hits(lists: Sequence[Sequence[tuple]], keep: int):
    """merge_hits for (seqno, score, qstrand, qframe, dstrand, dframe) tuples"""
    n = len(lists)
    stride = max((len(l) for l in lists), default=0)
    buf = (_lib.FrameHit * max(1, n * stride))()
    counts = (C.c_int64 * max(1, n))()
    for i, l in enumerate(lists):
        counts[i] = len(l)
        for j, h in enumerate(l):
            buf[i * stride + j] = _lib.FrameHit(*[int(x) for x in h])
    out = (_lib.FrameHit * max(1, keep))()
    nout = C.c_int64()
    _check(_lib.load().swa_fhits_merge(buf, counts, n, stride, keep, out, C.byref(nout)))
    return [(h.seqno, h.score, h.qstrand, h.qframe, h.dstrand, h.dframe) for h in out[: nout.value]]


def redzones_check():
    """(allocations checked, guard bytes written, report) - needs SWA_REDZONES=1 in the environment before the library is
    first used: every device allocation then has guard regions around it (swa_redzones_check)."""
    n, bad = C.c_int64(), C.c_int64()
    rep = C.create_string_buffer(4096)
    _check(_lib.load().swa_redzones_check(C.byref(n), C.byref(bad), rep, 4096))
    return n.value, bad.value, rep.value.decode()


class Database:
    """One database shard resident in the HBM of one MI355X."""

    def __init__(self, handle):
        self._h = handle

    @classmethod
    def open(cls, basename: str, *, symtype: int = 1, device: int = 0, first_seqno: int = 0, last_seqno: int = -1,
             hbm_budget: int = 0, wait: bool = True):
        """hbm_budget > 0 (bytes): volumes that may not be resident are streamed (swa_db_open_streamed, see from_arrays).
        wait=False: swa_db_open_async - the call returns once the index is read and the sequence files stream into HBM
        behind it; search / search_topk start on the parts that have arrived, everything else waits (see wait())."""
        h = C.c_void_p()
        if hbm_budget > 0:
            _check(_lib.load().swa_db_open_streamed(os.fsencode(basename), symtype, device, first_seqno, last_seqno,
                                                    hbm_budget, C.byref(h)))
        elif not wait:
            _check(_lib.load().swa_db_open_async(os.fsencode(basename), symtype, device, first_seqno, last_seqno, C.byref(h)))
        else:
            _check(_lib.load().swa_db_open(os.fsencode(basename), symtype, device, first_seqno, last_seqno, C.byref(h)))
        return cls(h)

    def wait(self):
        """Blocks until a shard opened with wait=False is resident; raises what the load ran into."""
        _check(_lib.load().swa_db_wait(self._h))

    def load_progress(self) -> dict:
        """bytes of sequence file handed to the copy engine / in all, parts searchable / in all (zeros once resident)."""
        b, t = C.c_int64(), C.c_int64()
        r, n = C.c_int32(), C.c_int32()
        _check(_lib.load().swa_db_load_progress(self._h, C.byref(b), C.byref(t), C.byref(r), C.byref(n)))
        return {"bytes_loaded": b.value, "bytes_total": t.value, "parts_ready": r.value, "parts_total": n.value}

    @classmethod
    def open_translated(cls, basename: str, *, db_gencode: int = 1, device: int = 0, first_seqno: int = 0,
                        last_seqno: int = -1):
        """A nucleotide database held as its six translations (reference -p 3 / -p 4)."""
        h = C.c_void_p()
        _check(_lib.load().swa_db_open_translated(os.fsencode(basename), db_gencode, device, first_seqno, last_seqno,
                                                  C.byref(h)))
        return cls(h)

    @classmethod
    def from_arrays(cls, residues: np.ndarray, offsets: np.ndarray, *, symtype: int = 1, device: int = 0,
                    first_seqno: int = 0, total_seqcount: int = 0, total_symcount: int = 0,
                    translate_gencode: Optional[int] = None, hbm_budget: int = 0):
        """translate_gencode: residues are nucleotides to be held as their six translations under that code.
        hbm_budget > 0 (bytes): a database that may not be resident is streamed through two device slots
        (swa_db_from_memory_streamed); search and search_topk work on it as on a resident shard."""
        residues = np.ascontiguousarray(residues, dtype=np.uint8)
        offsets = _i64(offsets)
        h = C.c_void_p()
        if hbm_budget > 0 and translate_gencode is not None:
            raise SwaError("a streamed shard (hbm_budget) cannot be a translated one (translate_gencode): translated shards are resident")
        if hbm_budget > 0:
            _check(_lib.load().swa_db_from_memory_streamed(residues.ctypes.data, offsets.ctypes.data, len(offsets) - 1, symtype,
                                                           device, first_seqno, total_seqcount, total_symcount, hbm_budget,
                                                           C.byref(h)))
        elif translate_gencode is not None:
            _check(_lib.load().swa_db_from_memory_translated(residues.ctypes.data, offsets.ctypes.data, len(offsets) - 1,
                                                             translate_gencode, device, first_seqno, total_seqcount,
                                                             total_symcount, C.byref(h)))
        else:
            _check(_lib.load().swa_db_from_memory(residues.ctypes.data, offsets.ctypes.data, len(offsets) - 1, symtype,
                                                  device, first_seqno, total_seqcount, total_symcount, C.byref(h)))
        return cls(h)

    @classmethod
    def from_sequences(cls, seqs, **kw):
        lens = np.fromiter((len(s) for s in seqs), dtype=np.int64, count=len(seqs))
        off = np.zeros(len(seqs) + 1, dtype=np.int64)
        np.cumsum(lens, out=off[1:])
        res = np.concatenate([np.asarray(s, dtype=np.uint8) for s in seqs]) if len(seqs) else np.zeros(0, np.uint8)
        return cls.from_arrays(res, off, **kw)

    def set_inclusion(self, include=None):
        """Search only the sequences with include[i] != 0 (i counted from the shard's first sequence); None = all.
        Excluded sequences report score -1."""
        if include is None:
            _check(_lib.load().swa_db_set_inclusion(self._h, None, 0))
            return
        inc = np.ascontiguousarray(include, dtype=np.uint8)
        _check(_lib.load().swa_db_set_inclusion(self._h, inc.ctypes.data, len(inc)))

    def info(self):
        i = _lib.DbInfo()
        _check(_lib.load().swa_db_info(self._h, C.byref(i)))
        return {f: getattr(i, f) for f, _ in i._fields_}

    def set_scoring(self, matrix: np.ndarray, gapopen: int, gapextend: int):
        """gapopen/gapextend as on the reference command line; the kernels get open+extend (swipe.cc:1126)."""
        M = _i64(matrix)
        _check(_lib.load().swa_set_scoring(self._h, M.ctypes.data, gapopen + gapextend, gapextend))

    def set_option(self, key: str, value=None):
        """swa_set_option: a tuning / test knob of this handle ("bound", "lanes", ...); None restores the default."""
        _check(_lib.load().swa_set_option(self._h, key.encode(), None if value is None else str(value).encode()))

    def search(self, query: np.ndarray, *, want_scores: bool = True):
        q = np.ascontiguousarray(query, dtype=np.uint8)
        c = _lib.Counters()
        i = self.info()
        scores = np.empty(i["seqcount"] * i["frames"], dtype=np.int64) if want_scores else None
        _check(_lib.load().swa_search(self._h, q.ctypes.data, len(q), scores.ctypes.data if want_scores else None,
                                      C.byref(c)))
        return scores, {f: getattr(c, f) for f, _ in c._fields_}

    def search_topk(self, query: np.ndarray, keep: int = 250, minscore: int = 1, maxscore: int = (1 << 62)):
        q = np.ascontiguousarray(query, dtype=np.uint8)
        c = _lib.Counters()
        hits = (_lib.Hit * max(1, keep))()
        n, tot, obv = C.c_int64(), C.c_int64(), C.c_int64()
        _check(_lib.load().swa_search_topk(self._h, q.ctypes.data, len(q), keep, minscore, maxscore, hits,
                                           C.byref(n), C.byref(tot), C.byref(obv), C.byref(c)))
        return ([(hits[i].seqno, hits[i].score) for i in range(n.value)], tot.value, obv.value,
                {f: getattr(c, f) for f, _ in c._fields_})

    def search_topk_array(self, query: np.ndarray, keep: int = 250, minscore: int = 1, maxscore: int = (1 << 62)):
        """search_topk with the hits as one int64 [n, 2] array of (seqno, score) rows (the layout of swa_hit_t):
        what the multi-GPU gather exchanges."""
        q = np.ascontiguousarray(query, dtype=np.uint8)
        c = _lib.Counters()
        hits = np.empty((max(1, keep), 2), dtype=np.int64)
        n, tot, obv = C.c_int64(), C.c_int64(), C.c_int64()
        _check(_lib.load().swa_search_topk(self._h, q.ctypes.data, len(q), keep, minscore, maxscore,
                                           C.cast(hits.ctypes.data, C.POINTER(_lib.Hit)), C.byref(n), C.byref(tot),
                                           C.byref(obv), C.byref(c)))
        return hits[: n.value], tot.value, obv.value, {f: getattr(c, f) for f, _ in c._fields_}

    def search2(self, query1: np.ndarray, query2: np.ndarray, *, want_scores: bool = True):
        """Two equal-length queries in one pass (nucleotide: plus strand and its reverse complement)."""
        q1 = np.ascontiguousarray(query1, dtype=np.uint8)
        q2 = np.ascontiguousarray(query2, dtype=np.uint8)
        if len(q1) != len(q2):
            raise SwaError("search2 needs two queries of equal length")
        c = _lib.Counters()
        n = self.info()["seqcount"] * self.info()["frames"]
        s1 = np.empty(n, dtype=np.int64) if want_scores else None
        s2 = np.empty(n, dtype=np.int64) if want_scores else None
        _check(_lib.load().swa_search2(self._h, q1.ctypes.data, q2.ctypes.data, len(q1),
                                       s1.ctypes.data if want_scores else None,
                                       s2.ctypes.data if want_scores else None, C.byref(c)))
        return s1, s2, {f: getattr(c, f) for f, _ in c._fields_}

    def search2_topk(self, query1, query2, keep: int = 250, minscore: int = 1, maxscore: int = (1 << 62)):
        q1 = np.ascontiguousarray(query1, dtype=np.uint8)
        q2 = np.ascontiguousarray(query2, dtype=np.uint8)
        if len(q1) != len(q2):
            raise SwaError("search2_topk needs two queries of equal length")
        c = _lib.Counters()
        hits = (_lib.Hit * max(1, keep))()
        which = (C.c_int32 * max(1, keep))()
        n, tot, obv = C.c_int64(), C.c_int64(), C.c_int64()
        _check(_lib.load().swa_search2_topk(self._h, q1.ctypes.data, q2.ctypes.data, len(q1), keep, minscore, maxscore,
                                            hits, which, C.byref(n), C.byref(tot), C.byref(obv), C.byref(c)))
        return ([(hits[i].seqno, hits[i].score, which[i]) for i in range(n.value)], tot.value, obv.value,
                {f: getattr(c, f) for f, _ in c._fields_})

    def search_pair_topk(self, query1, query2, keep=250, minscore=(1, 1), maxscore=((1 << 62), (1 << 62))):
        """Two different queries (lengths may differ) in one pass: ((hits1, total1, obvious1), (hits2, total2, obvious2),
        counters) - each part equal to search_topk of that query.  keep / minscore / maxscore: one value or a pair."""
        q1 = np.ascontiguousarray(query1, dtype=np.uint8)
        q2 = np.ascontiguousarray(query2, dtype=np.uint8)
        two = lambda v: tuple(v) if isinstance(v, (tuple, list)) else (v, v)
        k, lo, hi = two(keep), two(minscore), two(maxscore)
        c = _lib.Counters()
        h1, h2 = (_lib.Hit * max(1, k[0]))(), (_lib.Hit * max(1, k[1]))()
        n1, t1, o1, n2, t2, o2 = (C.c_int64() for _ in range(6))
        _check(_lib.load().swa_search_pair_topk(self._h, q1.ctypes.data, len(q1), q2.ctypes.data, len(q2), k[0], lo[0], hi[0],
                                                k[1], lo[1], hi[1], h1, C.byref(n1), C.byref(t1), C.byref(o1), h2, C.byref(n2),
                                                C.byref(t2), C.byref(o2), C.byref(c)))
        return (([(h1[i].seqno, h1[i].score) for i in range(n1.value)], t1.value, o1.value),
                ([(h2[i].seqno, h2[i].score) for i in range(n2.value)], t2.value, o2.value),
                {f: getattr(c, f) for f, _ in c._fields_})

    def search_frames_topk(self, queries, qtags=None, *, keep: int = 250, minscore: int = 1, maxscore: int = (1 << 62)):
        """Up to six query frames against every frame the shard holds, one merged hit list in the reference's
        order: ([(seqno, score, qstrand, qframe, dstrand, dframe)], totalhits, obvious, counters)."""
        L = _lib.load()
        qs = [np.ascontiguousarray(q, dtype=np.uint8) for q in queries]
        ptr = (C.c_void_p * len(qs))(*[q.ctypes.data for q in qs])
        lens = (C.c_int64 * len(qs))(*[len(q) for q in qs])
        tags = (C.c_int32 * len(qs))(*(qtags if qtags is not None else range(len(qs))))
        hits = (_lib.FrameHit * max(keep, 1))()
        n, tot, obv = C.c_int64(), C.c_int64(), C.c_int64()
        c = _lib.Counters()
        _check(L.swa_search_frames_topk(self._h, len(qs), ptr, lens, tags, keep, minscore, maxscore, hits, C.byref(n),
                                        C.byref(tot), C.byref(obv), C.byref(c)))
        return ([(h.seqno, h.score, h.qstrand, h.qframe, h.dstrand, h.dframe) for h in hits[: n.value]], tot.value,
                obv.value, {f: getattr(c, f) for f, _ in c._fields_})

    def search_endpoints(self, query: np.ndarray, seqnos, dstrands=None, dframes=None):
        """(score, bestpos, bestq) per listed sequence - the reference's search16s for the alignment phase.
        dstrands[i] = 1: against the reverse complement of that (nucleotide) sequence; translated shards take
        (dstrand, dframe)."""
        q = np.ascontiguousarray(query, dtype=np.uint8)
        ids = _i64(seqnos)
        out = [np.empty(len(ids), dtype=np.int64) for _ in range(3)]
        ds = None if dstrands is None else np.ascontiguousarray(dstrands, dtype=np.int32)
        df = None if dframes is None else np.ascontiguousarray(dframes, dtype=np.int32)
        _check(_lib.load().swa_search_endpoints_strand(self._h, q.ctypes.data, len(q), ids.ctypes.data,
                                                       None if ds is None else ds.ctypes.data,
                                                       None if df is None else df.ctypes.data, len(ids),
                                                       out[0].ctypes.data, out[1].ctypes.data, out[2].ctypes.data))
        return out

    def sequence(self, seqno: int, dstrand: int = 0, dframe: int = 0) -> np.ndarray:
        """db_getsequence: residues of one sequence of the shard (reverse-complemented for dstrand 1 of a
        nucleotide shard; the chosen translation of a translated shard)."""
        L = _lib.load()
        n = C.c_int64()
        rc = L.swa_db_sequence(self._h, seqno, dstrand, dframe, None, 0, C.byref(n), None)
        if rc not in (0, _lib.SWA_ERANGE):
            _check(rc)
        buf = np.empty(max(n.value, 1), dtype=np.uint8)
        _check(L.swa_db_sequence(self._h, seqno, dstrand, dframe, buf.ctypes.data, n.value, C.byref(n), None))
        return buf[: n.value]

    def align(self, query: np.ndarray, seqnos, dstrands=None, dframes=None):
        """The reference's alignment phase (align_chunk + hits_align + align) for the listed hits: a list of
        dicts with score, 0-based inclusive q_start/q_end/d_start/d_end, the edit script ("M..D..I.."),
        identities, positives, indels, aligned, gaps, dlen, and whether the GPU end point was used."""
        L = _lib.load()
        q = np.ascontiguousarray(query, dtype=np.uint8)
        ids = _i64(seqnos)
        ds = None if dstrands is None else np.ascontiguousarray(dstrands, dtype=np.int32)
        df = None if dframes is None else np.ascontiguousarray(dframes, dtype=np.int32)
        out = (_lib.Alignment * max(len(ids), 1))()
        cap = 1 << 16
        while True:
            text = C.create_string_buffer(cap)
            used = C.c_int64()
            rc = L.swa_align_hits(self._h, q.ctypes.data, len(q), ids.ctypes.data, None if ds is None else ds.ctypes.data,
                                  None if df is None else df.ctypes.data, len(ids), out, text, cap, C.byref(used))
            if rc == _lib.SWA_ERANGE:
                cap = used.value
                continue
            _check(rc)
            break
        return [_alignment_dict(out[i], text.raw) for i in range(len(ids))]

    def close(self):
        if self._h:
            _lib.load().swa_db_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Group(Database):
    """One database over several devices behind one handle (swa_group): N residue-balanced shards, one host thread per
    shard inside the library, per-shard top-K merged on the host with the reference comparator - what SWIPE's -a N
    threads / mpiswipe's workers do (swipe.cc:1599-1699, 1951-1974).  devices may repeat a device (two shards on one GPU).
    Same search / align interface as Database; the per-shard entry points that make no sense on a group raise."""
    _F = {"set_scoring": "swa_group_set_scoring", "set_option": "swa_group_set_option", "info": "swa_group_info"}

    @classmethod
    def open(cls, basename: str, *, symtype: int = 1, devices=(0,), db_gencode: int = 0, hbm_budget: int = 0):
        """hbm_budget > 0 (bytes PER DEVICE): shards that may not be resident walk their parts through two device slots."""
        h = C.c_void_p()
        dev = (C.c_int * len(devices))(*devices)
        if hbm_budget > 0:
            if db_gencode:
                raise SwaError("translated shards are resident: no hbm_budget with db_gencode")
            _check(_lib.load().swa_group_open_streamed(os.fsencode(basename), symtype, len(devices), dev, hbm_budget, C.byref(h)))
        else:
            _check(_lib.load().swa_group_open(os.fsencode(basename), symtype, db_gencode, len(devices), dev, C.byref(h)))
        return cls(h)

    @classmethod
    def from_arrays(cls, residues, offsets, *, symtype: int = 1, devices=(0,), first_seqno: int = 0, total_seqcount: int = 0,
                    total_symcount: int = 0, translate_gencode: Optional[int] = None):
        residues = np.ascontiguousarray(residues, dtype=np.uint8)
        offsets = _i64(offsets)
        h = C.c_void_p()
        dev = (C.c_int * len(devices))(*devices)
        _check(_lib.load().swa_group_from_memory(residues.ctypes.data, offsets.ctypes.data, len(offsets) - 1, symtype,
                                                 translate_gencode or 0, len(devices), dev, first_seqno, total_seqcount,
                                                 total_symcount, C.byref(h)))
        return cls(h)

    @classmethod
    def open_translated(cls, basename: str, *, db_gencode: int = 1, devices=(0,)):
        """A nucleotide database held as its six translations, over several devices (swa_group_open with a genetic code)."""
        return cls.open(basename, symtype=0, devices=devices, db_gencode=db_gencode)

    @classmethod
    def from_sequences(cls, seqs, *, devices=(0,), **kw):
        if "device" in kw:
            raise SwaError("a Group takes devices=(...), one per shard")
        lens = np.fromiter((len(s) for s in seqs), dtype=np.int64, count=len(seqs))
        off = np.zeros(len(seqs) + 1, dtype=np.int64)
        np.cumsum(lens, out=off[1:])
        res = np.concatenate([np.asarray(s, dtype=np.uint8) for s in seqs]) if len(seqs) else np.zeros(0, np.uint8)
        return cls.from_arrays(res, off, devices=devices, **kw)

    def wait(self):
        """swa_group_open streams every shard in behind the call: block until all are resident (a load error - a truncated
        .psq, a residue code out of range - is raised here rather than by the first search)"""
        _check(_lib.load().swa_group_wait(self._h))

    def load_progress(self):
        a, b, r, t = C.c_int64(), C.c_int64(), C.c_int32(), C.c_int32()
        _check(_lib.load().swa_group_load_progress(self._h, C.byref(a), C.byref(b), C.byref(r), C.byref(t)))
        return {"bytes_loaded": a.value, "bytes_total": b.value, "parts_ready": r.value, "parts_total": t.value}

    def info(self):
        i, n = _lib.DbInfo(), C.c_int()
        _check(_lib.load().swa_group_info(self._h, C.byref(i), C.byref(n)))
        return dict({f: getattr(i, f) for f, _ in i._fields_}, nshards=n.value)

    def shard_info(self, k: int):
        h, i = C.c_void_p(), _lib.DbInfo()
        _check(_lib.load().swa_group_shard(self._h, k, C.byref(h)))
        _check(_lib.load().swa_db_info(h, C.byref(i)))
        return {f: getattr(i, f) for f, _ in i._fields_}

    def set_inclusion(self, include=None):
        if include is None:
            _check(_lib.load().swa_group_set_inclusion(self._h, None, 0))
            return
        inc = np.ascontiguousarray(include, dtype=np.uint8)
        _check(_lib.load().swa_group_set_inclusion(self._h, inc.ctypes.data, len(inc)))

    def set_scoring(self, matrix: np.ndarray, gapopen: int, gapextend: int):
        M = _i64(matrix)
        _check(_lib.load().swa_group_set_scoring(self._h, M.ctypes.data, gapopen + gapextend, gapextend))

    def set_option(self, key: str, value=None):
        _check(_lib.load().swa_group_set_option(self._h, key.encode(), None if value is None else str(value).encode()))

    def search(self, query: np.ndarray, *, want_scores: bool = True):
        q = np.ascontiguousarray(query, dtype=np.uint8)
        c = _lib.Counters()
        i = self.info()
        scores = np.empty(i["seqcount"] * i["frames"], dtype=np.int64) if want_scores else None
        _check(_lib.load().swa_group_search(self._h, q.ctypes.data, len(q), scores.ctypes.data if want_scores else None, C.byref(c)))
        return scores, {f: getattr(c, f) for f, _ in c._fields_}

    def search_topk(self, query: np.ndarray, keep: int = 250, minscore: int = 1, maxscore: int = (1 << 62)):
        q = np.ascontiguousarray(query, dtype=np.uint8)
        c = _lib.Counters()
        hits = (_lib.Hit * max(1, keep))()
        n, tot, obv = C.c_int64(), C.c_int64(), C.c_int64()
        _check(_lib.load().swa_group_search_topk(self._h, q.ctypes.data, len(q), keep, minscore, maxscore, hits, C.byref(n),
                                                 C.byref(tot), C.byref(obv), C.byref(c)))
        return ([(hits[i].seqno, hits[i].score) for i in range(n.value)], tot.value, obv.value,
                {f: getattr(c, f) for f, _ in c._fields_})

    def search_pair_topk(self, query1, query2, keep=250, minscore=(1, 1), maxscore=((1 << 62), (1 << 62))):
        q1 = np.ascontiguousarray(query1, dtype=np.uint8)
        q2 = np.ascontiguousarray(query2, dtype=np.uint8)
        two = lambda v: tuple(v) if isinstance(v, (tuple, list)) else (v, v)
        k, lo, hi = two(keep), two(minscore), two(maxscore)
        c = _lib.Counters()
        h1, h2 = (_lib.Hit * max(1, k[0]))(), (_lib.Hit * max(1, k[1]))()
        n1, t1, o1, n2, t2, o2 = (C.c_int64() for _ in range(6))
        _check(_lib.load().swa_group_search_pair_topk(self._h, q1.ctypes.data, len(q1), q2.ctypes.data, len(q2), k[0], lo[0], hi[0],
                                                      k[1], lo[1], hi[1], h1, C.byref(n1), C.byref(t1), C.byref(o1), h2,
                                                      C.byref(n2), C.byref(t2), C.byref(o2), C.byref(c)))
        return (([(h1[i].seqno, h1[i].score) for i in range(n1.value)], t1.value, o1.value),
                ([(h2[i].seqno, h2[i].score) for i in range(n2.value)], t2.value, o2.value),
                {f: getattr(c, f) for f, _ in c._fields_})

    def search_frames_topk(self, queries, qtags=None, *, keep: int = 250, minscore: int = 1, maxscore: int = (1 << 62)):
        L = _lib.load()
        qs = [np.ascontiguousarray(q, dtype=np.uint8) for q in queries]
        ptr = (C.c_void_p * len(qs))(*[q.ctypes.data for q in qs])
        lens = (C.c_int64 * len(qs))(*[len(q) for q in qs])
        tags = (C.c_int32 * len(qs))(*(qtags if qtags is not None else range(len(qs))))
        hits = (_lib.FrameHit * max(keep, 1))()
        n, tot, obv = C.c_int64(), C.c_int64(), C.c_int64()
        c = _lib.Counters()
        _check(L.swa_group_search_frames_topk(self._h, len(qs), ptr, lens, tags, keep, minscore, maxscore, hits, C.byref(n),
                                              C.byref(tot), C.byref(obv), C.byref(c)))
        return ([(h.seqno, h.score, h.qstrand, h.qframe, h.dstrand, h.dframe) for h in hits[: n.value]], tot.value,
                obv.value, {f: getattr(c, f) for f, _ in c._fields_})

    def sequence(self, seqno: int, dstrand: int = 0, dframe: int = 0) -> np.ndarray:
        L = _lib.load()
        n = C.c_int64()
        rc = L.swa_group_db_sequence(self._h, seqno, dstrand, dframe, None, 0, C.byref(n), None)
        if rc not in (0, _lib.SWA_ERANGE):
            _check(rc)
        buf = np.empty(max(n.value, 1), dtype=np.uint8)
        _check(L.swa_group_db_sequence(self._h, seqno, dstrand, dframe, buf.ctypes.data, n.value, C.byref(n), None))
        return buf[: n.value]

    def align(self, query: np.ndarray, seqnos, dstrands=None, dframes=None):
        L = _lib.load()
        q = np.ascontiguousarray(query, dtype=np.uint8)
        ids = _i64(seqnos)
        ds = None if dstrands is None else np.ascontiguousarray(dstrands, dtype=np.int32)
        df = None if dframes is None else np.ascontiguousarray(dframes, dtype=np.int32)
        out = (_lib.Alignment * max(len(ids), 1))()
        cap = 1 << 16
        while True:
            text = C.create_string_buffer(cap)
            used = C.c_int64()
            rc = L.swa_group_align_hits(self._h, q.ctypes.data, len(q), ids.ctypes.data, None if ds is None else ds.ctypes.data,
                                        None if df is None else df.ctypes.data, len(ids), out, text, cap, C.byref(used))
            if rc == _lib.SWA_ERANGE:
                cap = used.value
                continue
            _check(rc)
            break
        return [_alignment_dict(out[i], text.raw) for i in range(len(ids))]

    def _unsupported(self, *a, **k):
        raise SwaError("not a group operation: use a Database shard")

    search_topk_array = search2 = search2_topk = search_endpoints = _unsupported

    def close(self):
        if self._h:
            _lib.load().swa_group_close(self._h)
            self._h = None


def shard_bounds(offsets, nshards: int) -> np.ndarray:
    """swa_shard_bounds: int64 [nshards + 1] cuts of residue-balanced contiguous shards (the C++ twin of parallel.shard_bounds)"""
    off = _i64(offsets)
    cuts = np.zeros(nshards + 1, dtype=np.int64)
    _check(_lib.load().swa_shard_bounds(off.ctypes.data, len(off) - 1, nshards, cuts.ctypes.data))
    return cuts


def blastdb_shard_bounds(basename: str, nshards: int, *, symtype: int = 1) -> np.ndarray:
    cuts = np.zeros(nshards + 1, dtype=np.int64)
    _check(_lib.load().swa_blastdb_shard_bounds(os.fsencode(basename), symtype, nshards, cuts.ctypes.data))
    return cuts


def _alignment_dict(a, text: bytes) -> dict:
    d = {f: getattr(a, f) for f, _ in a._fields_ if not f.startswith("cigar_") and f != "reserved"}
    d["cigar"] = text[a.cigar_offset: a.cigar_offset + a.cigar_len].decode()
    return d


def traceback(query, dseq, matrix, gapopen: int, gapextend: int, hint=None) -> dict:
    """Host part of the alignment phase for one sequence held by the caller (swa_traceback);
    hint = (score, q_end, d_end) from search_endpoints or None for a forward sweep."""
    L = _lib.load()
    q = np.ascontiguousarray(query, dtype=np.uint8)
    d = np.ascontiguousarray(dseq, dtype=np.uint8)
    M = np.ascontiguousarray(matrix, dtype=np.int64)
    hs, hq, hd = hint if hint else (0, 0, 0)
    a = _lib.Alignment()
    cap = 16 * (len(q) + len(d)) + 64
    text = C.create_string_buffer(cap)
    used = C.c_int64()
    _check(L.swa_traceback(q.ctypes.data, len(q), d.ctypes.data, len(d), M.ctypes.data, gapopen, gapextend, hs, hq, hd,
                           C.byref(a), text, cap, C.byref(used)))
    return _alignment_dict(a, text.raw)


class Headers:
    """Definition lines, OID mask and taxid filter of a BLAST v4 database (host only)."""
    SHOW_GIS, SHOW_TAXID = 1, 2

    def __init__(self, basename: str, *, symtype: int = 1, taxidfile: Optional[str] = None):
        self._h = C.c_void_p()
        _check(_lib.load().swa_headers_open(os.fsencode(basename), symtype, os.fsencode(taxidfile) if taxidfile else None,
                                            C.byref(self._h)))

    def info(self) -> dict:
        v = [C.c_int64() for _ in range(5)]
        title = C.create_string_buffer(4096)
        _check(_lib.load().swa_headers_info(self._h, *[C.byref(x) for x in v], title, 4096))
        keys = ("seqcount", "symcount", "masked_seqcount", "masked_symcount", "longest")
        return dict(zip(keys, (x.value for x in v)), title=title.value.decode())

    def get(self, seqno: int, flags: int = 0):
        """the definition lines of `seqno` that pass the membership / taxid filters"""
        need = C.c_int64()
        buf = C.create_string_buffer(4096)
        rc = _lib.load().swa_headers_get(self._h, seqno, flags, buf, 4096, C.byref(need))
        if rc == _lib.SWA_ERANGE:
            buf = C.create_string_buffer(need.value)
            rc = _lib.load().swa_headers_get(self._h, seqno, flags, buf, need.value, C.byref(need))
        _check(rc)
        text = buf.value.decode()
        return text.split("\n") if text else []

    def inclusion(self, first_seqno: int, n: int) -> np.ndarray:
        out = np.zeros(max(n, 1), dtype=np.uint8)
        _check(_lib.load().swa_headers_inclusion(self._h, first_seqno, n, out.ctypes.data))
        return out[:n]

    def close(self):
        if self._h:
            _lib.load().swa_headers_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def translate_table(gencode: int) -> np.ndarray:
    """table[4096] of NCBIstdaa codes indexed by three IUPAC nibbles (translate_createtable)"""
    t = np.zeros(4096, dtype=np.uint8)
    _check(_lib.load().swa_translate_table(gencode, t.ctypes.data))
    return t


def translate(dna, strand: int, frame: int, table: np.ndarray) -> np.ndarray:
    """translate(): frame 0..2 of strand 0/1 of a nucleotide sequence (nibble codes) as NCBIstdaa codes"""
    d = np.ascontiguousarray(dna, dtype=np.uint8)
    out = np.zeros(max(len(d) // 3, 1), dtype=np.uint8)
    n = C.c_int64()
    _check(_lib.load().swa_translate(d.ctypes.data, len(d), strand, frame, table.ctypes.data, out.ctypes.data, C.byref(n)))
    return out[: n.value].copy()


def read_blastdb(basename: str, *, symtype: int = 1, first_seqno: int = 0, last_seqno: int = -1):
    """(residues, offsets, info) of a BLAST v4 database through the C++ loader (host only)."""
    L = _lib.load()
    r, o = C.c_void_p(), C.c_void_p()
    n, ts, ty, lg = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
    _check(L.swa_blastdb_read(os.fsencode(basename), symtype, first_seqno, last_seqno, C.byref(r), C.byref(o),
                              C.byref(n), C.byref(ts), C.byref(ty), C.byref(lg)))
    try:
        off = np.ctypeslib.as_array(C.cast(o, C.POINTER(C.c_int64)), shape=(n.value + 1,)).copy()
        res = np.ctypeslib.as_array(C.cast(r, C.POINTER(C.c_uint8)), shape=(max(int(off[-1]), 1),))[: int(off[-1])].copy()
    finally:
        L.swa_free(r)
        L.swa_free(o)
    return res, off, {"total_seqcount": ts.value, "total_symcount": ty.value, "longest": lg.value}


def write_blastdb(basename: str, residues: np.ndarray, offsets: np.ndarray, *, symtype: int = 1, first_id: int = 0,
                  title: str = "swipe_amd synthetic") -> None:
    """One BLAST v4 volume from (residues, offsets) through the streaming C++ writer (swa_blastdb_write)."""
    res = np.ascontiguousarray(residues, dtype=np.uint8)
    off = _i64(offsets)
    _check(_lib.load().swa_blastdb_write(os.fsencode(basename), symtype, res.ctypes.data, off.ctypes.data, len(off) - 1,
                                         first_id, title.encode()))


def synth_offsets(seed: int, nseq: int, *, first: int = 0, query: Optional[np.ndarray] = None, threads: int = 0) -> np.ndarray:
    """int64 [nseq + 1] prefix sums of the lengths of synthetic sequences [first, first + nseq) - the lengths only,
    so that every rank can place its shard of ONE database without generating the others' residues."""
    from . import synth
    L = _lib.load()
    ltab = np.ascontiguousarray(synth.length_table(), dtype=np.int32)
    q = np.ascontiguousarray(query, dtype=np.uint8) if query is not None else np.zeros(0, np.uint8)
    off = np.zeros(nseq + 1, dtype=np.int64)
    L.swa_synth_offsets(seed, first, nseq, ltab.ctypes.data, q.ctypes.data if len(q) else None, len(q), off.ctypes.data,
                        threads or os.cpu_count() or 1)
    return off


def synth_db(seed: int, nseq: int, *, first: int = 0, query: Optional[np.ndarray] = None, protein: bool = True,
             threads: int = 0):
    """(residues uint8, offsets int64) of synthetic sequences [first, first+nseq) - C++ generator."""
    from . import synth
    L = _lib.load()
    threads = threads or os.cpu_count() or 1
    ltab = np.ascontiguousarray(synth.length_table(), dtype=np.int32)
    rtab = np.ascontiguousarray(synth.residue_table_protein() if protein else synth.residue_table_nucleotide())
    q = np.ascontiguousarray(query, dtype=np.uint8) if query is not None else np.zeros(0, np.uint8)
    qp = q.ctypes.data if len(q) else None
    off = np.zeros(nseq + 1, dtype=np.int64)
    total = L.swa_synth_offsets(seed, first, nseq, ltab.ctypes.data, qp, len(q), off.ctypes.data, threads)
    res = np.empty(total, dtype=np.uint8)
    L.swa_synth_fill(seed, first, nseq, ltab.ctypes.data, rtab.ctypes.data, qp, len(q), off.ctypes.data,
                     res.ctypes.data, threads)
    return res, off
