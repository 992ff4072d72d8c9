"""NCBI BLAST database *version 4* volumes: writer and reader (pure Python/numpy).

This is the on-disk format the reference reads with ``db_open`` / ``db_getsequence``
(reference database.cc:515-608 for the ``.pin``/``.nin`` index, database.cc:1237-1401
for sequence extraction, database.cc:406-489 for the ``.pal``/``.nal`` alias).  It is
used here to (a) produce fixture / benchmark databases that both the compiled
reference (``oracle/_ref/swipe``) and this repo's C++ loader open, and (b) as the
Python mirror of the C++ loader for tests.

Layout facts (all integers big-endian unless noted):

``.pin``/``.nin``: u32 version=4 | u32 type (1 protein, 0 nucleotide) | u32 n + title |
u32 n + date | zero pad to 4-byte alignment | u32 nseq | **little-endian** u64 total
residues (reference database.cc:595) | u32 longest | u32 hdr_off[nseq+1] |
u32 seq_off[nseq+1] | (nucleotide only) u32 amb_off[nseq+1].

``.psq``: 0x00, then each sequence as NCBIstdaa codes followed by 0x00; entry ``s``
spans ``[seq_off[s], seq_off[s+1])`` *including* its terminator.

``.nsq``: 4 bases per byte, A,C,G,T = 0..3, most significant pair first; the final
byte of every sequence carries ``len % 4`` in its low two bits; optional ambiguity
table ``[amb_off[s], seq_off[s+1])``: u32 count, then u32 entries
``code<<28 | (run-1)<<24 | offset``.

``.phr``/``.nhr``: ASN.1 BER Blast-def-line-set, indefinite lengths, one local id
and a title per sequence (the smallest form reference asnparse.cc:753 accepts).
"""
from __future__ import annotations

import os
import struct
from dataclasses import dataclass
from typing import Iterable, List, Optional, Sequence

import numpy as np

# NCBIstdaa alphabet, code = index (reference query.cc:178 sym_ncbi_aa)
NCBISTDAA = "-ABCDEFGHIKLMNPQRSTVWXYZU*OJ"
# NCBI4na / nt16 alphabet, code = bitmask A=1 C=2 G=4 T=8 (reference query.cc:176)
NCBI4NA = "-ACMGRSVTWYHKDBN"

AA_ENCODE = np.full(256, 255, dtype=np.uint8)
for _i, _c in enumerate(NCBISTDAA):
    AA_ENCODE[ord(_c)] = _i
    AA_ENCODE[ord(_c.lower())] = _i
NT16_ENCODE = np.full(256, 255, dtype=np.uint8)
for _i, _c in enumerate(NCBI4NA):
    if _c != "-":
        NT16_ENCODE[ord(_c)] = _i
        NT16_ENCODE[ord(_c.lower())] = _i
NT16_ENCODE[ord("U")] = 8
NT16_ENCODE[ord("u")] = 8
# complement of a 4-bit base mask (reference query.cc:112 ntcompl)
NT16_COMPLEMENT = np.array([0, 8, 4, 12, 2, 10, 6, 14, 1, 9, 5, 13, 3, 11, 7, 15], dtype=np.uint8)


def encode_protein(text: str | bytes) -> np.ndarray:
    """ASCII letters -> NCBIstdaa codes (unknown letters are dropped, as query.cc:310-330 does)."""
    raw = np.frombuffer(text.encode() if isinstance(text, str) else bytes(text), dtype=np.uint8)
    codes = AA_ENCODE[raw]
    return codes[codes != 255].copy()


def encode_nucleotide(text: str | bytes) -> np.ndarray:
    """ASCII IUPAC letters -> 4-bit base masks (A=1, C=2, G=4, T=8, N=15...)."""
    raw = np.frombuffer(text.encode() if isinstance(text, str) else bytes(text), dtype=np.uint8)
    codes = NT16_ENCODE[raw]
    return codes[codes != 255].copy()


def revcomp_nt16(codes: np.ndarray) -> np.ndarray:
    return NT16_COMPLEMENT[np.asarray(codes, dtype=np.uint8)[::-1]].copy()


def _ber_visible_string(s: bytes) -> bytes:
    n = len(s)
    if n < 128:
        return b"\x1a" + bytes([n]) + s
    nb = (n.bit_length() + 7) // 8
    return b"\x1a" + bytes([0x80 | nb]) + n.to_bytes(nb, "big") + s


def ber_defline(local_id: str, title: str) -> bytes:
    """Smallest Blast-def-line-set the reference's parser renders as ``lcl|id title``."""
    eoc = b"\x00\x00"
    out = b"\x30\x80" + b"\x30\x80"
    out += b"\xa0\x80" + _ber_visible_string(title.encode()) + eoc          # title
    out += b"\xa1\x80" + b"\x30\x80" + b"\xa0\x80" + b"\xa1\x80"            # seqid { local { str
    out += _ber_visible_string(local_id.encode()) + eoc + eoc + eoc + eoc
    out += eoc + eoc
    return out


_EOC = b"\x00\x00"
_TEXTSEQ_TAGS = {"gb": 0xA4, "emb": 0xA5, "pir": 0xA6, "sp": 0xA7, "ref": 0xA9, "dbj": 0xAC, "prf": 0xAD,
                 "tpg": 0xAF, "tpe": 0xB0, "tpd": 0xB1, "gpp": 0xB2, "nat": 0xB3}


def _ber_int(v: int) -> bytes:
    n = max(1, (int(v).bit_length() + 7) // 8)
    return b"\x02" + bytes([n]) + int(v).to_bytes(n, "big")


def _ctx(tag: int, body: bytes) -> bytes:
    """context-tagged constructed element, indefinite length (the only form the reference's parser reads)"""
    return bytes([tag, 0x80]) + body + _EOC


def _seq(body: bytes) -> bytes:
    return b"\x30\x80" + body + _EOC


def ber_seq_id(sid: tuple) -> bytes:
    """One Seq-id in the binary ASN.1 of BLAST v4 headers (NCBI seqloc.asn; reference asnparse.cc:640-751).
    Forms: ("lcl", str|int)  ("bbs"|"bbm", int)  ("gim", int)  ("gi", int)  ("gnl", db, str|int)
    ("pdb", mol, chain_code)  ("pat", country, number, seqno, granted)  and the Textseq-ids
    (db, accession, name[, version[, release]]) for gb emb pir sp ref dbj prf tpg tpe tpd gpp nat."""
    kind = sid[0]
    vs = lambda x: _ber_visible_string(str(x).encode())
    obj = lambda x: _ctx(0xA0, _ber_int(x)) if isinstance(x, int) else _ctx(0xA1, vs(x))
    if kind == "lcl":
        return _ctx(0xA0, obj(sid[1]))
    if kind in ("bbs", "bbm"):
        return _ctx(0xA1 if kind == "bbs" else 0xA2, _ber_int(sid[1]))
    if kind == "gim":
        return _ctx(0xA3, _seq(_ctx(0xA0, _ber_int(sid[1]))))
    if kind == "gi":
        return _ctx(0xAB, _ber_int(sid[1]))
    if kind == "gnl":
        return _ctx(0xAA, _seq(_ctx(0xA0, vs(sid[1])) + _ctx(0xA1, obj(sid[2]))))
    if kind == "pdb":
        body = _ctx(0xA0, vs(sid[1]))
        if len(sid) > 2 and sid[2] is not None:
            body += _ctx(0xA1, _ber_int(sid[2]))
        return _ctx(0xAE, _seq(body))
    if kind == "pat":
        _, country, number, seqno, granted = sid
        idpat = _seq(_ctx(0xA0, vs(country)) + _ctx(0xA1, _ctx(0xA0 if granted else 0xA1, vs(number))))
        return _ctx(0xA8, _seq(_ctx(0xA0, _ber_int(seqno)) + _ctx(0xA1, idpat)))
    if kind in _TEXTSEQ_TAGS:
        acc, name = sid[1], sid[2]
        version = sid[3] if len(sid) > 3 else 0
        release = sid[4] if len(sid) > 4 else None
        body = b""
        if name:
            body += _ctx(0xA0, vs(name))
        if acc:
            body += _ctx(0xA1, vs(acc))
        if release:
            body += _ctx(0xA2, vs(release))
        if version:
            body += _ctx(0xA3, _ber_int(version))
        return _ctx(_TEXTSEQ_TAGS[kind], _seq(body))
    raise ValueError(f"unknown Seq-id kind {kind!r}")


def ber_defline_set(deflines: Sequence[dict]) -> bytes:
    """A Blast-def-line-set: one dict per definition line with optional keys title, ids (tuples for
    ber_seq_id), taxid, memb, links (reference asnparse.cc:753-856)."""
    out = b""
    for d in deflines:
        body = b""
        if d.get("title") is not None:
            body += _ctx(0xA0, _ber_visible_string(d["title"].encode()))
        if d.get("ids"):
            body += _ctx(0xA1, _seq(b"".join(ber_seq_id(i) for i in d["ids"])))
        if d.get("taxid") is not None:
            body += _ctx(0xA2, _ber_int(d["taxid"]))
        if d.get("memb") is not None:
            body += _ctx(0xA3, _seq(_ber_int(d["memb"])))
        if d.get("links") is not None:
            body += _ctx(0xA4, _seq(_ber_int(d["links"])))
        out += _seq(body)
    return _seq(out)


def write_mask_alias(alias_basename: str, volume_basename: str, include: Sequence[bool], *, memb_bit: int = 1,
                     length: int = 0, title: str = "masked subset", protein: bool = True) -> None:
    """A masked database as NCBI ships swissprot/pdbaa: an alias naming ONE volume plus an OID mask file
    (bit 7-(s&7) of byte 4+(s>>3) set = sequence s is a member; reference database.cc:687-706, 775-870)."""
    inc = np.asarray(include, dtype=bool)
    maxoid = int(np.nonzero(inc)[0].max()) if inc.any() else 0
    bits = np.packbits(inc.astype(np.uint8))            # MSB first, as db_check_msk reads it
    msk = os.path.basename(alias_basename) + ".msk"
    with open(os.path.join(os.path.dirname(alias_basename), msk), "wb") as f:
        f.write(struct.pack(">I", maxoid) + bits.tobytes())
    with open(f"{alias_basename}.{'pal' if protein else 'nal'}", "w") as f:
        f.write(f"#\n# Alias file created by swipe_amd\n#\nTITLE {title}\nDBLIST {os.path.basename(volume_basename)}\n"
                f"OIDLIST {msk}\nLENGTH {length}\nNSEQ {int(inc.sum())}\nMAXOID {maxoid}\nMEMB_BIT {memb_bit}\n")


def pack_nucleotide(codes: np.ndarray):
    """4-bit masks -> (2-bit packed bytes incl. remainder byte, ambiguity table bytes)."""
    codes = np.asarray(codes, dtype=np.uint8)
    n = len(codes)
    two = np.zeros(n, dtype=np.uint8)
    onehot = {1: 0, 2: 1, 4: 2, 8: 3}
    amb_mask = np.ones(n, dtype=bool)
    for m, v in onehot.items():
        sel = codes == m
        two[sel] = v
        amb_mask[sel] = False
    full = n // 4
    body = np.zeros(full + 1, dtype=np.uint8)
    if full:
        q = two[: full * 4].reshape(full, 4)
        body[:full] = (q[:, 0] << 6) | (q[:, 1] << 4) | (q[:, 2] << 2) | q[:, 3]
    rem = n - full * 4
    last = 0
    for k in range(rem):
        last |= int(two[full * 4 + k]) << (6 - 2 * k)
    body[full] = last | rem
    amb = b""
    if amb_mask.any():
        entries = []
        i = 0
        while i < n:
            if amb_mask[i]:
                j = i
                while j + 1 < n and amb_mask[j + 1] and codes[j + 1] == codes[i] and j + 1 - i < 16:
                    j += 1
                if i >= (1 << 24):
                    raise ValueError("ambiguity offset needs the 64-bit table form; not written here")
                entries.append((int(codes[i]) << 28) | ((j - i) << 24) | i)
                i = j + 1
            else:
                i += 1
        amb = struct.pack(">I", len(entries)) + b"".join(struct.pack(">I", e) for e in entries)
    return body.tobytes(), amb


def write_volume(basename: str, seqs: Sequence[np.ndarray], *, protein: bool = True,
                 ids: Optional[Sequence[str]] = None, titles: Optional[Sequence[str]] = None,
                 title: str = "swipe_amd synthetic", date: str = "Jan 1, 2026  0:00 AM",
                 headers: Optional[Sequence[Sequence[dict]]] = None) -> None:
    """Write one v4 volume (``.pin/.psq/.phr`` or ``.nin/.nsq/.nhr``).  headers[i] (a list of defline dicts
    for ber_defline_set) overrides the plain ``lcl|id title`` header of sequence i."""
    ext = ("pin", "psq", "phr") if protein else ("nin", "nsq", "nhr")
    n = len(seqs)
    hdr_off = [0]
    hdr = bytearray()
    for i in range(n):
        if headers is not None and headers[i] is not None:
            hdr += ber_defline_set(headers[i])
        else:
            hdr += ber_defline(ids[i] if ids else f"s{i}", titles[i] if titles else f"seq{i}")
        hdr_off.append(len(hdr))
    seq_off = []
    amb_off = []
    total = 0
    longest = 0
    if protein:
        lens = np.fromiter((len(s) for s in seqs), dtype=np.int64, count=n)
        offs = np.empty(n + 1, dtype=np.int64)
        offs[0] = 1
        np.cumsum(lens + 1, out=offs[1:])
        offs[1:] += 1
        buf = np.zeros(int(offs[-1]), dtype=np.uint8)
        for i, s in enumerate(seqs):
            buf[offs[i]: offs[i] + lens[i]] = s
        seq_off = offs.tolist()
        total = int(lens.sum())
        longest = int(lens.max()) if n else 0
        sq = buf.tobytes()
    else:
        sqb = bytearray(b"\x00")
        for s in seqs:
            body, amb = pack_nucleotide(s)
            seq_off.append(len(sqb))
            sqb += body
            amb_off.append(len(sqb))
            sqb += amb
            total += len(s)
            longest = max(longest, len(s))
        seq_off.append(len(sqb))
        amb_off.append(len(sqb))
        sq = bytes(sqb)
    if len(sq) >= (1 << 32):
        raise ValueError("volume exceeds the 4 GiB u32 offset limit of the v4 format")
    t = title.encode()
    d = date.encode()
    pin = struct.pack(">II", 4, 1 if protein else 0)
    pin += struct.pack(">I", len(t)) + t + struct.pack(">I", len(d)) + d
    pin += b"\x00" * ((-len(pin)) % 4)
    pin += struct.pack(">I", n) + struct.pack("<Q", total) + struct.pack(">I", longest)
    pin += np.asarray(hdr_off, dtype=">u4").tobytes()
    pin += np.asarray(seq_off, dtype=">u4").tobytes()
    if not protein:
        pin += np.asarray(amb_off, dtype=">u4").tobytes()
    with open(f"{basename}.{ext[0]}", "wb") as f:
        f.write(pin)
    with open(f"{basename}.{ext[1]}", "wb") as f:
        f.write(sq)
    with open(f"{basename}.{ext[2]}", "wb") as f:
        f.write(bytes(hdr))


def write_protein_volume_arrays(basename: str, residues: np.ndarray, offsets: np.ndarray, *,
                                title: str = "swipe_amd synthetic", first_id: int = 0) -> None:
    """Vectorised protein volume writer from (residues, offsets) arrays (large sample databases)."""
    residues = np.asarray(residues, dtype=np.uint8)
    off = np.asarray(offsets, dtype=np.int64) - int(offsets[0])
    n = len(off) - 1
    lens = np.diff(off)
    total = int(off[-1])
    if total + n + 1 >= (1 << 32):
        raise ValueError("volume exceeds the 4 GiB u32 offset limit of the v4 format")
    sq = np.zeros(total + n + 1, dtype=np.uint8)
    pos = np.arange(total, dtype=np.int64) + np.repeat(np.arange(n, dtype=np.int64), lens) + 1
    sq[pos] = residues[int(offsets[0]): int(offsets[0]) + total]
    seq_off = off + np.arange(n + 1, dtype=np.int64) + 1
    hdrs = [ber_defline(f"s{first_id + i}", f"seq{first_id + i}") for i in range(n)]
    hdr_off = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(np.fromiter((len(h) for h in hdrs), dtype=np.int64, count=n), out=hdr_off[1:])
    t, d = title.encode(), b"Jan 1, 2026  0:00 AM"
    pin = struct.pack(">II", 4, 1) + struct.pack(">I", len(t)) + t + struct.pack(">I", len(d)) + d
    pin += b"\x00" * ((-len(pin)) % 4)
    pin += struct.pack(">I", n) + struct.pack("<Q", total) + struct.pack(">I", int(lens.max()) if n else 0)
    pin += hdr_off.astype(">u4").tobytes() + seq_off.astype(">u4").tobytes()
    with open(basename + ".pin", "wb") as f:
        f.write(pin)
    sq.tofile(basename + ".psq")
    with open(basename + ".phr", "wb") as f:
        f.write(b"".join(hdrs))


def write_alias(basename: str, volume_basenames: Iterable[str], *, protein: bool = True,
                title: str = "swipe_amd synthetic") -> None:
    """``.pal``/``.nal`` alias listing volumes by file name relative to the alias' directory."""
    names = " ".join(os.path.basename(v) for v in volume_basenames)
    with open(f"{basename}.{'pal' if protein else 'nal'}", "w") as f:
        f.write(f"#\n# Alias file created by swipe_amd\n#\nTITLE {title}\nDBLIST {names}\n")


def write_db(basename: str, seqs: Sequence[np.ndarray], *, protein: bool = True,
             volumes: int = 1, **kw) -> None:
    """Write a database as one volume, or ``volumes`` volumes behind an alias."""
    if volumes <= 1:
        write_volume(basename, seqs, protein=protein, **kw)
        return
    n = len(seqs)
    bounds = [n * v // volumes for v in range(volumes + 1)]
    names = []
    ids = kw.pop("ids", None)
    titles = kw.pop("titles", None)
    for v in range(volumes):
        name = f"{basename}.{v:02d}"
        lo, hi = bounds[v], bounds[v + 1]
        write_volume(name, seqs[lo:hi], protein=protein,
                     ids=(ids[lo:hi] if ids else [f"s{i}" for i in range(lo, hi)]),
                     titles=(titles[lo:hi] if titles else [f"seq{i}" for i in range(lo, hi)]), **kw)
        names.append(name)
    write_alias(basename, names, protein=protein, title=kw.get("title", "swipe_amd synthetic"))


@dataclass
class Volume:
    protein: bool
    title: str
    nseq: int
    total: int
    longest: int
    hdr_off: np.ndarray
    seq_off: np.ndarray
    amb_off: Optional[np.ndarray]
    sq: np.ndarray  # the raw .psq/.nsq bytes

    def sequence(self, s: int) -> np.ndarray:
        """Residue codes of entry ``s`` (protein: NCBIstdaa; nucleotide: 4-bit masks)."""
        o1, o2 = int(self.seq_off[s]), int(self.seq_off[s + 1])
        if self.protein:
            return self.sq[o1:o2 - 1]
        o3 = int(self.amb_off[s])
        body = self.sq[o1:o3]
        ntlen = 4 * (len(body) - 1) + int(body[-1] & 3)
        shifts = np.array([6, 4, 2, 0], dtype=np.uint8)
        two = ((body[:, None] >> shifts[None, :]) & 3).reshape(-1)[:ntlen]
        out = (1 << two).astype(np.uint8)
        if o2 > o3:
            amb = self.sq[o3:o2].tobytes()
            (cnt,) = struct.unpack(">I", amb[:4])
            if cnt >> 31:
                for k in range((len(amb) - 4) // 8):
                    (e,) = struct.unpack(">Q", amb[4 + 8 * k: 12 + 8 * k])
                    code, run, off = e >> 60, ((e >> 48) & 0xFFF) + 1, e & 0xFFFFFFFFFFF
                    out[off:off + run] = code
            else:
                for k in range((len(amb) - 4) // 4):
                    (e,) = struct.unpack(">I", amb[4 + 4 * k: 8 + 4 * k])
                    code, run, off = e >> 28, ((e >> 24) & 0xF) + 1, e & 0xFFFFFF
                    out[off:off + run] = code
        return out


def read_volume(basename: str, protein: bool = True) -> Volume:
    ext = ("pin", "psq") if protein else ("nin", "nsq")
    with open(f"{basename}.{ext[0]}", "rb") as f:
        pin = f.read()
    version, typ, tl = struct.unpack(">III", pin[:12])
    if version != 4:
        raise ValueError("Illegal database version (must be 4).")
    p = 12
    title = pin[p:p + tl].decode()
    p += tl
    (dl,) = struct.unpack(">I", pin[p:p + 4])
    p += 4 + dl
    p += (-p) % 4
    (nseq,) = struct.unpack(">I", pin[p:p + 4])
    (total,) = struct.unpack("<Q", pin[p + 4:p + 12])
    (longest,) = struct.unpack(">I", pin[p + 12:p + 16])
    p += 16
    tab = np.frombuffer(pin, dtype=">u4", offset=p)
    hdr_off = tab[: nseq + 1].astype(np.int64)
    seq_off = tab[nseq + 1: 2 * nseq + 2].astype(np.int64)
    amb_off = None if protein else tab[2 * nseq + 2: 3 * nseq + 3].astype(np.int64)
    sq = np.fromfile(f"{basename}.{ext[1]}", dtype=np.uint8)
    return Volume(bool(typ), title, nseq, total, longest, hdr_off, seq_off, amb_off, sq)


def read_db(basename: str, protein: bool = True) -> List[Volume]:
    """Open a database: alias (one level, ``DBLIST``) or single volume."""
    alias = f"{basename}.{'pal' if protein else 'nal'}"
    if os.path.exists(alias):
        names: List[str] = []
        with open(alias) as f:
            for line in f:
                if line.startswith("DBLIST"):
                    names += line.split()[1:]
        d = os.path.dirname(basename)
        return [read_volume(os.path.join(d, nm), protein) for nm in names]
    return [read_volume(basename, protein)]
