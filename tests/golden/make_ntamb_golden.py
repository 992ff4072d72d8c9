#!/usr/bin/env python3
"""tests/golden/ntamb.json: the reference CLI on a nucleotide database whose .nsq entries carry ambiguity tables in BOTH forms the
format has - 32-bit entries (runs of at most 16 bases) in the first volume, 64-bit entries (header bit 31, runs of up to 4 096;
database.cc:1284-1323) in the second - with the query planted on both strands across the ambiguous runs, ambiguity codes at the
first and last base, one sequence of N only.  build(dir) writes the database and the query (seeded, so the test rebuilds the same
bytes); main() runs oracle/_ref/swipe on it.  Build container only (needs oracle/_ref/swipe).

tests/golden/ntamb_overlap.json (round 6): the same database with its ambiguity tables DISORDERED - every
table written back to front, and inside the planted query three runs that overlap each other (N x 12 at b, R x 6 at b + 5,
Y x 3 at b - 2, in that order).  The reference applies the entries in file order, the last writer wins (database.cc:1296-1321);
no formatter writes such tables, the format allows them."""
import json, os, subprocess, sys, tempfile
import struct
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import numpy as np
from swipe_amd import blastdb

REF = os.path.join(ROOT, "oracle", "_ref", "swipe")
ARGS = ["-p", "0", "-r", "1", "-q", "-3", "-G", "5", "-E", "2", "-e", "10", "-b", "12", "-v", "40"]


def pack_new_format(codes, _old=blastdb.pack_nucleotide):
    """blastdb.pack_nucleotide with the ambiguity table in the 64-bit form: code:4 | run-1:12 | pad:4 | position:44"""
    body, _ = _old(codes)
    codes = np.asarray(codes, dtype=np.uint8)
    amb = ~np.isin(codes, (1, 2, 4, 8))
    entries, i, n = [], 0, len(codes)
    while i < n:
        if amb[i]:
            j = i
            while j + 1 < n and amb[j + 1] and codes[j + 1] == codes[i] and j + 1 - i < 4096:
                j += 1
            entries.append((int(codes[i]) << 60) | ((j - i) << 48) | i)
            i = j + 1
        else:
            i += 1
    table = b"" if not entries else struct.pack(">I", 0x80000000 | (2 * len(entries))) + b"".join(struct.pack(">Q", e) for e in entries)
    return body, table


def disordered(pack):
    """pack_nucleotide / pack_new_format with the entries of the table reversed, then - where the sequence has an ambiguous run of
    at least 12 bases that starts at b >= 2 - three more entries that overlap that run and each other"""
    def wrapped(codes):
        body, table = pack(codes)
        if not table:
            return body, table
        big = (struct.unpack(">I", table[:4])[0] >> 31) != 0
        es = 8 if big else 4
        ents = [table[4 + i:4 + i + es] for i in range(0, len(table) - 4, es)]
        codes = np.asarray(codes, dtype=np.uint8)
        amb = ~np.isin(codes, (1, 2, 4, 8))
        extra = []
        run = 0
        for i in range(len(codes) + 1):
            if i < len(codes) and amb[i]:
                run += 1
                continue
            b = i - run
            if run >= 12 and b >= 2 and b + 12 <= len(codes):
                for code, n, pos in ((15, 12, b), (5, 6, b + 5), (10, 3, b - 2)):
                    extra.append(struct.pack(">Q", (code << 60) | ((n - 1) << 48) | pos) if big else struct.pack(">I", (code << 28) | ((n - 1) << 24) | pos))
                break
            run = 0
        ents = ents[::-1] + extra
        n = len(ents)
        return body, struct.pack(">I", (0x80000000 | (2 * n)) if big else n) + b"".join(ents)
    return wrapped


def build(d, overlap=False):
    rng = np.random.default_rng(23)
    acgt = np.array([1, 2, 4, 8], np.uint8)
    q = acgt[rng.integers(0, 4, 400)]
    seqs = []
    for k in range(300):
        n = int(rng.integers(50, 1500))
        s = acgt[rng.integers(0, 4, n)]
        if k % 3 == 0 and n > 450:                       # the query (or its reverse complement) with an ambiguous run inside
            a = int(rng.integers(0, n - 400))
            s[a:a + 400] = q if k % 2 == 0 else blastdb.revcomp_nt16(q)
            b = a + int(rng.integers(20, 300))
            s[b:b + int(rng.integers(1, 60))] = int(rng.choice([15, 5, 10, 3, 12]))
        if k % 10 == 0:
            s[0] = 15
            s[-1] = 14
        seqs.append(s)
    seqs[7][:] = 15
    half = 150
    va, vb, base = os.path.join(d, "va"), os.path.join(d, "vb"), os.path.join(d, "amb")
    old = blastdb.pack_nucleotide
    try:
        if overlap:
            blastdb.pack_nucleotide = disordered(old)
        blastdb.write_volume(va, seqs[:half], protein=False, ids=[f"s{i}" for i in range(half)])
        blastdb.pack_nucleotide = disordered(lambda c: pack_new_format(c, old)) if overlap else (lambda c: pack_new_format(c, old))
        blastdb.write_volume(vb, seqs[half:], protein=False, ids=[f"s{i}" for i in range(half, 300)])
    finally:
        blastdb.pack_nucleotide = old
    blastdb.write_alias(base, [va, vb], protein=False)
    qf = os.path.join(d, "q.fa")
    open(qf, "w").write(">query test\n" + "".join(blastdb.NCBI4NA[c] for c in q) + "\n")
    import hashlib
    h = hashlib.sha1()
    for ext in ("va.nsq", "vb.nsq", "va.nin", "vb.nin"):
        h.update(open(os.path.join(d, ext), "rb").read())
    return base, qf, h.hexdigest()


def main():
    for name, overlap in (("ntamb", False), ("ntamb_overlap", True)):
        with tempfile.TemporaryDirectory() as d:
            base, qf, sha = build(d, overlap)
            out = {"sha1_of_volumes": sha, "args": ARGS}
            for m in ("8", "0", "7"):
                r = subprocess.run([REF, "-d", base, "-i", qf, "-m", m] + ARGS, capture_output=True, text=True, check=True)
                text = r.stdout
                if m == "0":
                    text = text[text.index("Sequences producing"):]
                out["m" + m] = text
        json.dump(out, open(os.path.join(HERE, name + ".json"), "w"), indent=0)
        print(name + ".json:", {k: len(v) for k, v in out.items() if k.startswith("m")})


if __name__ == "__main__":
    main()
