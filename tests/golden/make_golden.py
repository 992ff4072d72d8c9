#!/usr/bin/env python3
"""Generate tests/golden/*.json by running the COMPILED REFERENCE (oracle/_ref/swipe and
oracle/_ref/ref_harness, built from /root/reference by oracle/Makefile) on the cases of
tests/cases.py.  Build-container only.  What is stored is data: per-sequence raw kernel outputs
(7-bit SSSE3 / 7-bit SSE2 / 16-bit + bestpos / 63-bit) and the reference CLI's ranked hit list
with its printed E-values and bit scores, for 1 and 8 threads."""
import json, os, re, subprocess, sys, tempfile
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(HERE))
import numpy as np
import cases
from swipe_amd import blastdb

REF = os.path.join(ROOT, "oracle", "_ref")

def fasta(path, case):
    alpha = blastdb.NCBI4NA if case.query_is_nt else blastdb.NCBISTDAA
    with open(path, "w") as f:
        f.write(">query test\n" + "".join(alpha[c] for c in case.query) + "\n")

def run(case):
    d = tempfile.mkdtemp(prefix="golden_")
    base = os.path.join(d, case.name)
    blastdb.write_db(base, case.seqs, protein=case.protein, volumes=case.volumes)
    qf = os.path.join(d, "q.fa")
    fasta(qf, case)
    mat = case.matrix
    if case.matrix == "@text":
        mat = os.path.join(d, "matrix.txt")
        open(mat, "w").write(case.matrix_text)
    sym = str(case.sym)
    translated = case.sym >= 2
    h = subprocess.run([os.path.join(REF, "ref_harness"), base, qf, sym, mat if case.sym != 0 else "-",
                        str(case.gapopen), str(case.gapextend), str(case.match), str(case.mismatch), "align",
                        str(case.query_gencode), str(case.db_gencode)],
                       capture_output=True, text=True, check=True)
    lines = h.stdout.splitlines()
    m = re.match(r"# SCORELIMIT_7=(-?\d+) SCORELIMIT_16=(-?\d+)", lines[0])
    if translated:
        raw = [list(map(int, l.split()[1:])) for l in lines[1:] if l.startswith("T\t")]
    else:
        raw = [list(map(int, l.split())) for l in lines[1:] if not l.startswith("A\t")]
    align = []
    k = 1 if translated else 0           # translated A lines carry (seqno, qtag, dtag) instead of (seqno, dstrand)
    for l in lines[1:]:
        if not l.startswith("A\t"):
            continue
        f = l.split("\t")
        row = [int(x) for x in f[1:6 + k]] + [int(x) for x in f[7 + k:12 + k]] + [f[12 + k]]
        row.append(None if f[14 + k] == "-" else [int(x) for x in f[14 + k:19 + k]] + [f[19 + k]])
        align.append(row)
    out = {"name": case.name, "checksum": case.checksum(), "nseq": len(case.seqs),
           "scorelimit7": int(m.group(1)), "scorelimit16": int(m.group(2)),
           "raw_columns": (["seqno", "qtag", "dtag"] if translated else ["seqno", "strand"]) +
                          ["len", "s7_ssse3", "s7_sse2", "s16", "bestpos16", "s63", "s16s", "bestpos16s", "bestq16s"],
           "raw": raw,
           "align_columns": (["seqno", "qtag", "dtag"] if translated else ["seqno", "dstrand"]) + ["s16s", "bestpos", "bestq", "score", "q_start", "d_start", "q_end", "d_end", "cigar",
                             "with_hint [score, q_start, d_start, q_end, d_end, cigar] or null"],
           "align": align, "cli": {}}
    common = [os.path.join(REF, "swipe"), "-d", base, "-i", qf, "-p", sym, "-G", str(case.gapopen), "-E", str(case.gapextend),
              "-v", str(case.keep), "-e", "10"]
    if case.sym != 0:
        common += ["-M", mat]
    else:
        common += ["-r", str(case.match), "-q", str(case.mismatch)]
    if translated:
        common += ["-Q", str(case.query_gencode), "-D", str(case.db_gencode)]
    for threads in (1, 8):
        x = subprocess.run(common + ["-a", str(threads), "-m", "7", "-b", "0"], capture_output=True, text=True, check=True)
        nalign = min(case.keep, 30)
        xa = subprocess.run(common + ["-a", str(threads), "-m", "7", "-b", str(nalign)], capture_output=True, text=True, check=True).stdout
        pa = subprocess.run(common + ["-a", str(threads), "-m", "0", "-b", str(nalign)], capture_output=True, text=True, check=True).stdout
        pa = pa[pa.index("Sequences producing"):] if "Sequences producing" in pa else ""
        t9 = subprocess.run(common + ["-a", str(threads), "-m", "9", "-b", str(case.keep)], capture_output=True, text=True, check=True).stdout
        t9 = "\n".join(t9.split("\n")[1:])          # first line carries the compile date
        if threads == 1:
            out["xml"] = x.stdout
            out["nalign"] = nalign
            out["xml_align"], out["plain_align"], out["tsv9"] = xa, pa, t9
        else:
            assert (out["xml_align"], out["plain_align"], out["tsv9"]) == (xa, pa, t9), "alignment output depends on thread count"
        tracks = list(map(int, re.findall(r"<track>(\d+)</track>", x.stdout)))
        scores = list(map(int, re.findall(r"<score>(-?\d+)</score>", x.stdout)))
        t = subprocess.run(common + ["-a", str(threads), "-m", "8", "-b", str(case.keep)], capture_output=True, text=True, check=True)
        if threads == 1:
            out["tsv"] = t.stdout
        else:
            assert out["tsv"] == t.stdout
        ev, bits = [], []
        for l in t.stdout.splitlines():
            f = l.split("\t")
            if len(f) >= 12:
                ev.append(f[10]); bits.append(f[11])
            elif len(f) == 11:
                ev.append(None); bits.append(f[10])
        p = subprocess.run(common + ["-a", str(threads), "-m", "0", "-b", "0"], capture_output=True, text=True, check=True)
        if case.sym == 0:
            strands = re.findall(r"^lcl\|\S+.*? ([+-]) +\d+ +\S+\s*$", p.stdout, re.M)
        elif translated:     # "+1", "-3" or "+1/-2" before the score column
            strands = re.findall(r"^lcl\|\S+.*? ([+-]\d(?:/[+-]\d)?) +\d+ +\S+\s*$", p.stdout, re.M)
        else:
            strands = []
        if threads == 1:
            lines = p.stdout.splitlines()
            k = next(i for i, l in enumerate(lines) if l.startswith("Sequences producing")) if any(l.startswith("Sequences producing") for l in lines) else None
            out["plain_hits"] = [l for l in lines[k + 2:] if l.strip()] if k is not None else []
        out["cli"][str(threads)] = {"seqno": tracks, "score": scores, "evalue": ev, "bits": bits, "strand": strands}
    return out

def main():
    names = sys.argv[1:] or [f.__name__[5:] for f in cases.ALL + cases.TRANSLATED]
    for n in names:
        c = cases.get(n)
        g = run(c)
        with open(os.path.join(HERE, n + ".json"), "w") as f:
            json.dump(g, f, separators=(",", ":"))
        a, b = g["cli"]["1"], g["cli"]["8"]
        print(n, "nseq", g["nseq"], "limits", g["scorelimit7"], g["scorelimit16"], "hits", len(a["seqno"]),
              "threads-identical", a == b, "max s63", max(r[-4] for r in g["raw"]),
              ">=lim7", sum(r[-8] >= g["scorelimit7"] for r in g["raw"]), ">=lim16", sum(r[-6] >= g["scorelimit16"] for r in g["raw"]))

if __name__ == "__main__":
    main()
