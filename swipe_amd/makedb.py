"""FASTA -> BLAST database version 4 (the format SWIPE and swipe_amd read): a minimal stand-in for NCBI's
formatdb / `makeblastdb -blastdb_version 4`, which the reference relies on and this image lacks.

    python -m swipe_amd.makedb proteins.fasta mydb            # mydb.pin / .psq / .phr
    python -m swipe_amd.makedb --nucleotide genome.fa mydb    # mydb.nin / .nsq / .nhr, ambiguity runs kept
    python -m swipe_amd.makedb --volume-residues 1000000000 big.fasta mydb   # several volumes + mydb.pal

Definition lines become ``lcl|<first word> <rest>``; ``gi|N|...`` style ids are kept verbatim as local ids.
"""
from __future__ import annotations

import argparse
import sys

import numpy as np

from . import blastdb


def read_fasta(path: str, protein: bool):
    enc = blastdb.encode_protein if protein else blastdb.encode_nucleotide
    ids, titles, seqs, cur = [], [], [], []
    name = None

    def flush():
        if name is not None:
            seqs.append(enc("".join(cur)))

    with (sys.stdin if path == "-" else open(path)) as f:
        for line in f:
            if line.startswith(">"):
                flush()
                head = line[1:].strip()
                name, _, rest = head.partition(" ")
                ids.append(name or "seq%d" % len(ids))
                titles.append(rest)
                cur = []
            elif name is not None:
                cur.append(line.strip())
    flush()
    return ids, titles, seqs


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("fasta")
    ap.add_argument("basename")
    ap.add_argument("--nucleotide", action="store_true")
    ap.add_argument("--title", default=None)
    ap.add_argument("--volume-residues", type=int, default=3_500_000_000,
                    help="start a new volume after this many residues (v4 offsets are 32-bit: < 4 GiB per volume)")
    a = ap.parse_args(argv)
    protein = not a.nucleotide
    ids, titles, seqs = read_fasta(a.fasta, protein)
    if not seqs:
        print("no sequences in " + a.fasta, file=sys.stderr)
        return 1
    title = a.title or a.fasta
    lens = np.fromiter((len(s) for s in seqs), dtype=np.int64, count=len(seqs))
    bounds, acc = [0], 0
    for i, n in enumerate(lens):
        if acc and acc + n > a.volume_residues:
            bounds.append(i)
            acc = 0
        acc += int(n)
    bounds.append(len(seqs))
    if len(bounds) == 2:
        blastdb.write_volume(a.basename, seqs, protein=protein, ids=ids, titles=titles, title=title)
    else:
        names = []
        for v in range(len(bounds) - 1):
            lo, hi = bounds[v], bounds[v + 1]
            name = "%s.%02d" % (a.basename, v)
            blastdb.write_volume(name, seqs[lo:hi], protein=protein, ids=ids[lo:hi], titles=titles[lo:hi], title=title)
            names.append(name)
        blastdb.write_alias(a.basename, names, protein=protein, title=title)
    print("%d sequences, %d residues, %d volume(s) -> %s" % (len(seqs), int(lens.sum()), len(bounds) - 1, a.basename))
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
