#!/usr/bin/env python3
"""tests/golden/headers.json: the reference CLI (oracle/_ref/swipe) on the `headers` case of tests/cases.py -
a protein volume whose definition lines use every Seq-id flavour, behind (a) nothing, (b) an OID-mask alias,
(c) a taxid list, (d) both; with and without -I (show gi's) and -H (show taxid etc.); with --dump,
tests/golden/dump.json: the -N 1 / -N 2 FASTA dumps.  Build container only."""
import json, os, subprocess, sys, tempfile
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(HERE))
import cases
from swipe_amd import blastdb

REF = os.path.join(ROOT, "oracle", "_ref", "swipe")


def build(case, d):
    vol = os.path.join(d, "vol")
    blastdb.write_volume(vol, case.seqs, protein=True, headers=case.extra["headers"], title="headers volume")
    inc = case.extra["include"]
    length = int(sum(len(s) for s, k in zip(case.seqs, inc) if k))
    blastdb.write_mask_alias(os.path.join(d, "masked"), vol, inc, memb_bit=1, length=length, title="masked subset")
    tx = os.path.join(d, "taxids.txt")
    open(tx, "w").write("".join("%d\n" % t for t in case.extra["taxids"]))
    qf = os.path.join(d, "q.fa")
    open(qf, "w").write(">query test\n" + "".join(blastdb.NCBISTDAA[c] for c in case.query) + "\n")
    return vol, os.path.join(d, "masked"), tx, qf


VARIANTS = {            # name -> (database, extra options)
    "plain": ("vol", []), "plain_gis": ("vol", ["-I"]), "plain_taxid": ("vol", ["-H"]), "plain_gis_taxid": ("vol", ["-I", "-H"]),
    "masked": ("masked", []), "masked_gis_taxid": ("masked", ["-I", "-H"]),
    "taxlist": ("vol", ["-x", "TAXIDS"]), "taxlist_gis_taxid": ("vol", ["-x", "TAXIDS", "-I", "-H"]),
    "masked_taxlist": ("masked", ["-x", "TAXIDS", "-H"]),
}


def main():
    case = cases.get("headers")
    d = tempfile.mkdtemp(prefix="golden_hdr_")
    vol, masked, tx, qf = build(case, d)
    out = {"name": "headers", "checksum": case.checksum(), "variants": {}}
    for name, (dbn, opts) in VARIANTS.items():
        db = vol if dbn == "vol" else masked
        opts = [tx if o == "TAXIDS" else o for o in opts]
        common = [REF, "-d", db, "-i", qf, "-v", str(case.keep), "-e", "1e6", "-a", "1"] + opts
        res = {}
        for m, b in (("0", "5"), ("7", "5"), ("8", str(case.keep))):
            r = subprocess.run(common + ["-m", m, "-b", b], capture_output=True, text=True, check=True).stdout
            if m == "0":
                head = [l for l in r.splitlines() if l.startswith("Database size") or l.startswith("Database title")]
                res["db_lines"] = head
                r = r[r.index("Sequences producing"):] if "Sequences producing" in r else r[r.index("No hits"):]
            res["m" + m] = r
        out["variants"][name] = res
        print(name, len(res["m8"].splitlines()), "aligned hits;", res["db_lines"])
    with open(os.path.join(HERE, "headers.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))




def dump_golden():
    """-N 1 / -N 2 (database dump as FASTA) of the same volume, plain / masked / with the taxid list, and of the
    nucleotide fixture (ambiguity codes)"""
    case = cases.get("headers")
    d = tempfile.mkdtemp(prefix="golden_dump_")
    vol, masked, tx, qf = build(case, d)
    out = {}
    for name, db, extra in (("plain", vol, []), ("masked", masked, []), ("taxlist_taxid", vol, ["-x", tx, "-H"])):
        for n in ("1", "2"):
            out[name + "_N" + n] = subprocess.run([REF, "-d", db, "-N", n] + extra, capture_output=True, text=True, check=True).stdout
    nt = cases.get("nt")
    base = os.path.join(d, "nt")
    blastdb.write_db(base, nt.seqs[295:], protein=False)
    out["nt_N1"] = subprocess.run([REF, "-d", base, "-p", "0", "-N", "1"], capture_output=True, text=True, check=True).stdout
    json.dump(out, open(os.path.join(HERE, "dump.json"), "w"), separators=(",", ":"))
    print("dump golden:", {k: len(v) for k, v in out.items()})


if __name__ == "__main__":
    dump_golden() if "--dump" in sys.argv else main()
